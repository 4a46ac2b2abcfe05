// Manager.worker-equivalent decode loop in plain C++ over the C ABI: no Python, no torch —
// what a native host (or an hxcpp build of the player) would do with libjsplayer_amd.so.
//
//   jsp_play clip.avi            prints one line per frame:  index key|inter slot significant crc32
//   jsp_play clip.avi --pipelined [--depth D]
//                                the same lines, decoded through jsp_decompress_*_async / jsp_wait with D frames in
//                                flight: the host stage of frame n+1 overlaps the uploads and kernels of frame n
//   jsp_play clip.avi --pipelined --quiet [--streams T] [--repeat R | --seconds S] [--depth D]
//                                end-to-end rate: T independent streams (threads, a codec instance each) play the clip R
//                                times (or over and over for S seconds, all streams for the same interval) from the file's
//                                bytes in pinned memory; prints one JSON line
//   ... --pipelined [--prefetch MB]  MSVideo1: the file's bytes go to the device in ranges of about MB megabytes (default 32), one range
//                                ahead of the one being decoded and every pass over the file anew (jsp_prefetch): the frames then queue
//                                no upload of their own.  --prefetch 0: a copy, or a read over the bus inside the kernel, per frame
//   jsp_play a.avi,b.avi --pipelined --devices 0,1,... [--streams T] [--quiet ...]
//                                streams sharded one per GPU inside this process: stream s plays file s (mod their number) on
//                                device devices[s mod G], a host thread, a codec instance and a frame pool each; the per-device
//                                (frames, pixels) counters are summed through jsp_reduce_counters (RCCL all-reduce when it loads).
//                                Without --quiet every stream's per-frame lines are printed under a "# stream s device d file" header
//
// It restates, in this project's own words, only what touches the codec:
//   * the container facts that select and feed it (AVIParser.hx:42-88,142-171; ParserUtils.hx:24-27):
//     avih size / rate, strh fourcc, strf depth + palette, `00dc`/`00db` chunks of LIST movi padded to even size;
//   * the frame-buffer pool of NUM_BUFFERS + 1 device frames that never hands out the buffer holding the
//     previous frame (Manager.hx:114-118,424-443,470-477);
//   * the DecompressI / DecompressP protocol with its identity test (Manager.hx:499-524) and
//     frames_differ_significantly for key frames (Manager.hx:392-421): its pixel loop comes with the decode (option
//     "key_frame_compare", jsp_key_frame_differs / jsp_wait) — no pass of the player's own over the two frames.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <deque>
#include <string>
#include <thread>
#include <vector>

#include "jsplayer_amd.h"

namespace {
constexpr int kInsignificantLines = 36;   // Manager.hx:61
constexpr int kNumBuffers = 8;            // Main.hx:148

uint32_t le32(const uint8_t* p) { return p[0] | p[1] << 8 | p[2] << 16 | (uint32_t)p[3] << 24; }
uint32_t crc32(const uint8_t* p, size_t n) {
    static uint32_t table[256];
    if (!table[1])
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = c & 1 ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

struct Clip {
    int X = 0, Y = 0, bpp = 32, kind = JSP_CODEC_SCREENPRESSOR;
    std::vector<uint8_t> palette;
    std::vector<std::pair<size_t, size_t>> frames;   // offset, padded size inside `bytes`
    std::vector<uint8_t> index_key;                  // idx1 key flags of the video chunks, in file order (empty: no index)
    struct Bytes {                                   // the file, in pinned host memory (jsp_host_alloc): uploads need no copy
        uint8_t* p = nullptr;
        size_t n = 0;
        const uint8_t* data() const { return p; }
        size_t size() const { return n; }
        ~Bytes() { jsp_host_free(p); }
        Bytes() = default;
        Bytes(const Bytes&) = delete;             // (owns pinned memory: a Clip is never copied)
        Bytes& operator=(const Bytes&) = delete;
    } bytes;
};

bool is_msvc(const uint8_t* f) {
    return !std::memcmp(f, "MSVC", 4) || !std::memcmp(f, "msvc", 4) || !std::memcmp(f, "CRAM", 4) || !std::memcmp(f, "\0\0\0\0", 4);
}

void walk(Clip& c, size_t lo, size_t hi, bool in_movi, uint8_t (&fourcc)[4], bool& video_stream, bool& have_video) {
    const uint8_t* d = c.bytes.data();
    for (size_t pos = lo; pos + 8 <= hi;) {
        const uint8_t* tag = d + pos;
        const size_t size = le32(d + pos + 4), body = pos + 8, padded = (size + 1) & ~size_t(1);
        if (!std::memcmp(tag, "LIST", 4)) {
            const bool movi = !std::memcmp(d + body, "movi", 4);
            walk(c, body + 4, std::min(body + size, hi), in_movi || movi, fourcc, video_stream, have_video);
        } else if (!std::memcmp(tag, "avih", 4)) {
            c.X = (int)le32(d + body + 32);
            c.Y = (int)le32(d + body + 36);
        } else if (!std::memcmp(tag, "strh", 4)) {
            video_stream = !std::memcmp(d + body, "vids", 4) && !have_video;
            if (video_stream) std::memcpy(fourcc, d + body + 4, 4);
        } else if (!std::memcmp(tag, "strf", 4) && video_stream) {
            c.bpp = d[body + 14] | d[body + 15] << 8;
            const uint8_t* f = std::memcmp(fourcc, "\0\0\0\0", 4) ? fourcc : d + body + 16;
            if (is_msvc(f)) c.kind = c.bpp == 8 ? JSP_CODEC_MSVIDEO1_8 : JSP_CODEC_MSVIDEO1_16;
            if (c.bpp == 8 && padded > 40) c.palette.assign(d + body + 40, d + body + padded);
            video_stream = false;
            have_video = true;
        } else if (in_movi && (!std::memcmp(tag, "00dc", 4) || !std::memcmp(tag, "00db", 4))) {
            c.frames.emplace_back(body, std::min(padded, c.bytes.size() - body));
        } else if (!std::memcmp(tag, "idx1", 4)) {   // AVIOLDINDEX: ckid, flags (0x10 = key frame), offset, size
            for (size_t e = body; e + 16 <= std::min(body + size, hi); e += 16)
                if (!std::memcmp(d + e, "00dc", 4) || !std::memcmp(d + e, "00db", 4)) c.index_key.push_back((le32(d + e + 4) & 0x10) ? 1 : 0);
        }
        pos = body + padded;
    }
}

bool load(const char* path, Clip& c) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    c.bytes.n = n > 0 ? (size_t)n : 0;
    c.bytes.p = static_cast<uint8_t*>(jsp_host_alloc(c.bytes.n + 64));
    const bool ok = c.bytes.p && n > 12 && std::fread(c.bytes.p, 1, (size_t)n, f) == (size_t)n;
    std::fclose(f);
    if (!ok || std::memcmp(c.bytes.data(), "RIFF", 4) || std::memcmp(c.bytes.data() + 8, "AVI ", 4)) return false;
    uint8_t fourcc[4] = {0, 0, 0, 0};
    bool video_stream = false, have_video = false;
    walk(c, 12, std::min(c.bytes.size(), (size_t)8 + le32(c.bytes.data() + 4)), false, fourcc, video_stream, have_video);
    if (c.index_key.size() != c.frames.size()) c.index_key.clear();   // an index that does not describe these chunks is ignored
    return c.X > 0 && c.Y > 0;
}
// Key flag of frame i as the loaders attach it: the index's when the file has one (DataLoader.hx:373-401), else
// (first frame) || IsKeyFrame(bytes) (DataLoaderAVISeq.hx:45).
bool frame_is_key(const Clip& c, jsp_codec* dec, size_t i) {
    if (!c.index_key.empty()) return c.index_key[i] != 0;
    return i == 0 || jsp_is_key_frame(dec, c.bytes.data() + c.frames[i].first, c.frames[i].second);
}
}  // namespace

// The same loop with up to `depth` frames in flight.  A slot is handed out when it is neither the previous frame of the
// last submitted frame, nor a destination in flight, nor showing a frame the oldest frame in flight may still be compared
// with.  Returns frames decoded, or -1.
// Throughput runs (--quiet): `warmup` untimed passes over the clip (the codec's buffers reach their size, the clocks ramp up), then every
// stream waits at `gate` and the timed passes begin together; *t0 / *t1 bracket this stream's timed passes.
struct Gate { std::atomic<int> waiting{0}; int parties = 1; };
using Clock = std::chrono::steady_clock;
// `seconds` > 0: the timed passes start the file over until that much time has passed since the gate opened and stop where they are (frames
// in flight are collected) — streams with files of different lengths then all run for the same interval.
// `sink`: the per-frame lines go there instead of stdout (several streams printing side by side).
// --prefetch MB (g_prefetch_bytes): the file's bytes go to the device in ranges of about that size, the range after the current one ahead of the frames being
// submitted (jsp_prefetch), every pass over the file anew; the frames' own uploads then fall away (MSVideo1; other codecs ignore it).
// what the streams' decoders counted (jsp_counter), summed when they are destroyed: frames re-run through the synchronous path, frames that shared a
// launch, frames that found their bytes in a prefetched range
std::atomic<long long> g_async_reruns{0}, g_paired_frames{0}, g_prefetched_frames{0};
size_t g_prefetch_bytes = 32u << 20;                     // (--prefetch 0: every frame finds its own way up)
long play_pipelined(const Clip& clip, int depth, int repeat, bool quiet, int warmup = 0, Gate* gate = nullptr, Clock::time_point* t0 = nullptr,
                    Clock::time_point* t1 = nullptr, int device = 0, double seconds = 0, std::string* sink = nullptr) {
    auto say = [&](const char* fmt, auto... a) {
        if (!sink) { std::printf(fmt, a...); return; }
        char line[256];
        std::snprintf(line, sizeof line, fmt, a...);
        *sink += line;
    };
    jsp_codec* dec = jsp_codec_create(clip.kind, clip.X, clip.Y, clip.bpp, clip.palette.empty() ? nullptr : clip.palette.data(),
                                      (int)clip.palette.size(), device);
    if (!dec) { std::fprintf(stderr, "jsp_codec_create: %s\n", jsp_last_error()); return -1; }
    jsp_preinit(dec, kInsignificantLines);
    if (clip.kind != JSP_CODEC_SCREENPRESSOR) {
        jsp_set_option(dec, "msv1_parse", "gpu");
        if (const char* form = std::getenv("JSP_PLAY_MSV1_ASYNC")) jsp_set_option(dec, "msv1_async", form);   // measurements: "two_launches"
    }
    char dbuf[16];
    std::snprintf(dbuf, sizeof dbuf, "%d", depth);
    jsp_set_option(dec, "async_depth", dbuf);
    std::snprintf(dbuf, sizeof dbuf, "%d", kInsignificantLines);
    if (!quiet) jsp_set_option(dec, "key_frame_compare", dbuf);   // key frames are compared with the frame before them as they are decoded
    jsp_pool* pool = jsp_pool_create(device, clip.X, clip.Y, kNumBuffers + 1 + depth);
    if (!pool) { std::fprintf(stderr, "jsp_pool_create: %s\n", jsp_last_error()); jsp_codec_destroy(dec); return -1; }
    const int nbuf = jsp_pool_count(pool);
    const size_t npx = (size_t)clip.X * clip.Y;
    std::vector<long> first(nbuf, -1), last(nbuf, -1);
    std::vector<int32_t> host(quiet ? 0 : npx);
    struct Flight { uint64_t ticket; size_t index; long gindex; bool key; int slot, prev_slot; int32_t* prev; bool cmp_bytes, first_ever; };
    std::deque<Flight> flying;
    long done = 0;
    bool failed = false;
    auto collect = [&] {
        const Flight f = flying.front();
        flying.pop_front();
        int32_t* shown_ptr = nullptr;
        int signif = -1, shown = f.slot;
        const int state = jsp_wait(dec, f.ticket, &shown_ptr, &signif);
        if (f.key) {
            if (state != JSP_ZERO_STATE) { if (!quiet) say("%zu key error %d %s\n", f.index, state, jsp_last_error()); return; }
            const int compared = signif;             // (option "key_frame_compare": the key frame against the frame before it)
            signif = -1;
            if (!quiet) {   // frames_differ_significantly, Manager.hx:392-421
                const uint8_t* src = clip.bytes.data() + clip.frames[f.index].first;
                const size_t len = clip.frames[f.index].second;
                if (f.first_ever) signif = 1;
                else if (f.cmp_bytes) {
                    const uint8_t* psrc = clip.bytes.data() + clip.frames[f.index - 1].first;
                    signif = !(clip.frames[f.index - 1].second == len && !std::memcmp(psrc, src, len));
                } else if (!f.prev) signif = 1;
                else signif = compared;
            }
        } else {
            if (state != JSP_ZERO_STATE) { if (!quiet) say("%zu inter raised %s\n", f.index, jsp_last_error()); return; }
            if (shown_ptr && shown_ptr == f.prev && f.prev_slot >= 0) shown = f.prev_slot;
            else if (!shown_ptr) shown = -1;
        }
        ++done;
        if (quiet) return;
        uint32_t crc = 0;
        if (shown >= 0 && jsp_download(jsp_pool_buffer(pool, shown), host.data(), npx) == 0)
            crc = crc32(reinterpret_cast<const uint8_t*>(host.data()), npx * 4);
        say("%zu %s %d %d %08x\n", f.index, f.key ? "key" : "inter", shown, signif, crc);
    };
    Clock::time_point deadline{};
    bool timed_out = false;
    if (seconds > 0) repeat = 1 << 30;
    struct Ahead { int pass; size_t begin, end; };          // frames [begin, end) of a pass whose bytes were handed to jsp_prefetch
    std::deque<Ahead> ahead;
    // A file that fits in ONE range and is played over and over: the next pass's range is the very bytes this pass is decoding from, and a frame
    // takes the NEWEST copy of its bytes (jsp_prefetch's rule) — so every pass would wait for an upload issued at its own start.  A player that reads
    // ahead has the next stretch of its file in ANOTHER buffer: passes alternate between two pinned copies of the file, the copy for pass p + 1 travels
    // while pass p is decoded from the other one.
    uint8_t* alt = nullptr;
    struct AltFree { uint8_t*& p; ~AltFree() { if (p) jsp_host_free(p); } } alt_free{alt};
    if (g_prefetch_bytes && clip.kind != JSP_CODEC_SCREENPRESSOR && !clip.frames.empty() && (repeat > 1 || seconds > 0 || warmup > 0)) {
        const size_t lo = clip.frames.front().first, hi = clip.frames.back().first + clip.frames.back().second;
        if (hi - lo <= g_prefetch_bytes) {
            alt = static_cast<uint8_t*>(jsp_host_alloc(clip.bytes.size() + 64));
            if (alt) std::memcpy(alt, clip.bytes.data(), clip.bytes.size());
        }
    }
    auto base = [&](int pass) -> const uint8_t* { return alt && (pass & 1) ? alt : clip.bytes.data(); };
    auto fetch = [&](int pass, size_t i0) {
        const size_t lo = clip.frames[i0].first;
        size_t j = i0, hi = lo;
        while (j < clip.frames.size() && (j == i0 || clip.frames[j].first + clip.frames[j].second - lo <= g_prefetch_bytes)) {
            hi = clip.frames[j].first + clip.frames[j].second;
            ++j;
        }
        if (jsp_prefetch(dec, base(pass) + lo, hi - lo) != 0) std::fprintf(stderr, "jsp_prefetch: %s\n", jsp_last_error());
        ahead.push_back({pass, i0, j});
    };
    for (int rep = -warmup; rep < repeat && !failed && !timed_out; ++rep) {
        if (rep == 0) {
            if (gate) { gate->waiting.fetch_add(1); while (gate->waiting.load() < gate->parties) std::this_thread::yield(); }
            const Clock::time_point now = Clock::now();
            if (t0) *t0 = now;
            deadline = now + std::chrono::duration_cast<Clock::duration>(std::chrono::duration<double>(seconds));
            done = 0;
        }
        bool last_was_key = false;
        for (size_t i = 0; i < clip.frames.size(); ++i) {
            if (seconds > 0 && rep >= 0 && Clock::now() >= deadline) { timed_out = true; break; }
            if ((int)flying.size() == depth) collect();
            if (g_prefetch_bytes && clip.kind != JSP_CODEC_SCREENPRESSOR) {
                while (!ahead.empty() && (ahead.front().pass != rep || i < ahead.front().begin || i >= ahead.front().end)) ahead.pop_front();
                if (ahead.empty()) fetch(rep, i);
                if (ahead.size() < 2) {                       // the range after this one travels while this one is decoded
                    const Ahead& b = ahead.back();
                    if (b.end < clip.frames.size()) fetch(b.pass, b.end);
                    else if (rep + 1 < repeat) fetch(b.pass + 1, 0);
                }
            }
            const uint8_t* src = base(rep) + clip.frames[i].first;
            const size_t len = clip.frames[i].second;
            const bool key = frame_is_key(clip, dec, i);
            int32_t* prev = jsp_previous_frame(dec);          // as of the last submitted frame
            int prev_slot = -1, slot = -1;
            for (int k = 0; k < nbuf; ++k) if (prev && jsp_pool_buffer(pool, k) == prev) prev_slot = k;
            const long gi = (long)((rep + warmup) * (long)clip.frames.size() + (long)i);
            const long horizon = flying.empty() ? gi : flying.front().gindex - 1;   // what may still be looked at
            {
                long oldest = 1L << 60;
                int victim = -1;
                for (int k = 0; k < nbuf && slot < 0; ++k) {
                    bool busy = k == prev_slot;
                    for (const Flight& f : flying) busy |= f.slot == k || f.prev_slot == k;
                    if (busy) continue;
                    if (first[k] < 0) slot = k;
                    else if (last[k] < horizon && first[k] < oldest) { oldest = first[k]; victim = k; }
                }
                if (slot < 0) { slot = victim; if (slot >= 0) first[slot] = last[slot] = -1; }
            }
            if (slot < 0) { std::fprintf(stderr, "no free frame buffer\n"); failed = true; break; }
            uint64_t ticket = 0;
            int32_t* dst = jsp_pool_buffer(pool, slot);
            const int rc = key ? jsp_decompress_i_async(dec, src, len, dst, &ticket) : jsp_decompress_p_async(dec, src, len, dst, &ticket);
            if (rc != 0) { std::fprintf(stderr, "submit: %s\n", jsp_last_error()); failed = true; break; }
            // which slot shows this frame is decided by the host stage: the previous frame after submission
            int32_t* now = jsp_previous_frame(dec);
            if (now == dst) first[slot] = last[slot] = gi;
            else if (now && now == prev && prev_slot >= 0) last[prev_slot] = gi;
            flying.push_back({ticket, i, gi, key, slot, prev_slot, prev, key && last_was_key && i > 0, rep == -warmup && i == 0});
            last_was_key = key;
        }
        while (!flying.empty()) collect();
    }
    if (t1) *t1 = Clock::now();
    for (auto [name, sum] : {std::pair<const char*, std::atomic<long long>*>{"async_reruns", &g_async_reruns}, {"paired_frames", &g_paired_frames},
                             {"prefetched_frames", &g_prefetched_frames}}) {
        const long long v = jsp_counter(dec, name);
        if (v > 0) sum->fetch_add(v);
    }
    jsp_pool_destroy(pool);
    jsp_codec_destroy(dec);
    return failed ? -1 : done;
}

// --batch B: the file through the batch calls (jsp_stage_batch / jsp_staged_decode), B frames at a time, every frame of a batch
// into a buffer of its own.  Prints what each frame shows as "<index> <key|inter> <adopted> <crc32>" (a frame that changes
// nothing shows the picture before it), or with --quiet plays the file `repeat` times and returns the frames decoded.
long play_batched(const Clip& clip, int batch, int repeat, bool quiet, int warmup = 0, double* seconds = nullptr) {
    jsp_codec* dec = jsp_codec_create(clip.kind, clip.X, clip.Y, clip.bpp, clip.palette.empty() ? nullptr : clip.palette.data(),
                                      (int)clip.palette.size(), 0);
    if (!dec) { std::fprintf(stderr, "jsp_codec_create: %s\n", jsp_last_error()); return -1; }
    jsp_preinit(dec, kInsignificantLines);
    if (clip.kind != JSP_CODEC_SCREENPRESSOR) jsp_set_option(dec, "msv1_parse", "gpu");
    jsp_pool* pool = jsp_pool_create(0, clip.X, clip.Y, batch + 1);     // + the picture carried over from the batch before
    if (!pool) { std::fprintf(stderr, "jsp_pool_create: %s\n", jsp_last_error()); jsp_codec_destroy(dec); return -1; }
    const size_t npx = (size_t)clip.X * clip.Y;
    std::vector<int32_t> host(quiet ? 0 : npx);
    std::vector<const uint8_t*> srcs(batch);
    std::vector<size_t> lens(batch);
    std::vector<uint8_t> keys(batch);
    std::vector<int32_t*> dsts(batch);
    std::vector<int> status(batch), adopted(batch), signif(batch);
    long done = 0;
    bool failed = false;
    jsp_staged* st = nullptr;
    auto t0 = std::chrono::steady_clock::now();
    for (int rep = -warmup; rep < repeat && !failed; ++rep) {
        if (rep == 0) { t0 = std::chrono::steady_clock::now(); done = 0; }   // (untimed passes first: buffers, clocks)
        int carry = batch;                                  // pool slot holding the last picture of the batch before (none yet: unused)
        bool have_picture = false;
        for (size_t f0 = 0; f0 < clip.frames.size() && !failed; f0 += batch) {
            const int n = (int)std::min<size_t>(batch, clip.frames.size() - f0);
            int slot = 0;
            for (int i = 0; i < n; ++i) {
                if (slot == carry) ++slot;                  // the picture the first inter frames are decoded against stays
                srcs[i] = clip.bytes.data() + clip.frames[f0 + i].first;
                lens[i] = clip.frames[f0 + i].second;
                keys[i] = frame_is_key(clip, dec, f0 + i) ? 1 : 0;
                dsts[i] = jsp_pool_buffer(pool, slot++);
            }
            jsp_staged* next = st ? jsp_restage_batch(dec, st, n, srcs.data(), lens.data(), keys.data(), dsts.data())
                                  : jsp_stage_batch(dec, n, srcs.data(), lens.data(), keys.data(), dsts.data());
            if (!next) { std::fprintf(stderr, "jsp_stage_batch: %s\n", jsp_last_error()); failed = true; break; }
            st = next;                                      // one batch object for the whole file: its buffers are taken over
            if (jsp_staged_decode(dec, st) != 0 || jsp_sync(dec) != 0) { std::fprintf(stderr, "decode: %s\n", jsp_last_error()); failed = true; }
            jsp_staged_results(st, status.data(), adopted.data(), signif.data());
            const int32_t* shown = have_picture ? jsp_pool_buffer(pool, carry) : nullptr;
            for (int i = 0; i < n && !failed; ++i) {
                if (adopted[i]) { shown = dsts[i]; have_picture = true; }
                ++done;
                if (quiet) continue;
                uint32_t crc = 0;
                if (shown && jsp_download(shown, host.data(), npx) == 0) crc = crc32(reinterpret_cast<const uint8_t*>(host.data()), npx * 4);
                std::printf("%zu %s %d %08x\n", f0 + i, keys[i] ? "key" : "inter", adopted[i], crc);
            }
            if (shown)
                for (int k = 0; k <= batch; ++k) if (jsp_pool_buffer(pool, k) == shown) carry = k;
        }
    }
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (st) jsp_staged_destroy(st);
    jsp_pool_destroy(pool);
    jsp_codec_destroy(dec);
    return failed ? -1 : done;
}

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s clip.avi [--pipelined [--depth D] [--quiet [--streams T] [--repeat R] [--warmup W]]] | --batch B [--quiet [--repeat R]]\n", argv[0]); return 2; }
    // (throughput runs: several files, separated by commas — stream s plays file s modulo their number, so that the streams of a
    // multi-stream run are independent inputs)
    std::deque<Clip> clips;                                   // (a deque: elements never move)
    {
        std::string list = argv[1];
        for (size_t at = 0; at <= list.size();) {
            const size_t comma = list.find(',', at);
            const std::string name = list.substr(at, comma == std::string::npos ? std::string::npos : comma - at);
            clips.emplace_back();
            if (!load(name.c_str(), clips.back())) { std::fprintf(stderr, "%s: not a RIFF/AVI file this player understands\n", name.c_str()); return 2; }
            if (comma == std::string::npos) break;
            at = comma + 1;
        }
    }
    const Clip& clip = clips[0];
    bool pipelined = false, quiet = false;
    int depth = 4, streams = 0, repeat = 1, warmup = 1, batch = 0, device = 0;
    double seconds = 0;
    std::vector<int> devices;                                 // --devices: streams sharded one per GPU (stream s -> devices[s mod G])
    for (int a = 2; a < argc; ++a) {
        const std::string o = argv[a];
        if (o == "--pipelined") pipelined = true;
        else if (o == "--quiet") quiet = true;
        else if (o == "--depth" && a + 1 < argc) depth = std::atoi(argv[++a]);
        else if (o == "--prefetch" && a + 1 < argc) g_prefetch_bytes = (size_t)(std::atof(argv[++a]) * 1048576.0);
        else if (o == "--streams" && a + 1 < argc) streams = std::atoi(argv[++a]);
        else if (o == "--repeat" && a + 1 < argc) repeat = std::atoi(argv[++a]);
        else if (o == "--seconds" && a + 1 < argc) seconds = std::atof(argv[++a]);
        else if (o == "--warmup" && a + 1 < argc) warmup = std::atoi(argv[++a]);
        else if (o == "--batch" && a + 1 < argc) batch = std::atoi(argv[++a]);
        else if (o == "--device" && a + 1 < argc) device = std::atoi(argv[++a]);
        else if (o == "--devices" && a + 1 < argc) {
            const std::string list = argv[++a];
            for (size_t at = 0; at <= list.size();) {
                const size_t comma = list.find(',', at);
                devices.push_back(std::atoi(list.substr(at, comma == std::string::npos ? std::string::npos : comma - at).c_str()));
                if (comma == std::string::npos) break;
                at = comma + 1;
            }
        }
        else { std::fprintf(stderr, "unknown option %s\n", argv[a]); return 2; }
    }
    if (batch > 0) {
        batch = batch > 1024 ? 1024 : batch;
        if (!quiet) return play_batched(clip, batch, 1, false) < 0 ? 1 : 0;
        double sec = 0;
        const long frames = play_batched(clip, batch, repeat, true, warmup, &sec);
        if (frames < 0) return 1;
        std::printf("{\"batch\": %d, \"frames\": %ld, \"seconds\": %.6f, \"mpixels_per_s\": %.1f}\n", batch, frames, sec,
                    frames * (double)clip.X * clip.Y / sec / 1e6);
        return 0;
    }
    if (!devices.empty()) {
        const int have = jsp_device_count();
        for (int d : devices)
            if (d < 0 || d >= have) { std::fprintf(stderr, "--devices: no device %d (%d visible)\n", d, have); return 2; }
        if (!pipelined || batch > 0) { std::fprintf(stderr, "--devices goes with --pipelined\n"); return 2; }
        if (streams <= 0) streams = (int)devices.size();      // one stream per listed device unless told otherwise
    }
    if (pipelined && !devices.empty() && !quiet) {
        // every stream prints what a single-device run of its file prints, under a header of its own
        depth = depth < 1 ? 1 : (depth > 16 ? 16 : depth);
        std::vector<std::string> lines(streams);
        std::vector<long> done(streams, 0);
        std::vector<std::thread> pool;
        for (int s = 0; s < streams; ++s)
            pool.emplace_back([&, s] {
                done[s] = play_pipelined(clips[(size_t)s % clips.size()], depth, 1, false, 0, nullptr, nullptr, nullptr,
                                         jsp_assign_stream(s, devices.data(), (int)devices.size()), 0, &lines[s]);
            });
        for (auto& t : pool) t.join();
        std::vector<uint64_t> per((size_t)devices.size() * 2, 0);
        for (int s = 0; s < streams; ++s) {
            if (done[s] < 0) return 1;
            const Clip& c = clips[(size_t)s % clips.size()];
            std::printf("# stream %d device %d file %zu\n%s", s, jsp_assign_stream(s, devices.data(), (int)devices.size()), (size_t)s % clips.size(), lines[s].c_str());
            per[2 * ((size_t)s % devices.size())] += (uint64_t)done[s];
            per[2 * ((size_t)s % devices.size()) + 1] += (uint64_t)done[s] * (uint64_t)c.X * (uint64_t)c.Y;
        }
        uint64_t total[2] = {0, 0};
        int via = 0;
        if (jsp_reduce_counters(devices.data(), (int)devices.size(), per.data(), total, &via) != JSP_ZERO_STATE) { std::fprintf(stderr, "counter reduce failed: %s\n", jsp_shard_last_error()); return 1; }
        std::printf("# total frames %llu pixels %llu reduce %s\n", (unsigned long long)total[0], (unsigned long long)total[1], via ? "rccl" : "host");
        return 0;
    }
    if (pipelined) {
        depth = depth < 1 ? 1 : (depth > 16 ? 16 : depth);
        if (!quiet) return play_pipelined(clip, depth, 1, false) < 0 ? 1 : 0;
        streams = streams < 1 ? 1 : streams;
        std::vector<long> done(streams, 0);
        std::vector<Clock::time_point> begin(streams), end(streams);
        std::vector<std::thread> pool;
        Gate gate;
        gate.parties = streams;
        for (int s = 0; s < streams; ++s)
            pool.emplace_back([&, s] {
                const int dev = devices.empty() ? device : jsp_assign_stream(s, devices.data(), (int)devices.size());
                done[s] = play_pipelined(clips[(size_t)s % clips.size()], depth, repeat, true, warmup, &gate, &begin[s], &end[s], dev, seconds);
            });
        for (auto& t : pool) t.join();
        Clock::time_point first = begin[0], last = end[0];
        for (int s = 1; s < streams; ++s) { if (begin[s] < first) first = begin[s]; if (end[s] > last) last = end[s]; }
        const double sec = std::chrono::duration<double>(last - first).count();
        long frames = 0;
        double compressed = 0;                                   // bytes handed to the decoders in the timed passes (what crosses the bus)
        for (int s = 0; s < streams; ++s) {
            if (done[s] < 0) return 1;
            frames += done[s];
            const Clip& c = clips[(size_t)s % clips.size()];
            double per_pass = 0;
            for (const auto& fr : c.frames) per_pass += (double)fr.second;
            compressed += per_pass * (c.frames.empty() ? 0.0 : (double)done[s] / (double)c.frames.size());
        }
        // streams sharded over several devices: the per-device counters, summed by the library (RCCL all-reduce when it loads)
        std::string shard;
        if (!devices.empty()) {
            std::vector<uint64_t> per(devices.size() * 2, 0);
            for (int s = 0; s < streams; ++s) {
                const Clip& c = clips[(size_t)s % clips.size()];
                per[2 * ((size_t)s % devices.size())] += (uint64_t)done[s];
                per[2 * ((size_t)s % devices.size()) + 1] += (uint64_t)done[s] * (uint64_t)c.X * (uint64_t)c.Y;
            }
            uint64_t total[2] = {0, 0};
            int via = 0;
            if (jsp_reduce_counters(devices.data(), (int)devices.size(), per.data(), total, &via) != JSP_ZERO_STATE || total[0] != (uint64_t)frames) {
                std::fprintf(stderr, "counter reduce failed: %s\n", jsp_shard_last_error());
                return 1;
            }
            shard = ", \"devices\": [";
            for (size_t i = 0; i < devices.size(); ++i) shard += (i ? ", " : "") + std::to_string(devices[i]);
            shard += "], \"per_device_frames\": [";
            for (size_t i = 0; i < devices.size(); ++i) shard += (i ? ", " : "") + std::to_string(per[2 * i]);
            shard += "], \"total_pixels\": " + std::to_string(total[1]) + ", \"counter_reduce\": \"" + (via ? "rccl" : "host") + "\"";
        }
        std::printf("{\"streams\": %d, \"files\": %zu, \"depth\": %d, \"frames\": %ld, \"warmup_passes\": %d, \"seconds\": %.6f, \"mpixels_per_s\": %.1f, "
                    "\"compressed_bytes\": %.0f, \"uploaded_bytes_per_s\": %.0f, \"async_reruns\": %lld, \"paired_frames\": %lld, \"prefetched_frames\": %lld%s}\n",
                    streams, clips.size(), depth, frames, warmup, sec, frames * (double)clip.X * clip.Y / sec / 1e6, compressed, compressed / sec,
                    g_async_reruns.load(), g_paired_frames.load(), g_prefetched_frames.load(), shard.c_str());
        return 0;
    }
    jsp_codec* dec = jsp_codec_create(clip.kind, clip.X, clip.Y, clip.bpp, clip.palette.empty() ? nullptr : clip.palette.data(),
                                      (int)clip.palette.size(), 0);
    if (!dec) { std::fprintf(stderr, "jsp_codec_create: %s\n", jsp_last_error()); return 1; }
    jsp_preinit(dec, kInsignificantLines);
    {
        char row[16];
        std::snprintf(row, sizeof row, "%d", kInsignificantLines);
        jsp_set_option(dec, "key_frame_compare", row);
    }
    jsp_pool* pool = jsp_pool_create(0, clip.X, clip.Y, kNumBuffers + 1);
    if (!pool) { std::fprintf(stderr, "jsp_pool_create: %s\n", jsp_last_error()); return 1; }
    const int nbuf = jsp_pool_count(pool);
    const size_t npx = (size_t)clip.X * clip.Y;
    std::vector<long> first(nbuf, -1), last(nbuf, -1);   // frames each slot currently shows (-1: free)
    std::vector<int32_t> host(npx);
    const uint8_t* prev_key = nullptr;
    size_t prev_key_len = 0;
    bool last_was_key = false;
    int rc = 0;
    for (size_t i = 0; i < clip.frames.size(); ++i) {
        const uint8_t* src = clip.bytes.data() + clip.frames[i].first;
        const size_t len = clip.frames[i].second;
        const bool key = frame_is_key(clip, dec, i);
        int32_t* prev = jsp_previous_frame(dec);
        int prev_slot = -1, slot = -1;
        for (int k = 0; k < nbuf; ++k) if (prev && jsp_pool_buffer(pool, k) == prev) prev_slot = k;
        {   // a free slot, else the one showing the oldest frames already behind the frame of interest
            long oldest = 1L << 60;
            int victim = -1;
            for (int k = 0; k < nbuf && slot < 0; ++k) {
                if (k == prev_slot) continue;
                if (first[k] < 0) slot = k;
                else if (last[k] < (long)i && first[k] < oldest) { oldest = first[k]; victim = k; }
            }
            if (slot < 0) { slot = victim; if (slot >= 0) first[slot] = last[slot] = -1; }
        }
        if (slot < 0) { std::fprintf(stderr, "no free frame buffer\n"); rc = 1; break; }
        int32_t* dst = jsp_pool_buffer(pool, slot);
        int shown = slot, signif = -1;
        if (key) {
            const int state = jsp_decompress_i(dec, src, len, dst);
            if (state != JSP_ZERO_STATE) { std::printf("%zu key error %d %s\n", i, state, jsp_last_error()); last_was_key = true; continue; }
            first[slot] = last[slot] = (long)i;
            if (i == 0) signif = 1;
            else if (last_was_key && prev_key) signif = !(prev_key_len == len && !std::memcmp(prev_key, src, len));
            else if (!prev) signif = 1;
            else { signif = jsp_key_frame_differs(dec); if (signif < 0) signif = 1; }
            prev_key = src;
            prev_key_len = len;
        } else {
            int32_t* shown_ptr = nullptr;
            if (jsp_decompress_p(dec, src, len, dst, &shown_ptr, &signif) != 0) {
                std::printf("%zu inter raised %s\n", i, jsp_last_error());
                last_was_key = false;
                continue;
            }
            if (shown_ptr && shown_ptr == prev && prev_slot >= 0) {          // "no changes": the old slot keeps showing
                last[prev_slot] = (long)i;
                shown = prev_slot;
            } else if (shown_ptr) {
                first[slot] = last[slot] = (long)i;
            } else
                shown = -1;
        }
        last_was_key = key;
        uint32_t crc = 0;
        if (shown >= 0 && jsp_download(jsp_pool_buffer(pool, shown), host.data(), npx) == 0)
            crc = crc32(reinterpret_cast<const uint8_t*>(host.data()), npx * 4);
        std::printf("%zu %s %d %d %08x\n", i, key ? "key" : "inter", shown, signif, crc);
    }
    jsp_pool_destroy(pool);
    jsp_codec_destroy(dec);
    return rc;
}
