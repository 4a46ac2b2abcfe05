// MSVideo1_16bit / MSVideo1_8bit behind the IVideoCodec-shaped C ABI.
// Host side = control flow of MSVideo1.hx:106-209 / 293-393 (early-outs, changes, block_changes,
// adoption of dst as prevFrame); pixels are produced by msv1_kernels.hip.  The descriptor table is
// built either by the sequential host parser (msv1_host.cpp) or on the GPU
// (msv1_parse_kernels.hip, option "msv1_parse" = "gpu").
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <memory>
#include <exception>
#include <thread>
#include <unordered_set>

#include "codec.h"
#include "msv1.h"
#include <msv1_fused_hooks.h>   // (angle brackets: a lab build puts its own in front on the include path, see the Makefile)

namespace jsp {
namespace {

// Published tile tables carry the launch's epoch; 0 is what freshly zeroed memory reads as, so the counter skips it when it wraps.
inline uint32_t next_epoch(uint32_t& e) { if (++e == 0) ++e; return e; }

struct Msv1Staged : jsp_staged {
    Msv1Geometry geo{};
    const int32_t* d_palette = nullptr;
    bool vec_ok = true;
    int nframes = 0;
    DeviceBuffer d_stream, d_desc, d_frames;
    PinnedBuffer h_stream, h_desc, h_frames;
    struct Group {
        int first, count;
        bool edge_compare;
        bool temporal;   // inter-frame group: one launch, frames walked in registers per spatial tile
        bool fused;      // parse + reconstruction in one launch straight from the stream bytes (no descriptor table)
    };
    std::vector<Group> groups;
    bool need_signif = false;  // some frame asked for the stage-2 compare

    // on-GPU parse state
    bool gpu_parse = false;
    int ntiles = 0, max_tiles = 0, insignificant_blocks = 0;
    DeviceBuffer d_pframes, d_tile_frame, d_tile_tab, d_tile_entry, d_tile_block0, d_info;
    PinnedBuffer h_pframes, h_tile_frame, h_info;
    // fused parse + reconstruction (msv1_fused_kernel): published tile tables, ticket / fault words
    DeviceBuffer d_agg, d_sync, d_recs, d_recs_emit, d_agg_emit;
    int ntiles_emit = 0;       // tiles of the table-writing form: it parses in 8 KiB tiles (twice the workgroups per CU, half the serial work in each)
    PinnedBuffer h_fault, h_recs, h_recs_emit;
    uint32_t epoch = 0;
    bool needs_desc = false;   // some launch reads the descriptor table the parse kernels build
    std::vector<uint32_t> scrub;   // (tests: option "msv1_scrub_tables") frames whose table a replay rewrites: poisoned before it does
    bool any_fused = false;
    // (Round 4 also wrote the replay's tables in PIECES of frames on a second stream, beside the launches that read the piece before — option
    // "msv1_parse_pieces": bit-exact and 12 - 50 % slower in every split, profiles/r04_msv1_parse_pieces.txt; removed in round 5, commit history has it.)

    // Replays of an inter-frame batch (round 6, option "msv1_parse_ahead", default on): the table-writing parse of the NEXT replay runs on the codec's
    // second stream beside this replay's temporal launch, into the OTHER of two table sets — the parse reads nothing but the stream bytes, so only the
    // tables' readers and writers have to be kept apart (events below).  Every replay still costs one parse launch and its reconstruction launches; what
    // changes is that the parse (0.15 - 0.19 ms of a 1.02 ms step at 512 x 1080p) no longer stands in front of the launch that needs it.
    hipStream_t side = nullptr;        // the codec's second stream; null: everything in line on the caller's stream
    hipEvent_t ev_fork = nullptr, ev_tables = nullptr;   // "the stream has reached this replay's reconstruction launches" / "the tables written ahead are complete"
    int cur_set = 0;                   // which set this replay's launches read
    bool ahead_valid = false;          // the tables of cur_set were written ahead (on `side`) and ev_tables says when
    // What a replay writes and reads (round 6): the COMPACT table where every launch that reads tables is a temporal launch (`compact_ok`, settled at
    // staging) — 2 bytes per block and a base per 256 blocks instead of 4 bytes per block, msv1.h —, else 4-byte tables as ever.  Set 0 of the 4-byte kind is
    // d_desc itself (what staging built: the first decode and the look-back fall-back always read that); every other set is made by the first replay that
    // needs it.  Tables a replay does not rewrite (frames the host parser settled) are put into every set when it is made.
    bool compact_ok = false;
    struct TableSet { DeviceBuffer wide, tab16, bases, recs; bool made = false; } sets[2];
    uint32_t* wide_tables(int i) { return static_cast<uint32_t*>(compact_ok || i == 0 ? d_desc.p : sets[i].wide.p); }
    void make_set(int i, hipStream_t stream) {
        TableSet& t = sets[i];
        if (t.made) return;
        const size_t nblk = (size_t)std::max(geo.nblocks, 1);
        std::vector<Msv1TileRec> recs((size_t)ntiles_emit);
        std::memcpy(recs.data(), h_recs_emit.p, sizeof(Msv1TileRec) * recs.size());
        if (compact_ok) {
            const size_t pitch = (size_t)msv1_tab16_pitch(geo.nblocks), ngr = (size_t)msv1_tab16_groups(geo.nblocks);
            t.tab16.reserve(sizeof(uint16_t) * pitch * (size_t)nframes);
            t.bases.reserve(sizeof(uint32_t) * ngr * (size_t)nframes);
            JSP_HIP(hipMemsetAsync(t.tab16.p, 0xFF, sizeof(uint16_t) * pitch * (size_t)nframes, stream));   // (MSV1_TAB16_UNTOUCHED)
            JSP_HIP(hipMemsetAsync(t.bases.p, 0, sizeof(uint32_t) * ngr * (size_t)nframes, stream));
            std::vector<uint32_t> kept;                // frames whose tables the replays leave alone: converted from the 4-byte tables staging uploaded
            const auto* pf = static_cast<const Msv1ParseFrame*>(h_pframes.p);
            for (int f = 0; f < nframes; ++f) if (pf[f].host_parsed) kept.push_back((uint32_t)f);
            if (!kept.empty()) {
                d_kept.reserve(sizeof(uint32_t) * kept.size());
                JSP_HIP(hipMemcpy(d_kept.p, kept.data(), sizeof(uint32_t) * kept.size(), hipMemcpyHostToDevice));
                msv1_launch_tables_compact(geo, static_cast<const uint32_t*>(d_desc.p), static_cast<uint16_t*>(t.tab16.p), static_cast<uint32_t*>(t.bases.p),
                                           static_cast<const uint32_t*>(d_kept.p), (int)kept.size(), stream);
            }
            for (auto& r : recs) {
                const size_t f = (size_t)(reinterpret_cast<uint32_t*>(r.dst) - static_cast<uint32_t*>(d_desc.p)) / nblk;   // (staging pointed the record at the frame's 4-byte table)
                r.dst = reinterpret_cast<int32_t*>(static_cast<uint16_t*>(t.tab16.p) + f * pitch);
                r.prev = reinterpret_cast<const int32_t*>(static_cast<uint32_t*>(t.bases.p) + f * ngr);
            }
        } else if (i != 0) {
            const size_t table_bytes = sizeof(uint32_t) * nblk * (size_t)nframes;
            t.wide.reserve(table_bytes);
            JSP_HIP(hipMemcpyAsync(t.wide.p, d_desc.p, table_bytes, hipMemcpyDeviceToDevice, stream));
            for (auto& r : recs) r.dst = reinterpret_cast<int32_t*>(static_cast<uint32_t*>(t.wide.p) + (reinterpret_cast<uint32_t*>(r.dst) - static_cast<uint32_t*>(d_desc.p)));
        }
        if (compact_ok || i != 0) {
            t.recs.reserve(sizeof(Msv1TileRec) * std::max<size_t>(recs.size(), 1));
            JSP_HIP(hipMemcpy(t.recs.p, recs.data(), sizeof(Msv1TileRec) * recs.size(), hipMemcpyHostToDevice));
        }
        t.made = true;
    }
    DeviceBuffer d_kept;
    void launch_table_parse(int set, hipStream_t on) {
        make_set(set, on);
        const TableSet& t = sets[set];
        const size_t pitch = (size_t)msv1_tab16_pitch(geo.nblocks);
        for (uint32_t i : scrub) {
            if (compact_ok) JSP_HIP(hipMemsetAsync(static_cast<uint16_t*>(t.tab16.p) + (size_t)i * pitch, 0xEE, sizeof(uint16_t) * (size_t)geo.nblocks, on));
            else JSP_HIP(hipMemsetAsync(wide_tables(set) + (size_t)i * (size_t)std::max(geo.nblocks, 1), 0xEE, sizeof(uint32_t) * (size_t)geo.nblocks, on));
        }
        msv1_launch_fused(geo, static_cast<const uint8_t*>(d_stream.p), static_cast<const Msv1TileRec*>(t.recs.p ? t.recs.p : d_recs_emit.p), d_palette,
                          static_cast<unsigned long long*>(d_agg_emit.p), next_epoch(epoch), 0, ntiles_emit, static_cast<uint32_t*>(d_sync.p), on,
                          nullptr, 0, compact_ok ? 5 : 4, 0, nullptr, nullptr, nullptr, (uint32_t)ntiles_emit, nullptr, /*small_tiles=*/true);   // (`want`: where a lab build's phase clocks go)
    }
    void drop_sets() { for (auto& t : sets) { t.wide.release(); t.tab16.release(); t.bases.release(); t.recs.release(); t.made = false; } }
    void quiesce_side() {              // nothing of this batch is left running beside the stream (before its buffers are reused or freed)
        if (side) (void)hipStreamSynchronize(side);
        ahead_valid = false;
        cur_set = 0;
    }
    ~Msv1Staged() override {
        if (side) (void)hipStreamSynchronize(side);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_tables) (void)hipEventDestroy(ev_tables);
    }

    void launch_parse(hipStream_t stream) {
        msv1_launch_parse(geo, static_cast<const uint8_t*>(d_stream.p), static_cast<const Msv1ParseFrame*>(d_pframes.p),
                          nframes, static_cast<const uint32_t*>(d_tile_frame.p), ntiles, max_tiles,
                          static_cast<uint32_t*>(d_tile_tab.p), static_cast<uint32_t*>(d_tile_entry.p),
                          static_cast<uint32_t*>(d_tile_block0.p), static_cast<uint32_t*>(d_desc.p),
                          static_cast<Msv1FrameInfo*>(d_info.p), insignificant_blocks, stream);
    }

    void decode(hipStream_t stream) override {
        if (nframes == 0) return;
        last_stream = stream;
        // A replay re-executes the whole device pipeline: with the on-GPU parse that includes the
        // parse kernels, so a staged batch can be timed from raw stream bytes.  (The first decode
        // uses the tables the staging pass left behind.)
        const bool replay = gpu_parse && decoded && needs_desc;    // one launch: the fused kernel's parse, writing block tables instead of pixels
        const bool run_ahead = side != nullptr && gpu_parse && needs_desc && ntiles_emit > 0;
        if (replay) {
            if (run_ahead && ahead_valid) JSP_HIP(hipStreamWaitEvent(stream, ev_tables, 0));   // written beside the replay before this one
            else launch_table_parse(cur_set, stream);
        }
        const bool compact = replay && compact_ok;                 // this decode's temporal launches read the compact tables of cur_set
        const uint32_t* tables = replay ? wide_tables(cur_set) : static_cast<const uint32_t*>(d_desc.p);
        const uint16_t* tab16 = compact ? static_cast<const uint16_t*>(sets[cur_set].tab16.p) : nullptr;
        const uint32_t* bases = compact ? static_cast<const uint32_t*>(sets[cur_set].bases.p) : nullptr;
        if (run_ahead && decoded) {
            // the next replay's tables, into the other set, beside the launches below: the other set's last readers (the replay before this one) and the
            // last table-writing launch (it shares the published tile words and the fault word) are all in front of `ev_fork` on the stream
            try {
                make_set(cur_set ^ 1, stream);
            } catch (const std::exception&) {          // no room for a second table set: this batch's replays parse in line from now on
                (void)hipGetLastError();
                { TableSet& t = sets[cur_set ^ 1]; t.wide.release(); t.tab16.release(); t.bases.release(); t.recs.release(); t.made = false; }
                side = nullptr;
            }
        }
        if (run_ahead && decoded && side) {
            if (!ev_fork) JSP_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
            if (!ev_tables) JSP_HIP(hipEventCreateWithFlags(&ev_tables, hipEventDisableTiming));
            JSP_HIP(hipEventRecord(ev_fork, stream));
            JSP_HIP(hipStreamWaitEvent(side, ev_fork, 0));
            launch_table_parse(cur_set ^ 1, side);
            JSP_HIP(hipEventRecord(ev_tables, side));
        }
        if (need_signif) JSP_HIP(hipMemsetAsync(d_signif.p, 0, sizeof(uint32_t) * nframes, stream));
        const auto* frames = static_cast<const Msv1FrameArgs*>(d_frames.p);
        for (const Group& g : groups) {
            if (g.fused) {
                const auto* pf = static_cast<const Msv1ParseFrame*>(h_pframes.p);
                const uint32_t tile0 = pf[g.first].first_tile;
                const Msv1ParseFrame& last = pf[g.first + g.count - 1];
                msv1_launch_fused(geo, static_cast<const uint8_t*>(d_stream.p), static_cast<const Msv1TileRec*>(d_recs.p), d_palette,
                                  static_cast<unsigned long long*>(d_agg.p), next_epoch(epoch), tile0, (int)(last.first_tile + last.ntiles - tile0),
                                  static_cast<uint32_t*>(d_sync.p), stream, nullptr, 0, 0, 0, nullptr, nullptr, nullptr, (uint32_t)ntiles);
            } else if (g.temporal) {
                msv1_launch_blocks_temporal(geo, static_cast<const uint8_t*>(d_stream.p), tables, frames + g.first, g.count, d_palette, stream, tab16, bases);
            } else {
                msv1_launch_blocks(geo, static_cast<const uint8_t*>(d_stream.p), tables, frames + g.first, g.count, d_palette, vec_ok, stream);
            }
            if (g.edge_compare) msv1_launch_edge_compare(geo, frames + g.first, g.count, stream);
        }
        JSP_HIP(hipGetLastError());
        if (need_signif)
            JSP_HIP(hipMemcpyAsync(h_signif.p, d_signif.p, sizeof(uint32_t) * nframes, hipMemcpyDeviceToHost,
                                   stream));
        if (any_fused || needs_desc)
            JSP_HIP(hipMemcpyAsync(h_fault.p, d_sync.p, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        ++clock_launches;
        if (run_ahead && decoded && side) { cur_set ^= 1; ahead_valid = true; }   // the next replay reads what was just started beside this one
        decoded = true;
    }
    int clock_launches = 0;                                    // (read by the lab build's phase clocks only: msv1_fused_hooks.h)
    void after_sync() override {
        if (kFusedClocks && clock_launches) {   // lab build only: per-phase cycle sums since the last sync (see JSP_CLOCK), printed and cleared
            std::vector<unsigned long long> c((size_t)ntiles * 8);
            unsigned long long* dev = static_cast<unsigned long long*>(d_agg.p) + (size_t)ntiles * 9;
            JSP_HIP(hipMemcpy(c.data(), dev, c.size() * 8, hipMemcpyDeviceToHost));
            JSP_HIP(hipMemset(dev, 0, c.size() * 8));
            unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (size_t i = 0; i < c.size(); ++i) h[i & 7] += c[i];
            const unsigned long long n = (unsigned long long)ntiles * clock_launches;
            std::fprintf(stderr, "fused clocks per tile (cycles; %d tiles x %d launches): load %llu | tables+trees %llu | look-back %llu | down %llu | replay %llu | decode %llu\n",
                         ntiles, clock_launches, h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n, h[5] / n);
        }
        clock_launches = 0;
        if ((any_fused || needs_desc) && (*static_cast<const uint32_t*>(h_fault.p) || inject_fault)) {
            // A tile gave up waiting for the tile before it.  That says something about the GPU's timing (shared with other
            // work, serialised by a profiler), nothing about the stream: the batch is decoded again through the descriptor
            // path — msv1_parse_tiles / _chain / _emit build the block tables in three launches with no hand-off inside a
            // launch, the block kernels paint from them — and only what that path reports counts.
            inject_fault = false;
            ++fallback_runs;
            if (fallback_total) ++*fallback_total;
            note_kernel("msv1_parse_tiles (look-back fallback)");
            *static_cast<uint32_t*>(h_fault.p) = 0;
            hipStream_t stream = last_stream;
            quiesce_side();                            // (a table-writing launch may be running beside the stream: it shares the fault word, and its tables are not to be trusted either)
            JSP_HIP(hipMemsetAsync(d_sync.p, 0, 2 * sizeof(uint32_t) + 64, stream));
            launch_parse(stream);
            if (need_signif) JSP_HIP(hipMemsetAsync(d_signif.p, 0, sizeof(uint32_t) * nframes, stream));
            const auto* frames = static_cast<const Msv1FrameArgs*>(d_frames.p);
            for (const Group& g : groups) {
                if (g.temporal)
                    msv1_launch_blocks_temporal(geo, static_cast<const uint8_t*>(d_stream.p), static_cast<const uint32_t*>(d_desc.p), frames + g.first,
                                                g.count, d_palette, stream);
                else
                    msv1_launch_blocks(geo, static_cast<const uint8_t*>(d_stream.p), static_cast<const uint32_t*>(d_desc.p), frames + g.first, g.count,
                                       d_palette, vec_ok, stream);
                if (g.edge_compare) msv1_launch_edge_compare(geo, frames + g.first, g.count, stream);
            }
            JSP_HIP(hipGetLastError());
            if (need_signif) JSP_HIP(hipMemcpyAsync(h_signif.p, d_signif.p, sizeof(uint32_t) * nframes, hipMemcpyDeviceToHost, stream));
            JSP_HIP(hipStreamSynchronize(stream));
        }
    }
    hipStream_t last_stream = nullptr;   // where decode() queued its launches
    bool inject_fault = false;           // tests (option "msv1_inject_fault"): the next after_sync() behaves as if a tile had given up
    int fallback_runs = 0;
    std::shared_ptr<std::atomic<long long>> fallback_total;   // the codec's jsp_counter("lookback_fallbacks")
};

// One frame on the asynchronous path (jsp_decompress_*_async) with the on-GPU parse.  Nothing here waits for the GPU; what the
// host could not know (stage-1 / stage-2 significance, a stream the host parser has to settle) is read from the frame's
// report once its event has fired (Msv1Codec::async_finish).  Frames of up to MSV1_MERGED_MAX_TILES tiles take ONE launch
// (msv1_fused_kernel mode 3: parse, frame-wide verdict, then the pixels or nothing; the record is a kernel argument, the
// report lands in pinned memory); larger ones, or "msv1_async" = "two_launches", a scout launch and a decode launch it may
// veto, with per-tile records and the report copied up and down.
struct Msv1AsyncStaged : jsp_staged {
    Msv1AsyncStaged() { verdict_pending = true; }
    Msv1Geometry geo{};
    const int32_t* d_palette = nullptr;
    int ntiles = 0, insignificant_blocks = 0;
    size_t nbytes = 0;
    bool have_prev = false, compare = false, key = false;
    bool key_compare = false;                 // a key frame whose stage-2 compare answers option "key_frame_compare" (the kernel compares as it decodes)
    uint32_t* d_poison = nullptr;             // the codec's veto word (see msv1_launch_fused)
    DeviceBuffer d_stream, d_meta, d_agg;     // d_meta = [tile records | Msv1AsyncInfo] (two-launch form)
    PinnedBuffer h_stream, h_meta, h_info;
    uint32_t epoch = 0;                        // never reset: d_agg is zeroed only when it is (re)allocated
    size_t agg_tiles = 0;
    // one-launch form (frames of at most MSV1_MERGED_MAX_TILES tiles): the record all tiles share travels as a kernel
    // argument, the report comes back through pinned memory, and the kernel reads the frame's bytes from pinned host
    // memory itself (the caller's, or h_stream), leaving a copy in d_stream: no copy is queued at all
    bool merged = false, small_tiles = false;
    bool deaf = false;                        // tests: see MSV1_LAB_DEAF
    Msv1TileRec rec{};
    DeviceBuffer d_report;                    // one Msv1AsyncInfo, allocated (and zeroed) once: its counters run on
    uint32_t want = 0;                        // ... to this value once every launch so far is through
    const uint8_t* src_dev = nullptr;         // device-side address of the frame's bytes in pinned host memory
    hipEvent_t uploaded = nullptr;            // (copy-engine form) the frame's bytes are in d_stream
    bool dma = false;
    ~Msv1AsyncStaged() override { if (uploaded) (void)hipEventDestroy(uploaded); }

    Msv1AsyncInfo* d_info() const {
        if (merged) return static_cast<Msv1AsyncInfo*>(d_report.p);
        return reinterpret_cast<Msv1AsyncInfo*>(static_cast<uint8_t*>(d_meta.p) + sizeof(Msv1TileRec) * (size_t)ntiles);
    }
    uint32_t bad_mask() const { return MSV1_ASYNC_SHORT | MSV1_ASYNC_END | MSV1_ASYNC_STUCK | (have_prev ? 0u : MSV1_ASYNC_SKIPCODE) | (deaf ? MSV1_LAB_DEAF : 0u); }
    // This frame and the ones submitted right behind it (at most MSV1_MAX_RIDERS) in ONE launch: all one-launch frames with the same tile
    // size (msv1.h, Msv1Riders).
    void decode_with(const std::vector<Msv1AsyncStaged*>& behind, hipStream_t stream) {
        want += (uint32_t)ntiles;
        if (dma) JSP_HIP(hipStreamWaitEvent(stream, uploaded, 0));
        Msv1Riders riders;
        for (Msv1AsyncStaged* next : behind) {
            next->want += (uint32_t)next->ntiles;
            if (next->dma) JSP_HIP(hipStreamWaitEvent(stream, next->uploaded, 0));
            Msv1Rider& r = riders.f[riders.count++];
            r.stream = next->src_dev;
            r.agg = static_cast<unsigned long long*>(next->d_agg.p);
            r.info = next->d_info();
            r.host_info = static_cast<Msv1AsyncInfo*>(next->h_info.p);
            r.keep = next->dma ? nullptr : static_cast<uint8_t*>(next->d_stream.p);
            r.rec = next->rec;
            r.epoch = next_epoch(next->epoch);
            r.bad_mask = next->bad_mask();
            r.want = next->want;
        }
        auto* info_dev = d_info();
        msv1_launch_fused(geo, src_dev, nullptr, d_palette, static_cast<unsigned long long*>(d_agg.p),
                          next_epoch(epoch), 0, ntiles, &info_dev->fault, stream, info_dev, insignificant_blocks, 3, bad_mask(), d_poison, &rec,
                          static_cast<Msv1AsyncInfo*>(h_info.p), want, dma ? nullptr : static_cast<uint8_t*>(d_stream.p), small_tiles, &riders);
        JSP_HIP(hipGetLastError());
        decoded = true;
        for (Msv1AsyncStaged* next : behind) next->decoded = true;
    }
    void decode(hipStream_t stream) override {
        auto* info_dev = d_info();
        const uint32_t bad = bad_mask();
        if (merged) {
            want += (uint32_t)ntiles;
            if (dma) JSP_HIP(hipStreamWaitEvent(stream, uploaded, 0));
            msv1_launch_fused(geo, src_dev, nullptr, d_palette, static_cast<unsigned long long*>(d_agg.p),
                              next_epoch(epoch), 0, ntiles, &info_dev->fault, stream, info_dev, insignificant_blocks, 3, bad, d_poison, &rec,
                              static_cast<Msv1AsyncInfo*>(h_info.p), want, dma ? nullptr : static_cast<uint8_t*>(d_stream.p), small_tiles);
            JSP_HIP(hipGetLastError());
            decoded = true;
            return;
        }
        for (int mode = 1; mode <= 2; ++mode)   // scout, then the decode it may veto
            msv1_launch_fused(geo, static_cast<const uint8_t*>(d_stream.p), static_cast<const Msv1TileRec*>(d_meta.p), d_palette,
                              static_cast<unsigned long long*>(d_agg.p), next_epoch(epoch), 0, ntiles, &info_dev->fault, stream, info_dev,
                              insignificant_blocks, mode, bad, d_poison, nullptr, nullptr, 0, nullptr, small_tiles);
        JSP_HIP(hipGetLastError());
        JSP_HIP(hipMemcpyAsync(h_info.p, info_dev, sizeof(Msv1AsyncInfo), hipMemcpyDeviceToHost, stream));
        decoded = true;
    }
};

// MSVideo1 codec instances of this process that have used the asynchronous one-launch path (see "msv1_async" = "auto")
std::atomic<int> g_async_streams{0};
constexpr int kDmaStreams = 3;

struct Msv1Codec : jsp_codec {
    Msv1Geometry geo{};
    size_t size_of_just_skips = 0;
    int insignificant_blocks = 0;
    bool insign_lines_set = false;  // the 8-bit Preinit never sets it (MSVideo1.hx:281-291)
    int insign_lines = 0;
    std::vector<uint8_t> block_changes;  // persists across calls, like the reference's member
    std::vector<uint8_t> palette_bytes;
    int32_t palette[256];
    DeviceBuffer d_palette;
    bool opt_gpu_parse = true;    // "msv1_parse": frames are parsed on the GPU unless the caller asks for the host parser
    bool opt_scrub_tables = false;
    bool opt_inject_fault = false;
    bool opt_inject_deaf = false;        // tests ("msv1_inject_fault" = "2"): the next one-launch asynchronous frame never sees all its tiles report
    std::shared_ptr<std::atomic<long long>> lookback_fallbacks = std::make_shared<std::atomic<long long>>(0);
    long long counter(const char* name) override {
        if (std::strcmp(name, "prefetched_frames") == 0) return prefetched_frames;
        if (std::strcmp(name, "paired_frames") == 0) return paired_frames;
        return std::strcmp(name, "lookback_fallbacks") == 0 ? lookback_fallbacks->load() : -1;
    }
    bool opt_async_merged = true, opt_async_dma = true, opt_async_auto = true;
    bool counted_async = false;   // this instance is in g_async_streams
    // The copy engine takes a frame's bytes up on a stream of its own, next to the previous frame's kernel.  One such stream is enough: a
    // megabyte per copy goes at ~30 GB/s on one stream where the bus takes 57 (bench.py: e2e.h2d_ceiling_GBs), but taking consecutive frames up
    // on 2 - 4 streams in turn (JSP_MSV1_UP_STREAMS, lab) changes nothing for one player stream (55.5 Gpixels/s with 1, 2, 3 or 4: the frames'
    // kernels follow each other on one HIP stream, 37 us apiece, and that chain is the bound) and costs 3 - 40 % with two (profiles/
    // r04_msv1_up_streams.txt).
    static constexpr int kUpStreamsMax = 4;
    hipStream_t up_streams[kUpStreamsMax] = {nullptr, nullptr, nullptr, nullptr};
    int up_count = [] { const char* e = std::getenv("JSP_MSV1_UP_STREAMS"); const int v = e ? std::atoi(e) : 1; return v < 1 ? 1 : (v > kUpStreamsMax ? kUpStreamsMax : v); }();
    unsigned up_next = 0;
    // jsp_prefetch: ranges of the caller's host memory that are (being) copied to the device in ONE piece each, on a stream of their own.
    // An asynchronous frame whose bytes lie inside such a range takes them from the device copy: no copy of its own is queued and
    // nothing crosses the bus inside its kernel.  A megabyte per copy goes at 24 - 39 GB/s, 64 MB at 57 (bench.py: e2e.h2d_ceiling_GBs),
    // and sixteen player streams that each queue a copy, an event and a wait per frame are bound by those queues, not by the bus.
    // A ring: the range prefetched kRanges calls ago is given up (after the kernels that read it: `last_use`).
    struct UpRange {
        const uint8_t* host = nullptr;
        size_t bytes = 0, skew = 0;           // the device copy starts at dev.p + skew: same offset inside a 64-byte line as `host`
        DeviceBuffer dev;
        hipEvent_t up = nullptr, last_use = nullptr;
        bool waited = false;                  // the codec's stream (`waited_on`) already waits for `up`
        hipStream_t waited_on = nullptr;
        bool used = false;                    // some kernel read it since it was uploaded
    };
    static constexpr int kRanges = 4;
    UpRange ranges[kRanges];
    unsigned range_next = 0;
    hipStream_t prefetch_stream = nullptr;
    long long prefetched_frames = 0;          // jsp_counter("prefetched_frames"): asynchronous frames that found their bytes on the device
    int prefetch(const void* host, size_t n) override {
        if (!host || n == 0) {                // give every range up (the caller's memory changes or goes away)
            for (UpRange& r : ranges) { r.host = nullptr; r.bytes = 0; }
            return 0;
        }
        if (!opt_gpu_parse) return 0;         // (the host parser reads the caller's bytes: nothing to take up)
        launch_held();                        // (frames held for their successors may read the range that is about to be given up: their kernels go out first)
        UpRange& r = ranges[range_next++ % kRanges];
        r.host = nullptr;                     // (not to be found while it is being replaced)
        r.bytes = 0;
        if (!prefetch_stream) JSP_HIP(hipStreamCreateWithFlags(&prefetch_stream, hipStreamNonBlocking));
        if (!r.up) {
            JSP_HIP(hipEventCreateWithFlags(&r.up, hipEventDisableTiming));
            JSP_HIP(hipEventCreateWithFlags(&r.last_use, hipEventDisableTiming));
        }
        if (r.used) {                         // kernels queued so far may still read what the range held: the copy goes behind them
            JSP_HIP(hipEventRecord(r.last_use, stream));
            JSP_HIP(hipStreamWaitEvent(prefetch_stream, r.last_use, 0));
        }
        r.dev.reserve(n + 128);               // (growing frees the old copy: hipFree waits for the device)
        r.skew = (size_t)(reinterpret_cast<uintptr_t>(host) & 63u);
        JSP_HIP(hipMemcpyAsync(static_cast<uint8_t*>(r.dev.p) + r.skew, host, n, hipMemcpyHostToDevice, prefetch_stream));
        JSP_HIP(hipEventRecord(r.up, prefetch_stream));
        r.host = static_cast<const uint8_t*>(host);
        r.bytes = n;
        r.waited = false;
        r.used = false;
        return 0;
    }
    UpRange* range_of(const uint8_t* src, size_t n) {
        // NEWEST first: a caller that hands the same host address over again (a ring of file chunks, a file played again) means the bytes that
        // are there NOW — an older copy of that address may hold what was there before.  (Oldest first, until round 4, decoded from a copy two
        // passes old when a short file was played over and over.)
        for (int k = kRanges - 1; k >= 0; --k) {
            UpRange& r = ranges[(range_next + (unsigned)k) % kRanges];
            if (r.host && src >= r.host && n <= r.bytes && (size_t)(src - r.host) <= r.bytes - n) return &r;
        }
        return nullptr;
    }
    ~Msv1Codec() override {
        if (prefetch_stream) (void)hipStreamSynchronize(prefetch_stream);
        for (UpRange& r : ranges) {
            if (r.up) (void)hipEventDestroy(r.up);
            if (r.last_use) (void)hipEventDestroy(r.last_use);
        }
        if (prefetch_stream) (void)hipStreamDestroy(prefetch_stream);
        for (hipStream_t s : up_streams) if (s) (void)hipStreamDestroy(s);
        if (counted_async) g_async_streams.fetch_sub(1);
    }
    // block_changes is only maintained by the host parser; after frames parsed on the GPU it is
    // rebuilt on demand from the bytes of the last fully parsed frame
    bool block_changes_stale = false;
    std::vector<uint8_t> last_full_frame;
    // asynchronous path: the last fully parsed frame's bytes stay where they are, in HBM, until somebody needs them
    const void* last_full_dev = nullptr;
    size_t last_full_dev_bytes = 0;
    DeviceBuffer d_poison;   // asynchronous path: set by a vetoed decode pass, cleared by async_reset()
    void async_reset() override {
        held.clear();            // (frames held for their successors are among those about to be re-run: they are never launched)
        if (d_poison.p) JSP_HIP(hipMemsetAsync(d_poison.p, 0, sizeof(uint32_t), stream));
    }
    // Replays of staged inter-frame batches (Msv1Staged::decode): "msv1_parse_ahead" (default on) and "msv1_compact_tables" (default OFF: measured, round 6 — the
    // compact tables take 265 MB off a 5.15 GB step's traffic and nothing off its time: the temporal launch takes 868 us with either table, the table-writing
    // launch 168 us instead of 154; profiles/r06_msv1_inter70_compact_tables_*.  Kept as an option: half the table memory.)
    bool opt_compact_tables = [] { const char* e = std::getenv("JSP_MSV1_COMPACT_TABLES"); return e && e[0] == '1'; }();
    bool opt_parse_ahead = [] { const char* e = std::getenv("JSP_MSV1_PARSE_AHEAD"); return !(e && e[0] == '0'); }();
    // Several frames per launch (option "msv1_async_pairs", default on): a one-launch frame is HELD until enough frames are submitted behind it
    // — half of what may be in flight ("async_depth"), at most 1 + MSV1_MAX_RIDERS — and they go out together (Msv1AsyncStaged::decode_with);
    // or with whatever is held, as soon as anybody waits for one of them or anything else needs the stream.
    bool opt_async_pairs = [] { const char* e = std::getenv("JSP_MSV1_ASYNC_PAIRS"); return !(e && e[0] == '0'); }();
    std::vector<jsp_async_job*> held;
    long long paired_frames = 0;      // jsp_counter("paired_frames"): frames that shared a launch with others
    int frames_per_launch() const {
        static const int lab = [] { const char* e = std::getenv("JSP_MSV1_FRAMES_PER_LAUNCH"); return e ? std::atoi(e) : 0; }();
        const int k = lab > 0 ? lab : async_depth / 2;
        return k < 1 ? 1 : (k > 1 + MSV1_MAX_RIDERS ? 1 + MSV1_MAX_RIDERS : k);
    }
    void launch_held() {
        if (held.empty()) return;
        activate();
        std::vector<jsp_async_job*> group;
        group.swap(held);
        auto* first = static_cast<Msv1AsyncStaged*>(group[0]->st.get());
        if (group.size() == 1) {
            first->decode(stream);
        } else {
            std::vector<Msv1AsyncStaged*> behind;
            for (size_t i = 1; i < group.size(); ++i) behind.push_back(static_cast<Msv1AsyncStaged*>(group[i]->st.get()));
            first->decode_with(behind, stream);
            paired_frames += (long long)group.size();
        }
        for (jsp_async_job* j : group) JSP_HIP(hipEventRecord(j->done, stream));
    }
    bool async_launch(jsp_async_job& j) override {
        auto* st = dynamic_cast<Msv1AsyncStaged*>(j.st.get());
        const int k = frames_per_launch();
        if (!st || !st->merged || !opt_async_pairs || k < 2) { launch_held(); return false; }
        if (!held.empty() && static_cast<Msv1AsyncStaged*>(held[0]->st.get())->small_tiles != st->small_tiles) launch_held();
        held.push_back(&j);
        if ((int)held.size() >= k) launch_held();
        return true;
    }
    void async_flush(const jsp_async_job* only_if_held) override {
        if (held.empty()) return;
        if (!only_if_held || std::find(held.begin(), held.end(), only_if_held) != held.end()) launch_held();
    }
    void worker_drain() override { launch_held(); }   // (every call that needs the stream's work queued, or waits for it, comes through here)

    Msv1Codec(int bits, int w, int h, const uint8_t* pal, int pal_bytes) {
        kind = bits == 16 ? JSP_CODEC_MSVIDEO1_16 : JSP_CODEC_MSVIDEO1_8;
        X = w;
        Y = h;
        geo.bits = bits;
        geo.X = w;
        geo.Y = h;
        geo.nbx = w >> 2;
        geo.nby = h >> 2;
        geo.nblocks = geo.nbx * geo.nby;
        size_of_just_skips = (size_t)(geo.nblocks / 1023) * 2 + 10;  // MSVideo1.hx:29-30
        block_changes.assign(std::max(geo.nby, 0), 0);
        std::memset(palette, 0, sizeof palette);
        if (pal && pal_bytes > 0) palette_bytes.assign(pal, pal + pal_bytes);
    }

    int preinit(int lines) override {
        launch_held();                        // (frames held for a group launch were staged under the settings so far)
        insignificant_blocks = (lines + 3) >> 2;
        if (geo.bits == 16) {
            insign_lines = lines;
            insign_lines_set = true;
        } else {
            // readUnsignedInt() on the strf palette bytes, little-endian
            size_t pos = 0;
            int i = 0;
            while (i < 256 && palette_bytes.size() - pos >= 4) {
                uint32_t v;
                std::memcpy(&v, palette_bytes.data() + pos, 4);
                palette[i++] = (int32_t)v;
                pos += 4;
            }
            activate();
            d_palette.reserve(sizeof palette);
            JSP_HIP(hipMemcpy(d_palette.p, palette, sizeof palette, hipMemcpyHostToDevice));
        }
        return JSP_ZERO_STATE;
    }

    int is_key_frame(const uint8_t* src, size_t n) override { return msv1_is_key_frame(geo, src, n); }
    int needs_index() override { return 1; }
    bool may_leave_pixels(const jsp_frame_in&) override {
        // remainders outside the block grid are never written; an 8-bit end marker or an abort
        // can leave more.  Cheap to be conservative: only exact multiples of 4 with a 16-bit
        // stream are guaranteed to be fully written.
        // ... and while there is no previous frame a skip code aborts the frame (the reference raises), leaving
        // every block after it as the caller had it
        return (X & 3) || (Y & 3) || geo.bits == 8 || !prev_dev;
    }
    int set_option(const char* key, const char* value) override {
        if (std::strcmp(key, "msv1_parse") == 0) {
            if (std::strcmp(value, "gpu") == 0) { opt_gpu_parse = true; return 0; }
            if (std::strcmp(value, "host") == 0) { opt_gpu_parse = false; return 0; }
        }
        if (std::strcmp(key, "msv1_inject_fault") == 0) {   // tests: the next staged batch behaves as if a tile's look-back had timed out
            opt_inject_fault = std::strcmp(value, "1") == 0;
            opt_inject_deaf = std::strcmp(value, "2") == 0;   // ... or the next one-launch asynchronous frame as if its verdict wait had
            return 0;
        }
        if (std::strcmp(key, "msv1_async_pairs") == 0) {    // one-launch asynchronous frames two to a launch (on) or one by one (off)
            if (std::strcmp(value, "on") != 0 && std::strcmp(value, "off") != 0) return JSP_ERROR_OCCURED;
            launch_held();
            opt_async_pairs = std::strcmp(value, "on") == 0;
            return 0;
        }
        if (std::strcmp(key, "msv1_compact_tables") == 0) { // replays of an inter-frame batch write and read 2-byte block tables (on) or the 4-byte ones staging builds (off)
            if (std::strcmp(value, "on") != 0 && std::strcmp(value, "off") != 0) return JSP_ERROR_OCCURED;
            opt_compact_tables = std::strcmp(value, "on") == 0;
            return 0;
        }
        if (std::strcmp(key, "msv1_parse_ahead") == 0) {    // replays of an inter-frame batch: the next replay's table-writing parse beside this replay's launches (on), or in front of them (off)
            if (std::strcmp(value, "on") != 0 && std::strcmp(value, "off") != 0) return JSP_ERROR_OCCURED;
            opt_parse_ahead = std::strcmp(value, "on") == 0;
            return 0;
        }
        if (std::strcmp(key, "msv1_scrub_tables") == 0) {   // tests: a replay must rebuild every block table it reads
            opt_scrub_tables = std::strcmp(value, "1") == 0;
            return 0;
        }
        if (std::strcmp(key, "msv1_async") == 0) {   // frames of up to MSV1_MERGED_MAX_TILES tiles: one launch, or scout + decode
            if (std::strcmp(value, "auto") == 0) { opt_async_merged = true; opt_async_auto = true; return 0; }
            if (std::strcmp(value, "one_launch") == 0) { opt_async_merged = true; opt_async_auto = false; opt_async_dma = false; return 0; }
            if (std::strcmp(value, "one_launch_dma") == 0) { opt_async_merged = true; opt_async_auto = false; opt_async_dma = true; return 0; }
            if (std::strcmp(value, "two_launches") == 0) { opt_async_merged = false; return 0; }
        }
        return -1;
    }

    // Host parse of one frame into `desc`, keeping block_changes exact.
    void host_parse(const uint8_t* src, size_t n, uint32_t base, uint32_t* desc, Msv1Parse& pr) {
        if (block_changes_stale && last_full_dev) {   // (asynchronous path) fetch that frame's bytes from HBM first
            JSP_HIP(hipStreamSynchronize(stream));
            last_full_frame.resize(last_full_dev_bytes);
            if (last_full_dev_bytes) JSP_HIP(hipMemcpy(last_full_frame.data(), last_full_dev, last_full_dev_bytes, hipMemcpyDeviceToHost));
            last_full_dev = nullptr;
        }
        if (block_changes_stale) {  // replay the last GPU-parsed frame to recover the per-row flags
            std::vector<uint32_t> scratch((size_t)std::max(geo.nblocks, 1));
            Msv1Parse tmp;
            msv1_parse(geo, last_full_frame.data(), last_full_frame.size(), true, 0, insignificant_blocks, 0,
                       scratch.data(), block_changes, tmp);
            block_changes_stale = false;
        }
        msv1_parse(geo, src, n, prev_dev != nullptr, size_of_just_skips, insignificant_blocks, base, desc,
                   block_changes, pr);
    }

    // What the host can tell about a frame without parsing it: `changes` (a coded block is on the chain — the chain
    // starts at byte 0 and leading skip codes are one slot each, so the first coded block is found by walking them) and
    // whether the frame is one the asynchronous path leaves to the synchronous one.
    struct Prescan { bool sync_path, changes; };
    Prescan prescan(const uint8_t* src, size_t n) const {
        if (n < 2 || (geo.bits == 16 && n < size_of_just_skips)) return {true, false};   // early-outs / tiny frames: host parser
        long covered = 0;
        for (size_t si = 0; si + 1 < n && covered < geo.nblocks; si += 2) {
            const unsigned a = src[si], b = src[si + 1];
            if (geo.bits == 8 && a == 0 && b == 0) return {true, false};                 // end marker at the head of the chain
            if ((b & 0xFC) != 0x84) return {false, true};                                // a coded block
            const long cnt = (long)(((b - 0x84) << 8) + a);
            if (cnt == 0) return {true, false};                                          // "skip -1": everything left is copied
            covered += cnt;
        }
        return {covered < geo.nblocks, false};   // all skip codes: nothing coded; too short = host parser
    }

    // frames the scout + vetoable decode pair cannot take: they go through the synchronous staging
    bool sync_staging(const jsp_frame_in& f, const Prescan& ps) const {
        const bool aligned = (X & 3) == 0 && !(reinterpret_cast<uintptr_t>(f.dst) & 15) && !(reinterpret_cast<uintptr_t>(prev_dev) & 15);
        return !opt_gpu_parse || !aligned || (Y & 3) || ps.sync_path || geo.nblocks <= 0 || geo.nblocks >= (1 << 20) ||
               f.n + msv1_parse_tile_bytes() > 0xFFFFFFF0u || (geo.bits == 8 && !d_palette.p);
    }
    bool async_settle_first(const jsp_frame_in& f) override { return sync_staging(f, prescan(f.src, f.n)); }
    bool sync_through_async() const override { return opt_gpu_parse; }

    jsp_staged* stage_async(const jsp_frame_in& f, jsp_staged* reuse) override {
        activate();
        static const size_t small_limit = [] { const char* e = std::getenv("JSP_MSV1_SMALL_TILE_BYTES"); return e ? (size_t)std::atoll(e) : MSV1_SMALL_TILE_FRAME_BYTES; }();   // (lab)
        const bool small_tiles = f.n <= small_limit;
        const size_t tile_bytes = small_tiles ? msv1_small_tile_bytes() : msv1_parse_tile_bytes();
        const Prescan ps = prescan(f.src, f.n);
        if (sync_staging(f, ps)) {
            // the synchronous staging (it may wait for the GPU: tiny, odd or pre-parsed frames only)
            return stage(std::vector<jsp_frame_in>{f}, dynamic_cast<Msv1Staged*>(reuse));
        }
        const double t0 = now_ms();
        auto* st = dynamic_cast<Msv1AsyncStaged*>(reuse);
        std::unique_ptr<Msv1AsyncStaged> guard;
        if (!st) { st = new Msv1AsyncStaged(); guard.reset(st); }
        st->geo = geo;
        st->d_palette = static_cast<const int32_t*>(d_palette.p);
        st->insignificant_blocks = insignificant_blocks;
        st->decoded = false;
        st->why.clear();
        st->status.assign(1, JSP_ZERO_STATE);
        st->adopted.assign(1, ps.changes ? 1 : 0);
        st->significant.assign(1, 0);
        st->cleared.assign(1, 0);
        st->nbytes = f.n;
        st->have_prev = prev_dev != nullptr;
        st->key = f.key;
        if (!d_poison.p) {
            d_poison.reserve(sizeof(uint32_t));
            JSP_HIP(hipMemsetAsync(d_poison.p, 0, sizeof(uint32_t), stream));
        }
        st->d_poison = static_cast<uint32_t*>(d_poison.p);
        // inter frames with a previous frame are compared against it in any case (whether the result counts is known
        // only with the stage-1 flag the kernel reports)
        st->compare = geo.bits == 16 && insign_lines_set && !f.key && prev_dev != nullptr;
        // option "key_frame_compare": a key frame is compared with the frame before it by the very kernel that decodes it (the stage-2
        // compare of the inter frames, MSVideo1.hx:195-204, from the Manager's row on) — when the block grid covers the frame
        st->key_compare = f.key && key_compare_row >= 0 && prev_dev != nullptr && ps.changes && (X & 3) == 0 && (Y & 3) == 0;
        st->key_differs.assign(1, st->key_compare ? -3 : -2);   // -3: async_finish() says (jsp_api.cpp)
        if (st->key_compare) st->compare = true;
        const size_t n_even = f.n & ~size_t(1);
        const int nt = (int)((f.n + tile_bytes - 1) / tile_bytes);
        st->ntiles = nt;
        st->merged = opt_async_merged && nt <= MSV1_MERGED_MAX_TILES;
        st->small_tiles = small_tiles;
        st->deaf = st->merged && opt_inject_deaf;
        if (st->deaf) opt_inject_deaf = false;
        const size_t slot = (size_t)nt * tile_bytes;
        st->d_stream.reserve(slot + 64);
        st->h_info.reserve(sizeof(Msv1AsyncInfo));
        if ((size_t)nt > st->agg_tiles) {   // published tile tables carry the launch epoch: fresh memory must read as epoch 0
            st->d_agg.reserve(sizeof(unsigned long long) * 9 * (size_t)nt);
            st->agg_tiles = st->d_agg.cap / (sizeof(unsigned long long) * 9);
            JSP_HIP(hipMemsetAsync(st->d_agg.p, 0, st->d_agg.cap, stream));
        }
        if (st->merged && !st->d_report.p) {
            st->d_report.reserve(sizeof(Msv1AsyncInfo));
            JSP_HIP(hipMemsetAsync(st->d_report.p, 0, sizeof(Msv1AsyncInfo), stream));
            st->want = 0;
        }
        const size_t meta_bytes = sizeof(Msv1TileRec) * (size_t)nt + sizeof(Msv1AsyncInfo);
        if (!st->merged) {
            st->d_meta.reserve(meta_bytes);
            st->h_meta.reserve(meta_bytes);
        }
        auto* info_dev = st->d_info();
        auto fill = [&](Msv1TileRec& r, int k) {
            r.byte0 = (uint32_t)(k * tile_bytes);
            r.frame_end = (uint32_t)n_even;
            r.data_end = geo.bits == 16 ? (uint32_t)n_even : (uint32_t)f.n;
            r.k = (uint32_t)k;
            r.first_tile = 0;
            r.ntiles = (uint32_t)nt;
            r.cmp_row_lo = st->key_compare ? (uint32_t)key_compare_row : st->compare ? (uint32_t)std::max(insign_lines, 0) : 0xFFFFFFFFu;
            r.flags = 0;
            r.dst = f.dst;
            r.prev = prev_dev;
            r.signif = &info_dev->signif;
            r.pad = 0;
        };
        // the frame's bytes: on the device already when the caller had the range prefetched (jsp_prefetch); else from where they are
        // when the caller keeps them in pinned memory, else through our own
        const void* up = f.src;
        const uint8_t* up_dev = nullptr;
        hipPointerAttribute_t attr{};
        UpRange* range = range_of(f.src, f.n);
        if (range) {
            up_dev = static_cast<const uint8_t*>(range->dev.p) + range->skew + (f.src - range->host);
            if (!range->waited || range->waited_on != stream) {   // the first frame out of the range: the frames' stream waits for the range's copy once
                JSP_HIP(hipStreamWaitEvent(stream, range->up, 0));
                range->waited = true;
                range->waited_on = stream;
            }
            range->used = true;
            ++prefetched_frames;
        } else if (hipPointerGetAttributes(&attr, f.src) == hipSuccess && attr.type == hipMemoryTypeHost) {
            up_dev = static_cast<const uint8_t*>(attr.devicePointer ? attr.devicePointer : f.src);
        } else {
            (void)hipGetLastError();
            st->h_stream.reserve(f.n + 16);
            std::memcpy(st->h_stream.p, f.src, f.n);
            up = st->h_stream.p;
            up_dev = static_cast<const uint8_t*>(st->h_stream.p);
        }
        if (st->merged) {
            fill(st->rec, 0);
            st->src_dev = up_dev;
            if (!counted_async) { counted_async = true; g_async_streams.fetch_add(1); }
            // a few streams: the copy engine works next to the kernels; many: its queues become the bottleneck (16 streams on one
            // GPU: 57 against 68 Gpixels/s), the kernels then fetch the bytes themselves
            st->dma = opt_async_auto ? g_async_streams.load() <= kDmaStreams : opt_async_dma;
            if (range) st->dma = false;      // (the kernel reads the device copy of the range and leaves the frame's own copy in d_stream, as when it reads pinned memory)
            if (st->dma) {   // the copy engine brings the bytes up on a stream of its own, next to the previous frame's kernel
                hipStream_t& up_stream = up_streams[up_next++ % (unsigned)up_count];
                if (!up_stream) JSP_HIP(hipStreamCreateWithFlags(&up_stream, hipStreamNonBlocking));
                if (!st->uploaded) JSP_HIP(hipEventCreateWithFlags(&st->uploaded, hipEventDisableTiming));
                JSP_HIP(hipMemcpyAsync(st->d_stream.p, up, f.n, hipMemcpyHostToDevice, up_stream));
                JSP_HIP(hipEventRecord(st->uploaded, up_stream));
                st->src_dev = static_cast<const uint8_t*>(st->d_stream.p);
            }
        } else {
            auto* recs = static_cast<Msv1TileRec*>(st->h_meta.p);
            for (int k = 0; k < nt; ++k) fill(recs[k], k);
            std::memset(recs + nt, 0, sizeof(Msv1AsyncInfo));
            if (range) JSP_HIP(hipMemcpyAsync(st->d_stream.p, up_dev, f.n, hipMemcpyDeviceToDevice, stream));   // (larger frames: from the range's copy to the frame's own)
            else JSP_HIP(hipMemcpyAsync(st->d_stream.p, up, f.n, hipMemcpyHostToDevice, stream));
            JSP_HIP(hipMemcpyAsync(st->d_meta.p, recs, meta_bytes, hipMemcpyHostToDevice, stream));
        }
        // codec state, as the synchronous path leaves it
        if (ps.changes) prev_dev = f.dst;
        block_changes_stale = true;
        last_full_dev = st->d_stream.p;
        last_full_dev_bytes = f.n;
        st->info = jsp_staged_info{};
        st->info.frames = 1;
        st->info.pixels = (uint64_t)X * Y;
        st->info.stream_bytes = f.n;
        st->info.kernel_launches = st->merged ? 1 : 2;
        st->info.host_stage_ms = now_ms() - t0;
        st->kernels = "msv1_fused_kernel";
        guard.release();
        return st;
    }

    bool async_finish(jsp_staged* base) override {
        auto* st = dynamic_cast<Msv1AsyncStaged*>(base);
        if (!st) return true;                                    // went through the synchronous staging: already settled
        const Msv1AsyncInfo& in = *static_cast<const Msv1AsyncInfo*>(st->h_info.p);
        if (in.fault) return false;                              // a tile gave up waiting (timing, not the stream): the host path settles the frame
        if ((in.flags & (MSV1_ASYNC_SHORT | MSV1_ASYNC_END | MSV1_ASYNC_STUCK)) || ((in.flags & MSV1_ASYNC_SKIPCODE) && !st->have_prev))
            return false;                                        // the host parser has to settle this stream
        // significance, MSVideo1.hx:187-204 / 372-388 (key frames report none)
        const bool s1 = st->adopted[0] && (in.flags & MSV1_ASYNC_S1);
        int sg = 0;
        if (s1 && !st->key) {
            if (!st->have_prev) sg = 1;
            else if (st->compare) sg = in.signif ? 1 : 0;
            // 8-bit: NaN loop bound -> no pixel is compared -> false
        }
        st->significant[0] = sg;
        if (st->key_compare) st->key_differs[0] = in.signif ? 1 : 0;
        return true;
    }

    jsp_staged* stage(const std::vector<jsp_frame_in>& frames, jsp_staged* reuse) override {
        activate();
        const double t0 = now_ms();
        auto* st = dynamic_cast<Msv1Staged*>(reuse);
        if (!st) st = new Msv1Staged();
        std::unique_ptr<Msv1Staged> guard(reuse ? nullptr : st);
        const int nf = (int)frames.size();
        const size_t nblk = (size_t)std::max(geo.nblocks, 1);
        st->quiesce_side();                            // (a reused batch: nothing of its last replay still runs beside the stream)
        st->drop_sets();
        st->side = nullptr;
        if (opt_parse_ahead) {
            if (!side_stream) JSP_HIP(hipStreamCreateWithFlags(&side_stream, hipStreamNonBlocking));
            st->side = side_stream;
        }
        st->geo = geo;
        st->nframes = nf;
        st->decoded = false;
        st->groups.clear();
        st->status.assign(nf, JSP_ZERO_STATE);
        st->adopted.assign(nf, 0);
        st->significant.assign(nf, 0);
        st->info = jsp_staged_info{};
        st->why.clear();
        st->insignificant_blocks = insignificant_blocks;
        st->inject_fault = opt_inject_fault;
        st->fallback_total = lookback_fallbacks;
        // the on-GPU parse packs block counts in 20 bits
        st->gpu_parse = opt_gpu_parse && geo.nblocks > 0 && geo.nblocks < (1 << 20);
        if (geo.bits == 8 && !d_palette.p) {  // Preinit not called: all-zero palette
            d_palette.reserve(sizeof palette);
            JSP_HIP(hipMemcpy(d_palette.p, palette, sizeof palette, hipMemcpyHostToDevice));
        }
        st->d_palette = static_cast<const int32_t*>(d_palette.p);

        // ---- lay the frames out in one stream buffer (16-byte aligned starts) ----------------
        // With the on-GPU parse every frame starts on a tile boundary of the stream buffer, so tile t of the batch is the
        // bytes [t * tile, (t + 1) * tile): a workgroup of the fused kernel knows where its bytes are from its index alone.
        std::vector<size_t> beg(nf);
        size_t total_stream = 0;
        const size_t frame_align = st->gpu_parse ? (size_t)msv1_parse_tile_bytes() : 16;
        static const size_t lab_gap = [] { const char* e = std::getenv("JSP_MSV1_FRAME_GAP"); return e ? (size_t)std::atoll(e) & ~size_t(15) : size_t(0); }();   // lab: bytes left free after every frame's slot
        for (int i = 0; i < nf; ++i) {
            beg[i] = total_stream;
            total_stream += (frames[i].n + frame_align - 1) / frame_align * frame_align + (st->gpu_parse ? lab_gap : 0);
        }
        if (total_stream + 64 > 0xFFFFFFF0u) throw std::runtime_error("batch stream exceeds 4 GiB");
        // Where the frames' bytes are: a frame in pinned host memory (jsp_host_alloc, or memory the caller registered) is uploaded
        // from where it is; the others are first gathered in the batch's own pinned buffer.
        std::vector<uint8_t> in_pinned(nf, 0);
        size_t gather_bytes = 0;
        for (int i = 0; i < nf; ++i) {
            if (st->gpu_parse && frames[i].n >= 4096) {
                hipPointerAttribute_t attr{};
                if (hipPointerGetAttributes(&attr, frames[i].src) == hipSuccess && attr.type == hipMemoryTypeHost) in_pinned[i] = 1;
                else (void)hipGetLastError();      // plain malloc'd memory: not known to HIP
            }
            if (!in_pinned[i]) gather_bytes += frames[i].n;
        }
        if (gather_bytes || !st->gpu_parse) st->h_stream.reserve(total_stream + 64);
        // (host-built block tables: every frame's with the host parser, else only those of the frames the GPU cannot settle —
        // allocated when the first such frame turns up: a quarter of a gigabyte of pinned memory for 512 frames)
        if (!st->gpu_parse) st->h_desc.reserve(sizeof(uint32_t) * nblk * std::max(nf, 1));
        st->h_frames.reserve(sizeof(Msv1FrameArgs) * std::max(nf, 1));
        st->d_signif.reserve(sizeof(uint32_t) * std::max(nf, 1));
        st->h_signif.reserve(sizeof(uint32_t) * std::max(nf, 1));
        st->d_stream.reserve(total_stream + 64);
        st->d_desc.reserve(sizeof(uint32_t) * nblk * std::max(nf, 1));
        st->d_frames.reserve(sizeof(Msv1FrameArgs) * std::max(nf, 1));
        auto* h_stream = static_cast<uint8_t*>(st->h_stream.p);
        auto* h_frames = static_cast<Msv1FrameArgs*>(st->h_frames.p);
        auto* d_signif = static_cast<uint32_t*>(st->d_signif.p);
        {
            // gathered on several threads when there is much to gather: one thread copies ~6 GB/s, a batch of 512 1080p frames
            // is half a gigabyte
            auto gather = [&](int lo, int hi) {
                for (int i = lo; i < hi; ++i) {
                    if (in_pinned[i]) continue;
                    if (frames[i].n) std::memcpy(h_stream + beg[i], frames[i].src, frames[i].n);
                    const size_t padded = (frames[i].n + 15) & ~size_t(15);   // (the rest of the frame's slot is never read)
                    std::memset(h_stream + beg[i] + frames[i].n, 0, padded - frames[i].n);
                }
            };
            const int nthreads = gather_bytes > (32u << 20) ? (int)std::min<size_t>(8, (size_t)std::max(1, usable_cpus() / 2)) : 1;
            if (nthreads > 1) {
                std::vector<std::thread> pool;
                std::exception_ptr failed;
                try {
                    for (int t = 1; t < nthreads; ++t) pool.emplace_back(gather, (int)((long)nf * t / nthreads), (int)((long)nf * (t + 1) / nthreads));
                    gather(0, nf / nthreads);
                } catch (...) {
                    failed = std::current_exception();
                }
                for (auto& th : pool) th.join();
                if (failed) std::rethrow_exception(failed);
            } else {
                gather(0, nf);
            }
        }
        // the batch's stream buffer in HBM: runs of gathered frames go up in one copy each, pinned frames one by one
        auto upload_stream = [&] {
            int i = 0;
            while (i < nf) {
                if (in_pinned[i]) {
                    if (frames[i].n) JSP_HIP(hipMemcpyAsync(static_cast<uint8_t*>(st->d_stream.p) + beg[i], frames[i].src, frames[i].n, hipMemcpyHostToDevice, stream));
                    ++i;
                    continue;
                }
                int j = i;
                while (j < nf && !in_pinned[j]) ++j;
                const size_t lo = beg[i], hi = j < nf ? beg[j] : total_stream;
                if (hi > lo) JSP_HIP(hipMemcpyAsync(static_cast<uint8_t*>(st->d_stream.p) + lo, h_stream + lo, hi - lo, hipMemcpyHostToDevice, stream));
                i = j;
            }
        };
        uint32_t* h_desc = static_cast<uint32_t*>(st->h_desc.p);
        auto host_table = [&](int i) -> uint32_t* {              // frame i's block table on the host (host parser)
            if (!h_desc) {
                st->h_desc.reserve(sizeof(uint32_t) * nblk * std::max(nf, 1));
                h_desc = static_cast<uint32_t*>(st->h_desc.p);
            }
            return h_desc + (size_t)i * nblk;
        };
        // uploads are queued on the codec's stream and waited for once, where the host needs them
        double h2d_ms = 0;
        auto upload = [&](void* d, const void* h, size_t bytes) {
            if (bytes) JSP_HIP(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, stream));
        };
        auto wait_uploads = [&](double since) {
            JSP_HIP(hipStreamSynchronize(stream));
            h2d_ms += now_ms() - since;
        };

        // ---- on-GPU parse: upload the raw bytes, parse, read the per-frame counters back --------
        Msv1ParseFrame* h_pf = nullptr;
        const Msv1FrameInfo* h_info = nullptr;
        double gpu_parse_ms = 0;
        if (st->gpu_parse && nf) {
            const uint32_t tile_bytes = msv1_parse_tile_bytes();
            st->h_pframes.reserve(sizeof(Msv1ParseFrame) * nf);
            h_pf = static_cast<Msv1ParseFrame*>(st->h_pframes.p);
            int ntiles = 0, max_tiles = 1;
            for (int i = 0; i < nf; ++i) {
                // an odd trailing byte is left to the host parser (it only matters when the chain reaches it,
                // and then the stream counts as too short)
                const size_t n_even = frames[i].n & ~size_t(1);
                // the frame's slot in the stream buffer, in tiles (the last one is all padding when an odd trailing
                // byte is the only thing in it: it parses as "covers nothing")
                const int t = (int)((frames[i].n + tile_bytes - 1) / tile_bytes);
                // 16-bit early-outs (MSVideo1.hx:109-110) are settled by the host parser
                const bool pre_host = geo.bits == 16 && frames[i].n < size_of_just_skips;
                h_pf[i] = Msv1ParseFrame{(uint32_t)beg[i], (uint32_t)(beg[i] + n_even), (uint32_t)(i * nblk), (uint32_t)ntiles,
                                         (uint32_t)t, pre_host || t == 0 ? 1u : 0u, 0, 0};
                ntiles += t;
                max_tiles = std::max(max_tiles, t);
            }
            st->ntiles = ntiles;
            st->max_tiles = max_tiles;
            st->h_tile_frame.reserve(sizeof(uint32_t) * std::max(ntiles, 1));
            auto* h_tf = static_cast<uint32_t*>(st->h_tile_frame.p);
            for (int i = 0; i < nf; ++i)
                for (uint32_t k = 0; k < h_pf[i].ntiles; ++k) h_tf[h_pf[i].first_tile + k] = (uint32_t)i;
            st->d_pframes.reserve(sizeof(Msv1ParseFrame) * nf);
            st->d_tile_frame.reserve(sizeof(uint32_t) * std::max(ntiles, 1));
            st->d_tile_tab.reserve(sizeof(uint32_t) * 9 * std::max(ntiles, 1));
            st->d_tile_entry.reserve(sizeof(uint32_t) * std::max(ntiles, 1));
            st->d_tile_block0.reserve(sizeof(uint32_t) * std::max(ntiles, 1));
            st->d_info.reserve(sizeof(Msv1FrameInfo) * nf);
            st->h_info.reserve(sizeof(Msv1FrameInfo) * nf);
            const double tu = now_ms();
            upload_stream();
            upload(st->d_pframes.p, h_pf, sizeof(Msv1ParseFrame) * nf);
            upload(st->d_tile_frame.p, h_tf, sizeof(uint32_t) * ntiles);
            st->launch_parse(stream);
            JSP_HIP(hipGetLastError());
            JSP_HIP(hipMemcpyAsync(st->h_info.p, st->d_info.p, sizeof(Msv1FrameInfo) * nf, hipMemcpyDeviceToHost, stream));
            JSP_HIP(hipStreamSynchronize(stream));   // the one wait of the staging pass: counters are needed now
            gpu_parse_ms = now_ms() - tu;             // uploads + parse, not separable without more waits
            h_info = static_cast<const Msv1FrameInfo*>(st->h_info.p);
        }

        // ---- per-frame protocol decisions, in stream order ------------------------------------
        bool vec_ok = (X & 3) == 0;
        struct Attr { bool dependent, noop, special, edge; };
        std::vector<Attr> attr(nf, Attr{false, false, false, false});
        std::vector<uint64_t> frame_stream(nf, 0);   // stream bytes each frame's codes occupy
        int last_gpu_frame = -1;
        bool pframes_dirty = false;
        for (int i = 0; i < nf; ++i) {
            const jsp_frame_in& f = frames[i];
            uint32_t* desc = nullptr;
            Msv1Parse pr;
            bool from_gpu = false;
            if (st->gpu_parse && !h_pf[i].host_parsed) {
                const Msv1FrameInfo& fi = h_info[i];
                const bool needs_host = fi.total_blocks < (uint32_t)geo.nblocks ||      // stream too short
                                        (fi.flags & MSV1_INFO_END_MARKER) ||             // 8-bit end marker
                                        (fi.n_skip_codes && !prev_dev);                   // the reference raises
                if (!needs_host) {
                    pr.changes = fi.n_coded != 0;
                    pr.s1 = pr.changes && (fi.flags & MSV1_INFO_S1);
                    pr.n_coded = fi.n_coded;
                    pr.n_skipped = (uint64_t)geo.nblocks - fi.n_coded;
                    pr.consumed = std::min<uint64_t>(fi.consumed, f.n);
                    from_gpu = true;
                    last_gpu_frame = i;
                    block_changes_stale = true;
                }
            }
            if (!from_gpu) {
                if (st->gpu_parse && last_gpu_frame >= 0 && block_changes_stale) {
                    const jsp_frame_in& g = frames[last_gpu_frame];
                    last_full_frame.assign(g.src, g.src + g.n);
                    last_full_dev = nullptr;
                }
                desc = host_table(i);
                host_parse(f.src, f.n, (uint32_t)beg[i], desc, pr);
                if (pr.early_out) std::fill(desc, desc + geo.nblocks, MSV1_DESC_UNTOUCHED);
                if (st->gpu_parse) {  // this frame's table comes from the host from now on
                    h_pf[i].host_parsed = 1;
                    pframes_dirty = true;
                    upload(static_cast<uint32_t*>(st->d_desc.p) + (size_t)i * nblk, desc, sizeof(uint32_t) * geo.nblocks);
                }
            }
            Msv1FrameArgs& fa = h_frames[i];
            fa.dst = f.dst;
            fa.prev = prev_dev;
            fa.signif = d_signif + i;
            fa.stream_end = (uint32_t)(beg[i] + f.n);
            fa.desc_base = (uint32_t)((size_t)i * nblk);
            fa.cmp_row_lo = 0xFFFFFFFFu;
            fa.pad = 0;
            if ((reinterpret_cast<uintptr_t>(f.dst) & 15) || (reinterpret_cast<uintptr_t>(prev_dev) & 15))
                vec_ok = false;
            bool dependent = pr.n_skipped != 0;  // reads its predecessor
            if (pr.early_out) {
                // nothing to paint: every block untouched, the frame rides along as a no-op
            } else if (pr.aborted) {
                st->status[i] = JSP_ERROR_OCCURED;  // the reference raises out of DecompressP here
                st->why = "skip code before any frame was decoded: the reference raises here";
            } else {
                // significance, MSVideo1.hx:187-204 / 372-388
                int sg = 0;
                if (pr.s1) {
                    if (!prev_dev) sg = 1;
                    else if (geo.bits == 16 && insign_lines_set && !f.key) {
                        // stage 2 on the GPU; result lands in the frame's signif word
                        fa.cmp_row_lo = (uint32_t)std::max(insign_lines, 0);
                        sg = -1;
                        dependent = true;
                    }
                    // 8-bit: NaN loop bound -> no pixel is compared -> false
                }
                st->significant[i] = f.key ? 0 : sg;
                if (pr.changes) st->adopted[i] = 1;
            }
            if (dependent && prev_dev) fa.pad |= MSV1_FRAME_USES_PREV;
            st->info.units_coded += pr.n_coded;
            st->info.units_copied += pr.n_skipped;
            st->info.stream_bytes += pr.consumed;
            frame_stream[i] = pr.consumed;

            attr[i].dependent = dependent;
            attr[i].noop = pr.early_out;
            attr[i].special = pr.aborted || (pr.n_untouched > 0 && !pr.early_out);
            attr[i].edge = fa.cmp_row_lo != 0xFFFFFFFFu && ((X & 3) || (Y & 3));
            if (pr.early_out) fa.pad |= MSV1_FRAME_NOOP;
            if (st->adopted[i]) prev_dev = f.dst;
        }
        if (st->gpu_parse && last_gpu_frame >= 0 && block_changes_stale) {
            const jsp_frame_in& g = frames[last_gpu_frame];
            last_full_frame.assign(g.src, g.src + g.n);
                    last_full_dev = nullptr;
        }
        st->vec_ok = vec_ok;
        // ---- launch plan ---------------------------------------------------------------------
        // "special" frames (abort, partially written) run alone with the per-frame kernel.  Between
        // them: a run of frames none of which reads its predecessor is one launch with grid.y = frame;
        // a run containing inter frames is one launch of the temporal kernel (tile per workgroup,
        // frames walked in registers) when the buffers allow 16-byte rows, else one launch per frame.
        {
            int i = 0;
            while (i < nf) {
                if (attr[i].special) { st->groups.push_back({i, 1, attr[i].edge, false, false}); ++i; continue; }
                int j = i;
                bool any_dep = false;
                while (j < nf && !attr[j].special) { any_dep |= attr[j].dependent; ++j; }
                std::unordered_set<const void*> seen;
                if (any_dep && vec_ok) {
                    // every access to a tile, in whichever buffer, comes from the same workgroup in
                    // program order, so buffers may even repeat inside the group
                    bool edge = false;
                    for (int k = i; k < j; ++k) edge |= attr[k].edge;
                    st->groups.push_back({i, j - i, edge, true, false});
                } else {
                    // frames that do not read their predecessor: one launch per run of them with grid.y = frame —
                    // or, straight from the stream bytes, one fused launch per run of GPU-parsed frames
                    bool closed = true;
                    for (int k = i; k < j; ++k) {
                        const bool writes = !attr[k].noop;
                        const bool fusable = st->gpu_parse && vec_ok && !attr[k].dependent && !attr[k].noop && !h_pf[k].host_parsed;
                        if (attr[k].dependent || closed || (writes && seen.count(frames[k].dst)) || st->groups.back().fused != fusable) {
                            st->groups.push_back({k, 1, attr[k].edge, false, fusable});
                            seen.clear();
                            closed = attr[k].dependent;
                        } else {
                            st->groups.back().count++;
                        }
                        if (writes) seen.insert(frames[k].dst);
                    }
                }
                i = j;
            }
        }
        st->need_signif = false;
        for (int v : st->significant) st->need_signif |= v < 0;
        st->info.frames = nf;
        st->info.pixels = (uint64_t)X * Y * nf;
        st->info.descriptor_bytes = sizeof(uint32_t) * (uint64_t)geo.nblocks * nf + sizeof(Msv1FrameArgs) * nf;
        // A = S + 64*N_coded + 128*N_skipped
        st->info.algorithmic_bytes = st->info.stream_bytes + 64 * st->info.units_coded + 128 * st->info.units_copied;
        // What the plan launches and moves.  Fused launches read their frames' stream bytes once and write every block
        // once.  Descriptor launches read stream + table, write the blocks, and read a previous frame per skipped /
        // compared block (per-frame kernel) or once per tile (temporal launch); when the table of any of them comes from
        // the parse kernels, every replay also runs tiles + chain + emit over the whole batch (two more reads of the
        // stream, one write of the table).
        {
            st->needs_desc = false;
            st->any_fused = false;
            st->kernels.clear();
            // replays write and read the compact block table (2 bytes per block + a base per 256 blocks, msv1.h) when every launch that reads tables is a
            // temporal launch of the loader-wave kernel: the per-frame block kernel and the frame-at-a-time temporal kernel read 4-byte tables
            static const bool temporal_old = std::getenv("JSP_MSV1_TEMPORAL_OLD") != nullptr;
            bool compact = st->gpu_parse && opt_compact_tables && !temporal_old;
            for (const auto& g : st->groups) compact = compact && (g.fused || g.temporal);
            st->compact_ok = compact;
            const uint64_t table_bytes_per_frame = compact ? sizeof(uint16_t) * (uint64_t)geo.nblocks + sizeof(uint32_t) * (uint64_t)msv1_tab16_groups(geo.nblocks)
                                                           : sizeof(uint32_t) * (uint64_t)geo.nblocks;
            uint64_t moved = 0;
            for (const auto& g : st->groups) {
                uint64_t written = 0, prev_reads = 0, sbytes = 0;
                bool uses_prev = false, gpu_table = false;
                for (int k = g.first; k < g.first + g.count; ++k) {
                    if (!attr[k].noop) written += (uint64_t)geo.nblocks;
                    uses_prev |= (h_frames[k].pad & MSV1_FRAME_USES_PREV) != 0;
                    if (h_frames[k].pad & MSV1_FRAME_USES_PREV) prev_reads += (uint64_t)geo.nblocks;
                    sbytes += frame_stream[k];
                    gpu_table |= st->gpu_parse && !h_pf[k].host_parsed;
                }
                moved += sbytes + 64 * written;
                if (g.fused) {
                    st->any_fused = true;
                    st->note_kernel("msv1_fused_kernel");
                } else {
                    st->needs_desc |= gpu_table;
                    moved += table_bytes_per_frame * g.count +
                             64 * (g.temporal ? (uses_prev ? (uint64_t)geo.nblocks : 0) : prev_reads);
                    st->note_kernel(g.temporal ? "msv1_blocks_temporal_kernel" : "msv1_blocks_kernel");
                }
                if (g.edge_compare) st->note_kernel("msv1_edge_compare_kernel");
            }
            if (st->needs_desc) {   // (a replay: msv1_fused_kernel in its descriptor form reads the stream once and writes the tables)
                moved += st->info.stream_bytes + table_bytes_per_frame * nf;
                if (!st->any_fused) st->kernels = "msv1_fused_kernel" + (st->kernels.empty() ? std::string() : " + " + st->kernels);
            }
            st->info.moved_bytes = moved;
            st->info.kernel_launches = st->groups.size() + (st->needs_desc ? 1 : 0);
            if (st->any_fused || st->needs_desc) {
                st->d_agg.reserve(sizeof(unsigned long long) * (9 + 8) * (size_t)std::max(st->ntiles, 1));   // (+ 8 per tile: the lab build's phase clocks)
                st->d_sync.reserve(2 * sizeof(uint32_t) + 64);   // (+ room for the lab build's phase clocks)
                st->h_fault.reserve(sizeof(uint32_t));
                *static_cast<uint32_t*>(st->h_fault.p) = 0;
                st->epoch = 0;
                // one record per tile: what msv1_fused_kernel needs to know about it
                st->h_recs.reserve(sizeof(Msv1TileRec) * (size_t)std::max(st->ntiles, 1));
                st->d_recs.reserve(sizeof(Msv1TileRec) * (size_t)std::max(st->ntiles, 1));
                auto* recs = static_cast<Msv1TileRec*>(st->h_recs.p);
                const uint32_t tile_bytes = msv1_parse_tile_bytes();
                for (int i = 0; i < nf; ++i)
                    for (uint32_t k = 0; k < h_pf[i].ntiles; ++k) {
                        Msv1TileRec& r = recs[h_pf[i].first_tile + k];
                        r.byte0 = h_pf[i].beg + k * tile_bytes;   // == (first_tile + k) * tile_bytes: frames start on tile boundaries
                        r.frame_end = h_pf[i].end;
                        r.data_end = geo.bits == 16 ? h_pf[i].end : h_frames[i].stream_end;
                        r.k = k;
                        r.first_tile = h_pf[i].first_tile;
                        r.ntiles = h_pf[i].ntiles;
                        r.cmp_row_lo = h_frames[i].cmp_row_lo;
                        r.flags = h_pf[i].host_parsed ? MSV1_TILE_SKIP : 0u;
                        r.dst = h_frames[i].dst;
                        r.prev = h_frames[i].prev;
                        r.signif = h_frames[i].signif;
                        r.pad = 0;
                    }
                // Launch order: tile-major over the frames of a launch — tile j of every frame before tile j + 1 of any — so that
                // when a tile starts, its predecessor in the frame is long done and has published where the chain stands (the
                // kernel's one-word look-back).  A launch covers the contiguous record range of its frames; the records are
                // permuted inside that range (a record carries its own byte offset and its number in stream order).
                auto tile_major = [&](Msv1TileRec* rr, int f0, int f1) {   // frames [f0, f1)
                    if (f1 - f0 < 2) return;
                    const uint32_t t0 = h_pf[f0].first_tile, t1 = h_pf[f1 - 1].first_tile + h_pf[f1 - 1].ntiles;
                    std::vector<Msv1TileRec> tmp(rr + t0, rr + t1);
                    uint32_t maxt = 0, o = t0;
                    for (int i = f0; i < f1; ++i) maxt = std::max(maxt, h_pf[i].ntiles);
                    // ... and the frames do not march in step: frame i starts (i mod 64) rounds late, so that at any time the batch's write
                    // fronts stand at different depths of their frames instead of all at tile j.  What the memory system makes of
                    // hundreds of fronts depends on where the frames lie in physical memory; staggered, the same frames
                    // take 2 - 6 % less time whichever way they lie (one process, same buffers: profiles/archive/r03_stagger_one_process.txt).
                    const uint32_t stagger = [] { const char* e = std::getenv("JSP_MSV1_STAGGER"); return e ? (uint32_t)std::atoi(e) : 64u; }();   // (lab: read at every staging)
                    for (uint32_t j = 0; j < maxt + stagger; ++j)
                        for (int i = f0; i < f1; ++i) {
                            const uint32_t late = stagger ? (uint32_t)(i - f0) % stagger : 0u;
                            if (j >= late && j - late < h_pf[i].ntiles) rr[o++] = tmp[h_pf[i].first_tile - t0 + (j - late)];
                        }
                };
                static const int major_frames = [] { const char* e = std::getenv("JSP_MSV1_TILE_MAJOR_FRAMES"); return e ? std::atoi(e) : 0; }();   // lab: permute within runs of this many frames
                for (const auto& g : st->groups)
                    if (g.fused) {
                        if (major_frames > 0)
                            for (int f = g.first; f < g.first + g.count; f += major_frames) tile_major(recs, f, std::min(f + major_frames, g.first + g.count));
                        else
                            tile_major(recs, g.first, g.first + g.count);
                    }
                JSP_HIP(hipMemcpyAsync(st->d_recs.p, recs, sizeof(Msv1TileRec) * (size_t)st->ntiles, hipMemcpyHostToDevice, stream));
                if (st->needs_desc) {   // the same records for the descriptor form: `dst` = the frame's block table; frames whose
                                        // table nobody reads (fused groups) or that came from the host parser are skipped
                    // The table-writing form has no pixel stores to hide its parse behind: it runs in 8 KiB tiles (msv1_fused_kernel<BITS, 4, 16>: 64 VGPRs and
                    // 19 KB of LDS, eight workgroups per CU instead of four, half the serial work per tile).  Frames start on 16 KiB boundaries of the
                    // stream buffer, so a frame's 8 KiB tiles are numbered from twice its first 16 KiB tile... minus the halves that are all padding: the
                    // tiles are counted per frame, records and published words (d_agg_emit) are this form's own.
                    const uint32_t tile8 = msv1_small_tile_bytes();
                    std::vector<uint32_t> first8((size_t)nf), n8((size_t)nf);
                    uint32_t nt8 = 0;
                    for (int i = 0; i < nf; ++i) {
                        first8[i] = nt8;
                        n8[i] = h_pf[i].ntiles ? (uint32_t)((frames[i].n + tile8 - 1) / tile8) : 0u;
                        nt8 += n8[i];
                    }
                    st->ntiles_emit = (int)nt8;
                    st->h_recs_emit.reserve(sizeof(Msv1TileRec) * (size_t)std::max<uint32_t>(nt8, 1));
                    st->d_recs_emit.reserve(sizeof(Msv1TileRec) * (size_t)std::max<uint32_t>(nt8, 1));
                    st->d_agg_emit.reserve(sizeof(unsigned long long) * (9 + 8) * (size_t)std::max<uint32_t>(nt8, 1));
                    auto* er = static_cast<Msv1TileRec*>(st->h_recs_emit.p);
                    std::vector<uint8_t> in_fused(nf, 0);
                    for (const auto& g : st->groups)
                        if (g.fused) std::fill(in_fused.begin() + g.first, in_fused.begin() + g.first + g.count, 1);
                    // launch order: tile-major over the frames, frame i (i mod 64) rounds late — as for the pixel-writing form above
                    {
                        uint32_t maxt = 0, o = 0;
                        for (int i = 0; i < nf; ++i) maxt = std::max(maxt, n8[i]);
                        const uint32_t stagger = 64u;
                        for (uint32_t j = 0; j < maxt + stagger; ++j)
                            for (int i = 0; i < nf; ++i) {
                                const uint32_t late = nf > 1 ? (uint32_t)i % stagger : 0u;
                                if (j < late || j - late >= n8[i]) continue;
                                const uint32_t k = j - late;
                                Msv1TileRec& r = er[o++];
                                r = Msv1TileRec{};
                                r.byte0 = h_pf[i].beg + k * tile8;
                                r.frame_end = h_pf[i].end;
                                r.data_end = geo.bits == 16 ? h_pf[i].end : h_frames[i].stream_end;
                                r.k = k;
                                r.first_tile = first8[i];
                                r.ntiles = n8[i];
                                r.signif = h_frames[i].signif;
                                r.pad = 0;
                                r.dst = reinterpret_cast<int32_t*>(static_cast<uint32_t*>(st->d_desc.p) + (size_t)i * nblk);
                                r.prev = nullptr;
                                r.cmp_row_lo = 0xFFFFFFFFu;
                                r.flags = (h_pf[i].host_parsed || in_fused[i]) ? MSV1_TILE_SKIP : 0u;
                            }
                    }
                    JSP_HIP(hipMemcpyAsync(st->d_recs_emit.p, er, sizeof(Msv1TileRec) * (size_t)nt8, hipMemcpyHostToDevice, stream));
                    JSP_HIP(hipMemsetAsync(st->d_agg_emit.p, 0, sizeof(unsigned long long) * (9 + 8) * (size_t)std::max<uint32_t>(nt8, 1), stream));
                    st->scrub.clear();
                    if (opt_scrub_tables)
                        for (int i = 0; i < nf; ++i)
                            if (!h_pf[i].host_parsed && !in_fused[i]) st->scrub.push_back((uint32_t)i);
                }
                // published tile tables carry the launch epoch (first launch: 1), so stale words must read as epoch 0
                JSP_HIP(hipMemsetAsync(st->d_agg.p, 0, sizeof(unsigned long long) * (9 + 8) * (size_t)std::max(st->ntiles, 1), stream));
                JSP_HIP(hipMemsetAsync(st->d_sync.p, 0, 2 * sizeof(uint32_t) + 64, stream));   // the fault word
            }
        }
        st->info.host_stage_ms = now_ms() - t0 - gpu_parse_ms;

        if (nf) {
            // queued behind whatever is on the stream; jsp_staged_decode queues behind them in turn.
            // (The pinned staging buffers belong to the staged batch and outlive the copies.)
            const double tu = now_ms();
            if (!st->gpu_parse) {
                upload(st->d_stream.p, h_stream, total_stream);
                upload(st->d_desc.p, h_desc, sizeof(uint32_t) * nblk * nf);
            } else if (pframes_dirty) {
                upload(st->d_pframes.p, h_pf, sizeof(Msv1ParseFrame) * nf);
            }
            upload(st->d_frames.p, h_frames, sizeof(Msv1FrameArgs) * nf);
            if (nf > 1) wait_uploads(tu);   // batches: report the upload time; single frames skip the wait
        }
        st->info.h2d_ms = h2d_ms;
        st->info.device_parse_ms = gpu_parse_ms;
        guard.release();
        return st;
    }
};

}  // namespace
}  // namespace jsp

jsp_codec* jsp_make_msv1(int bits, int w, int h, const uint8_t* palette, int palette_bytes) {
    return new jsp::Msv1Codec(bits, w, h, palette, palette_bytes);
}
