// TEST INFRASTRUCTURE: the call sequences of tests/test_async_gpu.py and examples/jsp_play.cpp against the ThreadSanitizer build of the product's
// host layers (tools/tsan_cpu.sh; stub HIP runtime and kernels in this directory: no GPU, nothing is painted).  Every scenario runs on several
// host threads at once, each with codecs of its own — the layers under test are the ones that start threads or share state between codecs:
// sp_codec.cpp's worker groups and shared host-thread budget, msv1_codec.cpp's asynchronous ring / held frames / prefetch ranges / re-runs,
// jsp_api.cpp's pools, option table and error string.  The run is clean when ThreadSanitizer prints nothing and the driver ends with "tsan driver: ok".
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/jsplayer_amd.h"

struct Clip {
    int kind, w, h, bpp;
    std::vector<uint8_t> palette;
    std::vector<std::vector<uint8_t>> frames;
    std::vector<uint8_t> keys;
};

static std::vector<Clip> load(const char* path) {
    FILE* f = std::fopen(path, "rb");
    if (!f) { std::perror(path); std::exit(2); }
    auto rd = [&](void* p, size_t n) { if (n && std::fread(p, 1, n, f) != n) { std::fprintf(stderr, "short read\n"); std::exit(2); } };
    uint32_t n = 0;
    rd(&n, 4);
    std::vector<Clip> clips(n);
    for (auto& c : clips) {
        int32_t hdr[6];
        rd(hdr, sizeof hdr);
        c.kind = hdr[0]; c.w = hdr[1]; c.h = hdr[2]; c.bpp = hdr[3];
        c.palette.resize(hdr[4]);
        rd(c.palette.data(), c.palette.size());
        for (int i = 0; i < hdr[5]; ++i) {
            uint8_t key; uint32_t len;
            rd(&key, 1); rd(&len, 4);
            c.keys.push_back(key);
            c.frames.emplace_back(len);
            rd(c.frames.back().data(), len);
        }
    }
    std::fclose(f);
    return clips;
}

static std::atomic<int> g_failures{0};
#define EXPECT(x) do { if (!(x)) { std::fprintf(stderr, "driver: %s failed at line %d (%s)\n", #x, __LINE__, jsp_last_error()); ++g_failures; } } while (0)

static jsp_codec* make(const Clip& c) {
    jsp_codec* k = jsp_codec_create(c.kind, c.w, c.h, c.bpp, c.palette.empty() ? nullptr : c.palette.data(), (int)c.palette.size(), 0);
    if (k) jsp_preinit(k, 36);
    return k;
}

// one stream played through the asynchronous calls, `depth` frames in flight, frames waited for out of phase, synchronous calls in between
static void play_async(const Clip& c, int passes, int depth, const char* opt_key, const char* opt_val, bool prefetch) {
    jsp_codec* k = make(c);
    EXPECT(k != nullptr);
    if (!k) return;
    if (opt_key) EXPECT(jsp_set_option(k, opt_key, opt_val) == 0);
    EXPECT(jsp_set_option(k, "async_depth", std::to_string(depth + 4).c_str()) == 0);
    jsp_pool* pool = jsp_pool_create(0, c.w, c.h, depth + 2);
    EXPECT(pool != nullptr);
    // the file in pinned memory, frames back to back (what jsp_prefetch ranges are cut from)
    size_t total = 0;
    for (auto& f : c.frames) total += (f.size() + 15) & ~size_t(15);
    uint8_t* file = static_cast<uint8_t*>(jsp_host_alloc(total + 16));
    std::vector<size_t> at;
    size_t off = 0;
    for (auto& f : c.frames) { at.push_back(off); std::memcpy(file + off, f.data(), f.size()); off += (f.size() + 15) & ~size_t(15); }
    std::vector<uint64_t> tickets;
    int slot = 0;
    for (int pass = 0; pass < passes; ++pass) {
        for (size_t i = 0; i < c.frames.size(); ++i) {
            if (prefetch && i % 5 == 0) {                           // the next stretch of the file; older ranges are given up while frames from them are in flight
                const size_t last = std::min(c.frames.size(), i + 5) - 1;
                (void)jsp_prefetch(k, file + at[i], at[last] + c.frames[last].size() - at[i]);
            }
            int32_t* dst = jsp_pool_buffer(pool, slot);
            if (dst == jsp_previous_frame(k)) { slot = (slot + 1) % (depth + 2); dst = jsp_pool_buffer(pool, slot); }
            slot = (slot + 1) % (depth + 2);
            uint64_t t = 0;
            const int rc = c.keys[i] ? jsp_decompress_i_async(k, file + at[i], c.frames[i].size(), dst, &t)
                                     : jsp_decompress_p_async(k, file + at[i], c.frames[i].size(), dst, &t);
            EXPECT(rc != JSP_ERROR_OCCURED);
            tickets.push_back(t);
            // tickets are waited for in submission order, but out of phase with the submissions: the flight is let fill up, then emptied by one, two or three
            if ((int)tickets.size() >= depth) {
                for (int w = 0; w < 1 + (int)(i % 3) && !tickets.empty(); ++w) {
                    int32_t* data = nullptr; int sig = 0;
                    (void)jsp_wait(k, tickets.front(), &data, &sig);
                    tickets.erase(tickets.begin());
                }
            }
            if (i % 11 == 10) {                                     // a synchronous call in the middle of the flight
                int32_t* data = nullptr; int sig = 0;
                int32_t* d2 = jsp_pool_buffer(pool, slot);
                if (d2 == jsp_previous_frame(k)) { slot = (slot + 1) % (depth + 2); d2 = jsp_pool_buffer(pool, slot); }
                slot = (slot + 1) % (depth + 2);
                if (c.keys[i]) (void)jsp_decompress_i(k, file + at[i], c.frames[i].size(), d2);
                else (void)jsp_decompress_p(k, file + at[i], c.frames[i].size(), d2, &data, &sig);
            }
        }
        if (pass % 2 == 0) {                                        // drain by jsp_sync, tickets left unwaited
            EXPECT(jsp_sync(k) != JSP_ERROR_OCCURED);
        }
        for (uint64_t t : tickets) { int32_t* data = nullptr; int sig = 0; (void)jsp_wait(k, t, &data, &sig); }
        tickets.clear();
        (void)jsp_counter(k, "async_reruns");
    }
    jsp_codec_destroy(k);
    jsp_pool_destroy(pool);
    jsp_host_free(file);
}

// a staged batch decoded a few times while other threads do the same with codecs of their own (ScreenPressor: the host stage's thread budget is shared)
static void play_staged(const Clip& c, int rounds, const char* threads) {
    jsp_codec* k = make(c);
    EXPECT(k != nullptr);
    if (!k) return;
    if (c.kind == JSP_CODEC_SCREENPRESSOR && threads) EXPECT(jsp_set_option(k, "sp_host_threads", threads) == 0);
    const int n = (int)c.frames.size();
    jsp_pool* pool = jsp_pool_create(0, c.w, c.h, n);
    EXPECT(pool != nullptr);
    std::vector<const uint8_t*> srcs;
    std::vector<size_t> lens;
    std::vector<int32_t*> dsts;
    for (int i = 0; i < n; ++i) { srcs.push_back(c.frames[i].data()); lens.push_back(c.frames[i].size()); dsts.push_back(jsp_pool_buffer(pool, i)); }
    jsp_staged* st = nullptr;
    for (int r = 0; r < rounds; ++r) {
        st = st ? jsp_restage_batch(k, st, n, srcs.data(), lens.data(), c.keys.data(), dsts.data()) : jsp_stage_batch(k, n, srcs.data(), lens.data(), c.keys.data(), dsts.data());
        EXPECT(st != nullptr);
        if (!st) break;
        EXPECT(jsp_staged_decode(k, st) != JSP_ERROR_OCCURED);
        EXPECT(jsp_sync(k) != JSP_ERROR_OCCURED);
        std::vector<int> status(n), adopted(n), sig(n);
        (void)jsp_staged_results(st, status.data(), adopted.data(), sig.data());
        jsp_staged_info info;
        (void)jsp_staged_get_info(st, &info);
    }
    if (st) jsp_staged_destroy(st);
    jsp_codec_destroy(k);
    jsp_pool_destroy(pool);
}

// pools created and destroyed side by side (a pool of 32 frames or more is probed: candidates, the process-wide form hint)
static void churn_pools(int rounds) {
    for (int r = 0; r < rounds; ++r) {
        jsp_pool* a = jsp_pool_create(0, 64, 48, 3 + r % 4);
        jsp_pool* b = jsp_pool_create(0, 64, 48, 32 + r % 3);
        EXPECT(a && b);
        int tried = 0;
        if (b) { (void)jsp_pool_store_rate(b, &tried); double ms; uint64_t held, lim; (void)jsp_pool_probe_info(b, &ms, &held, &lim); }
        if (a) jsp_pool_destroy(a);
        if (b) jsp_pool_destroy(b);
        // errors raised on purpose: the error string is per thread
        EXPECT(jsp_codec_create(99, 64, 48, 24, nullptr, 0, 0) == nullptr);
        (void)jsp_last_error();
    }
}

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: driver <clips file> [passes]\n"); return 2; }
    const std::vector<Clip> clips = load(argv[1]);
    const int passes = argc > 2 ? std::atoi(argv[2]) : 3;
    std::vector<std::thread> threads;
    for (int rep = 0; rep < 2; ++rep)
        for (const Clip& c : clips) {
            const bool sp = c.kind == JSP_CODEC_SCREENPRESSOR;
            threads.emplace_back(play_async, std::cref(c), passes, 8, sp ? "sp_async_threads" : "msv1_async", sp ? (rep ? "4" : "1") : (rep ? "one_launch" : "auto"), !sp);
            threads.emplace_back(play_async, std::cref(c), passes, 3, sp ? nullptr : "msv1_async_pairs", sp ? nullptr : (rep ? "off" : "on"), false);
            threads.emplace_back(play_staged, std::cref(c), passes, rep ? "4" : "1");
        }
    threads.emplace_back(churn_pools, 4 * passes);
    threads.emplace_back(churn_pools, 4 * passes);
    for (auto& t : threads) t.join();
    if (g_failures.load()) { std::fprintf(stderr, "tsan driver: %d expectation(s) failed\n", g_failures.load()); return 1; }
    std::printf("tsan driver: ok (%zu threads, %d passes)\n", threads.size(), passes);
    return 0;
}
