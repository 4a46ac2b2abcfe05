// extern "C" surface of libjsplayer_amd.so (declared in include/jsplayer_amd.h).
#include <algorithm>
#include <numeric>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <mutex>

#include "codec.h"

namespace jsp {
std::string& last_error_slot() {
    thread_local std::string s;
    return s;
}
void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error_slot() = buf;
}
}  // namespace jsp

using namespace jsp;

// ---- jsp_codec common parts ---------------------------------------------------------------

jsp_codec::~jsp_codec() {
    if (next_ticket != oldest_ticket) {   // frames still in flight (never waited for): their kernels write the buffers freed below
        (void)hipSetDevice(device);
        (void)hipDeviceSynchronize();
    }
    for (auto& j : jobs) {
        j.st.reset();
        if (j.done) (void)hipEventDestroy(j.done);
    }
    scratch.reset();
    if (side_stream) { (void)hipStreamSynchronize(side_stream); (void)hipStreamDestroy(side_stream); }
    if (own_stream) (void)hipStreamDestroy(own_stream);
}

void jsp_codec::init_device(int device_id) {
    int count = 0;
    JSP_HIP(hipGetDeviceCount(&count));
    if (count <= 0) throw std::runtime_error("no HIP device visible: the HIP path is mandatory, there is no CPU fallback");
    if (device_id < 0 || device_id >= count) throw std::runtime_error("device_id out of range");
    device = device_id;
    JSP_HIP(hipSetDevice(device));
    // a BLOCKING stream: ordered after whatever the caller queued on the legacy default stream (e.g. the fill that
    // initialised a freshly allocated frame buffer) — a caller that fills and decodes back to back needs no sync of
    // its own.  Distinct codec instances still overlap each other.  (jsp_set_stream replaces it.)
    JSP_HIP(hipStreamCreateWithFlags(&own_stream, hipStreamDefault));
    stream = own_stream;
}

void jsp_codec::activate() { JSP_HIP(hipSetDevice(device)); }

void jsp_staged::finish_results() noexcept {
    if (!decoded) return;
    try {
        if (device >= 0) JSP_HIP(hipSetDevice(device));   // (a re-run launches kernels: on the batch's device, whatever the caller's thread had current)
        after_sync();
    } catch (const std::exception& e) {
        status.assign(status.size(), JSP_ERROR_OCCURED);
        adopted.assign(adopted.size(), 0);
        for (int& s : significant) if (s < 0) s = 0;
        why = e.what();
        set_error("%s", e.what());
        return;
    }
    const auto* words = static_cast<const uint32_t*>(h_signif.p);
    for (size_t i = 0; i < significant.size(); ++i)
        if (significant[i] < 0) significant[i] = words[i] ? 1 : 0;
}

namespace jsp {
// Device memory put together by hand: one address range, backed by physical allocations (hipMemCreate) mapped into it.  What hipMalloc gives for
// a multi-gigabyte request and what it gives for a run of smaller ones differ, board by board, in what the decode kernels' store shapes get from them
// (5.6 - 7.0 TB/s); memory made this way took 6.8 - 7.0 TB/s from all three shapes in every arrangement on every board it was tried on
// (profiles/r06_vmm_pool_board*.txt), holds exactly the pool and is set up in under a millisecond.
struct MappedRange {
    void* va = nullptr;
    size_t bytes = 0, handle_bytes = 0, mapped = 0;      // (mapped: how many of `handles` are mapped, in order from the range's start)
    std::vector<hipMemGenericAllocationHandle_t> handles;
    bool empty() const { return va == nullptr; }
    void release() {
        if (va) {
            for (size_t h = 0; h < mapped; ++h) (void)hipMemUnmap(static_cast<char*>(va) + h * handle_bytes, handle_bytes);   // piece by piece: a range only partly mapped (a failed make) unmaps what it has
            for (auto h : handles) (void)hipMemRelease(h);
            (void)hipMemAddressFree(va, bytes);
            (void)hipGetLastError();
        }
        va = nullptr; bytes = handle_bytes = mapped = 0; handles.clear();
    }
    // `nbuf` frames of `frame_bytes`, `per` to a physical allocation; `dealt`: frame i and i + 1 never share one (frame i lies in allocation i mod n),
    // else the frames lie in order.  False (and nothing held) when the device or the runtime does not do this.
    bool make(int device, size_t frame_bytes, int nbuf, std::vector<int32_t*>& frames, int per = 16, bool dealt = true) {
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = device;
        size_t gran = 0;
        if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || gran == 0) { (void)hipGetLastError(); return false; }
        const int kPer = per < 1 ? 1 : per;
        const size_t align = std::max<size_t>(gran, (size_t)2 << 20);
        const size_t stride = (frame_bytes + gran - 1) / gran * gran;
        handle_bytes = (stride * kPer + align - 1) / align * align;
        const size_t nh = ((size_t)nbuf + kPer - 1) / kPer;
        bytes = nh * handle_bytes;
        if (hipMemAddressReserve(&va, bytes, align, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); va = nullptr; bytes = handle_bytes = 0; return false; }
        bool ok = true;
        for (size_t h = 0; h < nh && ok; ++h) {
            hipMemGenericAllocationHandle_t handle;
            ok = hipMemCreate(&handle, handle_bytes, &prop, 0) == hipSuccess;
            if (ok) {
                handles.push_back(handle);
                ok = hipMemMap(static_cast<char*>(va) + h * handle_bytes, handle_bytes, 0, handle, 0) == hipSuccess;
                if (ok) ++mapped;
            }
        }
        hipMemAccessDesc acc{};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        ok = ok && hipMemSetAccess(va, bytes, &acc, 1) == hipSuccess;
        // (the rest of the library — and the caller's torch — must see these addresses as device memory)
        hipPointerAttribute_t at{};
        ok = ok && hipPointerGetAttributes(&at, va) == hipSuccess && at.type == hipMemoryTypeDevice;
        if (!ok) {
            (void)hipGetLastError();
            release();
            return false;
        }
        frames.clear();
        for (int i = 0; i < nbuf; ++i) {
            const size_t h = dealt ? (size_t)i % nh : (size_t)i / (size_t)kPer, slot = dealt ? (size_t)i / nh : (size_t)i % (size_t)kPer;
            frames.push_back(reinterpret_cast<int32_t*>(static_cast<char*>(va) + h * handle_bytes + slot * stride));
        }
        return true;
    }
};
}  // namespace jsp
struct jsp_pool {
    int device = 0;
    int X = 0, Y = 0;
    std::vector<int32_t*> bufs;
    std::vector<void*> allocs;     // what to free: one allocation per frame, or one for all of them (bufs point into it)
    jsp::MappedRange mapped;       // ... or one address range over physical allocations of the pool's own making (the first form tried)
    double store_rate = 0;         // GB/s the chosen slab took from the probe (0: not probed)
    int attempts = 0;              // allocations tried
    double fill_rate = 0;          // GB/s of a plain fill over the first candidate: what the probe is held against
    std::vector<double> tried;     // ... and what each of them took
    double probe_ms = 0;           // wall time of the placement probe (allocations, launches, releases)
    uint64_t held_peak = 0;        // most device memory the probe held at one time, candidates kept while asking for the next
    uint64_t hold_limit = 0;       // ... and what it was allowed to hold
};
namespace jsp {
double pool_store_rate(uint32_t* const* d_frames, int nframes, int X, int Y, uint32_t fill);
double pool_fill_rate(uint32_t* slab, size_t nbytes);
void launch_frames_differ(const int32_t* a, const int32_t* b, size_t first_pixel, size_t npixels, uint32_t* d_flag, hipStream_t stream);
}
void jsp_codec::queue_key_compare(const int32_t* dst, const int32_t* prev, int slot) {
    uint32_t* d = static_cast<uint32_t*>(d_keyflag.p) + slot;
    jsp::launch_frames_differ(dst, prev, (size_t)key_compare_row * (size_t)X, (size_t)X * (size_t)Y, d, stream);
    JSP_HIP(hipMemcpyAsync(static_cast<uint32_t*>(h_keyflag.p) + slot, d, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
}

namespace {

// 1 = device memory, 2 = host memory
int classify_pointer(const void* p) {
    hipPointerAttribute_t attr{};
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // plain malloc'd memory: not known to HIP
        return 2;
    }
    return (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged) ? 1 : 2;
}

template <class F>
int guarded(F&& f, int on_error = JSP_ERROR_OCCURED) {
    try {
        return f();
    } catch (const std::exception& e) {
        set_error("%s", e.what());
        return on_error;
    }
}

// Shared body of DecompressI / DecompressP.
int decompress_one(jsp_codec* c, const uint8_t* src, size_t n, int32_t* dst, bool key, int32_t** data_pnt,
                   int* significant) {
    if (!c || !dst || (!src && n)) {
        set_error("null argument");
        return JSP_ERROR_OCCURED;
    }
    c->activate();
    c->worker_drain();
    const int mode = classify_pointer(dst);
    if (c->ptr_mode == 0) c->ptr_mode = mode;
    if (c->ptr_mode != mode) {
        set_error("host and device frame buffers mixed on one codec instance");
        return JSP_ERROR_OCCURED;
    }
    const size_t npx = (size_t)c->X * c->Y;
    jsp_frame_in f{src, n, key, dst};
    if (mode == 2) {
        for (auto& b : c->compat) b.reserve(npx * sizeof(int32_t) + 16);
        int32_t* d0 = static_cast<int32_t*>(c->compat[0].p);
        int32_t* d1 = static_cast<int32_t*>(c->compat[1].p);
        f.dst = (c->prev_dev == d0) ? d1 : d0;
        f.caller_host_dst = dst;
        if (c->may_leave_pixels(f))
            JSP_HIP(hipMemcpyAsync(f.dst, dst, npx * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    }
    std::vector<jsp_frame_in> frames{f};
    const int32_t* prev_before = c->prev_dev;          // what a key frame is compared with (Manager.hx:470, 499-504)
    jsp_staged* st = c->stage(frames, c->scratch.get());
    st->device = c->device;
    if (st != c->scratch.get()) c->scratch.reset(st);
    st->decode(c->stream);
    int key_known = -1;
    bool key_queued = false;
    if (key && c->key_compare_row >= 0 && st->status[0] == JSP_ZERO_STATE && st->adopted[0] && prev_before) {
        key_known = st->key_differs.empty() ? -2 : st->key_differs[0];
        if (key_known == -2) { c->queue_key_compare(f.dst, prev_before, 0); key_queued = true; }
    }
    // the reference paints dst in place, adopted or not: hand back whatever was written
    if (mode == 2 && (st->info.units_coded || st->info.units_copied))
        JSP_HIP(hipMemcpyAsync(dst, f.dst, npx * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    JSP_HIP(hipStreamSynchronize(c->stream));
    st->finish_results();
    if (key) c->last_key_differs = key_queued ? c->read_key_compare(0) : key_known;
    if (!st->cleared.empty() && st->cleared[0]) c->prev_caller = nullptr;
    if (st->adopted[0]) c->prev_caller = dst;
    if (data_pnt) *data_pnt = c->prev_caller;
    if (significant) *significant = st->significant[0];
    if (st->status[0] != JSP_ZERO_STATE)
        set_error("%s", st->why.empty() ? "decode aborted: the reference raises on this stream" : st->why.c_str());
    return st->status[0];
}

}  // namespace

extern "C" {

const char* jsp_last_error(void) { return last_error_slot().c_str(); }
const char* jsp_version(void) { return "jsplayer_amd 0.1 gfx950"; }

jsp_codec* jsp_codec_create(int kind, int width, int height, int bpp, const uint8_t* palette,
                            int palette_bytes, int device_id) {
    try {
        if (width <= 0 || height <= 0) throw std::runtime_error("bad frame size");
        if ((uint64_t)width * height > (1ull << 28)) throw std::runtime_error("frame too large");
        std::unique_ptr<jsp_codec> c;
        switch (kind) {
            case JSP_CODEC_MSVIDEO1_16: c.reset(jsp_make_msv1(16, width, height, nullptr, 0)); break;
            case JSP_CODEC_MSVIDEO1_8: c.reset(jsp_make_msv1(8, width, height, palette, palette_bytes)); break;
            case JSP_CODEC_SCREENPRESSOR: c.reset(jsp_make_screenpressor(width, height, bpp)); break;
            default: throw std::runtime_error("unknown codec kind");
        }
        c->init_device(device_id);
        return c.release();
    } catch (const std::exception& e) {
        set_error("%s", e.what());
        return nullptr;
    }
}

void jsp_codec_destroy(jsp_codec* c) {
    if (!c) return;
    try {
        c->activate();
        c->worker_drain();
        (void)hipStreamSynchronize(c->stream);
        // frames still in flight (never waited for) on other streams of this codec: their kernels write buffers that the
        // derived class's members own — wait here, before any destructor runs
        if (c->next_ticket != c->oldest_ticket) (void)hipDeviceSynchronize();
    } catch (...) {
    }
    delete c;
}

int jsp_preinit(jsp_codec* c, int lines) {
    if (!c) { set_error("null codec"); return JSP_ERROR_OCCURED; }
    return guarded([&] { return c->preinit(lines); });
}

int32_t* jsp_previous_frame(jsp_codec* c) { return c ? c->prev_caller : nullptr; }

int jsp_is_key_frame(jsp_codec* c, const uint8_t* src, size_t n) {
    if (!c || (!src && n)) return 0;
    return guarded([&] { return c->is_key_frame(src, n); }, 0);
}

int jsp_state(jsp_codec*) { return JSP_ZERO_STATE; }
int jsp_continue_i(jsp_codec*) { return JSP_ZERO_STATE; }
int jsp_needs_index(jsp_codec* c) { return c ? c->needs_index() : 0; }

// (defined with the asynchronous path below) the synchronous call as submit + wait, when the codec says that serves it better
bool sync_call_takes_async_path(jsp_codec* c, const int32_t* dst);
int submit_and_wait(jsp_codec* c, const uint8_t* src, size_t n, int32_t* dst, bool key, int32_t** data_pnt, int* significant);

int jsp_decompress_i(jsp_codec* c, const uint8_t* src, size_t n, int32_t* dst) {
    return guarded([&] {
        if (sync_call_takes_async_path(c, dst)) return submit_and_wait(c, src, n, dst, true, nullptr, nullptr);
        return decompress_one(c, src, n, dst, true, nullptr, nullptr);
    });
}

int jsp_decompress_p(jsp_codec* c, const uint8_t* src, size_t n, int32_t* dst, int32_t** data_pnt,
                     int* significant_changes) {
    if (data_pnt) *data_pnt = c ? c->prev_caller : nullptr;
    if (significant_changes) *significant_changes = 0;
    return guarded([&] {
        if (sync_call_takes_async_path(c, dst)) return submit_and_wait(c, src, n, dst, false, data_pnt, significant_changes);
        return decompress_one(c, src, n, dst, false, data_pnt, significant_changes);
    });
}

// ---- pool ---------------------------------------------------------------------------------

namespace {
// Which of the older forms won the last probe of this process (-1: a chunked candidate, or nothing yet): boards differ in which form their memory likes,
// a board does not change its mind between two pools — the next pool tries that form first instead of finding it again behind seven others.
// (Per device: a process that shards streams over several GPUs, jsp_shard.cpp, has as many boards as devices.)
constexpr int kHintDevices = 64;
std::atomic<int> g_pool_form_hint[kHintDevices];
struct HintInit { HintInit() { for (auto& h : g_pool_form_hint) h.store(-1); } } g_hint_init;
std::atomic<int>* pool_form_hint(int device) { return &g_pool_form_hint[device >= 0 && device < kHintDevices ? device : 0]; }
}  // namespace

jsp_pool* jsp_pool_create(int device_id, int width, int height, int nbuf) {
    try {
        if (width <= 0 || height <= 0 || nbuf <= 0) throw std::runtime_error("bad pool shape");
        int count = 0;
        JSP_HIP(hipGetDeviceCount(&count));
        if (device_id < 0 || device_id >= count) throw std::runtime_error("device_id out of range");
        JSP_HIP(hipSetDevice(device_id));
        auto p = std::make_unique<jsp_pool>();
        p->device = device_id;
        p->X = width;
        p->Y = height;
        const size_t bytes = (size_t)width * height * sizeof(int32_t);
        // A pool large enough for batches (the staged-batch calls write tile j of EVERY frame at about the same time: as many write
        // fronts as frames) is probed: the same store shape gets 5.4 - 7.0 TB/s from one set of allocations or another of the same
        // process, persistently — a property of where the frames lie in physical memory, relative to each other, that no query
        // reveals (profiles/archive/r03_fused_notes.txt, tools/front_lab.hip): sometimes one allocation per frame is the fast form and one
        // allocation for all the slow one, sometimes the other way round, sometimes the second try of the same form.  So the pool
        // measures what it was given (a few milliseconds per candidate), going round three forms (two frames per allocation, one
        // allocation for all, one per frame), keeps slow candidates
        // allocated while it asks for the next (else the allocator hands the same pages back) and settles for the first that takes
        // what a plain fill of the same memory takes, or the best of sixteen.  JSP_POOL_PROBE=0: one allocation per frame, first come.
        // What the probe may hold is bounded: at most JSP_POOL_PROBE_MAX candidates (default 16) and at most a quarter of the device memory that
        // was free when it began (JSP_POOL_PROBE_HOLD_GB: another limit, in GB) — a caller with 8 such pools to make must not find the
        // seventh refused because the sixth was still holding 68 GB of rejects.
        constexpr int kProbeFrom = 32;
        int kCandidates = 16;
        if (const char* m = std::getenv("JSP_POOL_PROBE_MAX")) kCandidates = std::max(1, std::min(64, std::atoi(m)));
        const char* env = std::getenv("JSP_POOL_PROBE");
        const bool probe = nbuf >= kProbeFrom && (width & 3) == 0 && (height & 3) == 0 && (size_t)nbuf * (size_t)((width / 4) * (height / 4) + 8191) / 8192 * 256 < (1ull << 32) &&   // (the probe: one launch, fewer than 2^32 lanes)
                           !(env && std::atoi(env) == 0);
        if (probe) {
            struct Candidate { std::vector<void*> allocs; std::vector<int32_t*> frames; double rate = 0; int form = -1; jsp::MappedRange mapped; };   // form: -1 chunked or mapped, 0 .. 2 the older forms
            std::vector<Candidate> cands;
            std::vector<void*> run;                            // the run of chunk allocations behind the first candidates (spread x the pool)
            auto release = [](Candidate& c) { for (void* d : c.allocs) (void)hipFree(d); c.allocs.clear(); c.mapped.release(); };
            uint32_t** d_table = nullptr;
            int best = -1;
            double yardstick = 0;
            const auto probe_t0 = std::chrono::steady_clock::now();
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
            uint64_t hold_limit = free_b ? (uint64_t)free_b / 4 : ~0ull;
            if (const char* g = std::getenv("JSP_POOL_PROBE_HOLD_GB")) hold_limit = std::min<uint64_t>(hold_limit, (uint64_t)(std::atof(g) * 1e9));
            const uint64_t one = (uint64_t)bytes * (uint64_t)nbuf;
            hold_limit = std::max(hold_limit, one);            // (the pool itself is always allowed)
            p->hold_limit = hold_limit;
            try {
                JSP_HIP(hipMalloc(reinterpret_cast<void**>(&d_table), sizeof(uint32_t*) * (size_t)nbuf));
                // the yardstick: what a plain fill takes from this device right now.  Measured on the first mapped candidate's own memory below (a 2 GB slab
                // of its own, as until round 6, costs a quarter of a second in hipMalloc alone on most boards); the slab only when no mapped form can be made.
                auto slab_yardstick = [&] {
                    void* slab = nullptr;
                    const size_t slab_bytes = (size_t)std::min<uint64_t>(one, 2ull << 30);
                    if (hipMalloc(&slab, slab_bytes) == hipSuccess) {
                        yardstick = jsp::pool_fill_rate(static_cast<uint32_t*>(slab), slab_bytes);
                        (void)hipFree(slab);
                    } else (void)hipGetLastError();
                };
                // Round 5 (tools/front_lab.hip; profiles/r05_front_lab_chunks.txt, r05_front_lab_spread.txt, r05_front_lab_spread_shapes.txt,
                // r05_front_lab_frame_order.txt).  Two things make a pool slow, and neither is visible to any query:
                //  (1) frames lying NEXT TO each other.  Chunks of 64 frames of slow and of fast pools, put together into one pool, take 7.1 TB/s
                //      whichever chunks they are; separately allocated chunks of 16 frames take 5.6 - 6.0 as neighbours and 7.0 when only every
                //      fourth of them is used.  Kernels whose workgroups WALK the frames (the inter-frame group kernels, the key-frame tile kernel)
                //      are hit hardest: the same 512 slots of one allocation give their store shapes 5.0 / 5.9 TB/s when frame i lies in slot i and
                //      6.3 / 6.9 when it lies in slot 17 i mod 512 — consecutive frames must not be neighbours in memory;
                //  (2) stretches of memory that are slow whatever the arrangement (5.7 against 7.0, the first 4 GB a process gets in one session).
                // So: four times the chunks the pool needs, allocated in one run; candidate k = every fourth chunk starting with the k-th, its frames
                // DEALT round-robin over the chunks (frame i and frame i + 1 in different chunks, >= 0.5 GB apart); the first candidate that takes what
                // a plain fill takes is kept, the other chunks are given back.  Only when none of the four comes near do the older forms get a try.
                auto make_old = [&](int form, Candidate& c) {   // 0: two frames per allocation, 1: all frames in one allocation, 2: an allocation per frame
                    bool ok = true;
                    if (form == 1) {
                        void* d = nullptr;
                        ok = hipMalloc(&d, bytes * (size_t)nbuf) == hipSuccess;
                        if (ok) { c.allocs.push_back(d); for (int i = 0; i < nbuf; ++i) c.frames.push_back(static_cast<int32_t*>(d) + (size_t)i * width * height); }
                    } else if (form == 0) {
                        for (int i = 0; i < nbuf && ok; i += 2) {
                            void* d = nullptr;
                            const int k = i + 1 < nbuf ? 2 : 1;
                            ok = hipMalloc(&d, bytes * k) == hipSuccess;
                            if (ok) { c.allocs.push_back(d); for (int q = 0; q < k; ++q) c.frames.push_back(static_cast<int32_t*>(d) + (size_t)q * width * height); }
                        }
                    } else {
                        for (int i = 0; i < nbuf && ok; ++i) {
                            void* d = nullptr;
                            ok = hipMalloc(&d, bytes) == hipSuccess;
                            if (ok) { c.allocs.push_back(d); c.frames.push_back(static_cast<int32_t*>(d)); }
                        }
                    }
                    if (!ok) { (void)hipGetLastError(); release(c); c.frames.clear(); return false; }
                    // consecutive frames must not be neighbours in memory (see below): frame i takes slot (i x K) mod n, K coprime to n
                    int K = 1;
                    for (int cand : {17, 19, 23, 29, 31, 37, 41, 43})
                        if (cand < nbuf && std::gcd(cand, nbuf) == 1) { K = cand; break; }
                    if (K > 1) {
                        std::vector<int32_t*> in_order(c.frames.size());
                        for (int i = 0; i < nbuf; ++i) in_order[i] = c.frames[(size_t)((long long)i * K % nbuf)];
                        c.frames.swap(in_order);
                    }
                    try {
                        JSP_HIP(hipMemcpy(d_table, c.frames.data(), sizeof(uint32_t*) * (size_t)nbuf, hipMemcpyHostToDevice));
                        c.rate = jsp::pool_store_rate(d_table, nbuf, width, height, 0u);
                    } catch (...) {
                        release(c);                            // (not among `cands` yet: nobody else would give its memory back)
                        throw;
                    }
                    c.form = form;
                    p->tried.push_back(c.rate);
                    if (std::getenv("JSP_POOL_PROBE_LOG")) std::fprintf(stderr, "[jsp_pool] candidate %d (%s): %.0f GB/s (plain fill %.0f)\n", (int)p->tried.size() - 1, form == 1 ? "one allocation" : form == 0 ? "two frames per allocation" : "one allocation per frame", c.rate, yardstick);
                    return true;
                };
                // Round 6: the pool made by hand first (MappedRange above), in up to three arrangements: sixteen frames per physical allocation and the frames
                // dealt over them; an allocation per frame; sixteen per allocation, frames in order.  Each holds the pool and nothing else, costs a few
                // milliseconds to set up and one probe launch; the best so far stays held while the next is measured (twice the pool, briefly, and only when
                // the first was not good enough).  The first that comes within 3 % of a plain fill is kept — on the boards of profiles/r06_vmm_pool_board*.txt
                // this form took 6.75 - 6.99 TB/s from all three store shapes where the best hipMalloc arrangement of the board took 6.42 - 7.03.  And when none
                // does (a board in its slow state: profiles/r06_h_bench_default_slow_board.json, every one of sixteen candidates of every kind at 5.2 - 5.8) the
                // best of the three is kept all the same: the hipMalloc forms cost half a second and a pool's worth of memory EACH to try (16 candidates, 2.2 s
                // and 55 GB held per pool in that run) and were not better there.  They are still tried when this form cannot be made at all, or on request
                // (JSP_POOL_PROBE_THOROUGH=1).  JSP_POOL_PROBE_MAPPED=0: skip the mapped forms (lab).
                bool settled = false, last_resort = false;
                {
                    const char* m = std::getenv("JSP_POOL_PROBE_MAPPED");
                    const char* th = std::getenv("JSP_POOL_PROBE_THOROUGH");
                    const bool thorough = th && std::atoi(th) != 0;
                    struct Form { int per; bool dealt; const char* what; };
                    // (the order: on the one board of five where the three differed, an allocation per frame took 7.2 TB/s and sixteen per allocation 6.3 - 6.5,
                    // profiles/r06_j_bench_default.json)
                    static const Form forms[] = {{1, true, "a physical allocation per frame"}, {16, true, "16 frames per physical allocation, frames dealt"}, {16, false, "16 frames per physical allocation, frames in order"}};
                    int best_form = -1;
                    double best_rate = 0;
                    Candidate held;
                    std::vector<Candidate> rejects;            // mapped candidates kept allocated so that the next one is made of OTHER memory
                    uint64_t rejects_bytes = 0;
                    // Phase one: the three arrangements, the best so far held while the next is measured.  Phase two, while none has come within 3 % of the plain fill:
                    // more candidates of the first arrangement, each made while the ones before it are still held — different physical memory every time.  What a pool
                    // gets from its memory is a property of WHERE that memory lies that lasts as long as the allocation does (tools/front_lab.hip LAB_TIME,
                    // profiles/r06_time_lab_memory.txt: four pools kept and re-probed through six rounds of churn and idling 6.97 / 6.68 / 5.95 / 6.68 TB/s every time,
                    // fresh pools beside them 6.4 - 6.6 / 5.4 / 5.5 by their place in the order of allocation) — unless the whole board is in its slow state
                    // (r06_time_lab_slow_state.txt: everything 5.4 - 5.8, kept or fresh).  So the best of up to nine is worth ~10 ms apiece (a millisecond to make, a
                    // probe launch) and a transient hold of up to nine pools within the hold limit; a hipMalloc candidate cost half a second.
                    // ... and within a time budget (JSP_POOL_PROBE_MS, default 250): a physical allocation is usually made in microseconds, but right after gigabytes have been
                    // given back the driver can take half a second over the next ones (a pool's search once took 1.7 s that way: profiles/r06_w_pool_probe_log.txt)
                    double budget_ms = 250.0;
                    if (const char* b = std::getenv("JSP_POOL_PROBE_MS")) budget_ms = std::max(0.0, std::atof(b));
                    auto spent_ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - probe_t0).count(); };
                    for (int k = 0; k < 9 && !(m && std::atoi(m) == 0) && (int)p->tried.size() < kCandidates && (k < 3 || spent_ms() < budget_ms); ++k) {   // (three arrangements, then at most six more of the first)
                        const bool second_phase = k >= 3;
                        if (second_phase && !(yardstick > 0 && best_rate < 0.97 * yardstick)) break;   // (unreachable: a candidate that good ended the loop; kept for the reader)
                        const Form& form = forms[second_phase ? 0 : k];
                        Candidate c;
                        if (best_form >= 0 && (uint64_t)held.mapped.bytes + rejects_bytes + one > hold_limit) break;   // (the best so far stays held while the next is measured: twice the pool, briefly)
                        // (frames far smaller than the 2 MB a physical allocation is rounded up to share one: "an allocation per frame" is an allocation per 2 MB of frames)
                        const int per = form.per == 1 ? (int)std::max<size_t>(1, ((size_t)2 << 20) / std::max<size_t>(bytes, 1)) : form.per;
                        if (!c.mapped.make(device_id, bytes, nbuf, c.frames, per, form.dealt)) break;
                        try {
                            if (yardstick <= 0) yardstick = jsp::pool_fill_rate(static_cast<uint32_t*>(c.mapped.va), (size_t)std::min<uint64_t>((uint64_t)c.mapped.bytes, 2ull << 30));
                            JSP_HIP(hipMemcpy(d_table, c.frames.data(), sizeof(uint32_t*) * (size_t)nbuf, hipMemcpyHostToDevice));
                            c.rate = jsp::pool_store_rate(d_table, nbuf, width, height, 0u);
                        } catch (...) {
                            release(c);
                            release(held);
                            for (auto& r : rejects) release(r);
                            throw;
                        }
                        p->tried.push_back(c.rate);
                        if (std::getenv("JSP_POOL_PROBE_LOG")) std::fprintf(stderr, "[jsp_pool] candidate %d (one address range over %zu physical allocations: %s%s): %.0f GB/s (plain fill %.0f)\n", (int)p->tried.size() - 1, c.mapped.handles.size(), form.what, second_phase ? ", other memory" : "", c.rate, yardstick);
                        p->held_peak = std::max<uint64_t>(p->held_peak, (uint64_t)c.mapped.bytes + (uint64_t)held.mapped.bytes + rejects_bytes);
                        const bool good = yardstick > 0 && c.rate >= 0.97 * yardstick;
                        Candidate loser;
                        if (c.rate > best_rate) { best_rate = c.rate; best_form = k; loser = std::move(held); held = std::move(c); c.mapped = jsp::MappedRange{}; }
                        else { loser = std::move(c); c.mapped = jsp::MappedRange{}; }
                        if (good) { release(loser); break; }
                        if (!loser.mapped.empty()) {           // what is rejected stays allocated until the search ends: the next candidate is then made of other memory
                            rejects_bytes += loser.mapped.bytes;
                            rejects.push_back(std::move(loser));
                            loser.mapped = jsp::MappedRange{};
                        }
                    }
                    for (auto& r : rejects) release(r);
                    if (best_form < 0) slab_yardstick();        // (no mapped form could be made: the older forms are held against a slab's fill)
                    if (best_form >= 0) {
                        cands.push_back(std::move(held));
                        held.mapped = jsp::MappedRange{};
                        best = 0;
                        // The search ends here with the best of the mapped candidates.  (On request, JSP_POOL_PROBE_THOROUGH=1, and unless that best is within 3 % of the
                        // plain fill, the hipMalloc forms of rounds 3 - 5 are tried after it.  An earlier version of this round tried one run of hipMalloc chunks as a last
                        // resort whenever nine mapped candidates stayed a tenth under the fill: on boards in their slow state — the only ones where that happens — it cost
                        // 0.4 - 1 s per pool and never found anything, profiles/r06_y_bench_default_final_slow_state_board.json.)
                        settled = !thorough || (yardstick > 0 && best_rate >= 0.97 * yardstick);
                        last_resort = false;
                    }
                }
                int hint = pool_form_hint(device_id)->load();
                if (const char* f = std::getenv("JSP_POOL_PROBE_FORM")) { const int v = std::atoi(f); if (v >= 0 && v < 3) hint = v; }   // (start with that older form: 0 two frames per allocation, 1 one allocation, 2 one per frame)
                if (!settled && !last_resort && hint >= 0 && hint < 3) {      // the form this board liked last time, next
                    Candidate c;
                    if (make_old(hint, c)) {
                        cands.push_back(std::move(c));
                        p->held_peak = std::max<uint64_t>(p->held_peak, (uint64_t)cands.size() * one);
                        if (best < 0 || cands.back().rate > cands[best].rate) best = (int)cands.size() - 1;
                        settled = yardstick > 0 && cands[best].rate >= 0.985 * yardstick;
                    }
                }
                const int kChunkFrames = 16;
                int spread = 4;
                if (settled) spread = 1;                       // (nothing more to try)
                const int nch = (nbuf + kChunkFrames - 1) / kChunkFrames;
                // (what a run of s x the pool really holds: whole chunks, so up to 15 frames more per s than s pools)
                auto run_bytes = [&](int s) { return (uint64_t)nch * (uint64_t)s * (uint64_t)kChunkFrames * (uint64_t)bytes; };
                while (spread > 1 && run_bytes(spread) + (uint64_t)cands.size() * one > hold_limit) --spread;
                auto chunk_frames = [&](int ch) { return std::min(kChunkFrames, nbuf - ch * kChunkFrames); };
                if (spread > 1) {
                    bool ok = true;
                    for (int q = 0; q < nch * spread && ok; ++q) {
                        void* d = nullptr;
                        ok = hipMalloc(&d, bytes * (size_t)kChunkFrames) == hipSuccess;   // (whole chunks all: any chunk of the run can stand for any chunk of the pool)
                        if (ok) run.push_back(d);
                    }
                    if (!ok) { (void)hipGetLastError(); for (void* d : run) (void)hipFree(d); run.clear(); }
                    else p->held_peak = std::max<uint64_t>(p->held_peak, run_bytes(spread) + (uint64_t)cands.size() * one);
                }
                // Which chunks of the run a candidate takes.  "Every fourth" is not always the answer: in some sessions all four such candidates are slow
                // (5.7 - 6.4 TB/s) while a form made of many small allocations is fast (profiles/r05_q_bench_all.jsonl: candidates_GBs) — the run's chunks
                // do not always lie in memory in the order they were asked for.  So the candidates differ in kind: every fourth from the first, a
                // pseudo-random choice, every third from the second, another pseudo-random choice.
                auto pick = [&](int k) {
                    std::vector<int> ids;
                    const int total = (int)run.size();
                    if (k == 0 || (k == 2 && spread < 3)) {
                        for (int ch = 0; ch < nch; ++ch) ids.push_back(ch * spread + (k ? 1 : 0));
                    } else if (k == 2) {
                        for (int ch = 0; ch < nch; ++ch) ids.push_back(1 + ch * 3);
                    } else {                                   // a partial Fisher-Yates shuffle, seeded by the candidate
                        std::vector<int> all(total);
                        for (int i = 0; i < total; ++i) all[i] = i;
                        uint64_t seed = 0x9E3779B97F4A7C15ull * (uint64_t)(k + 1);
                        for (int i = 0; i < nch; ++i) {
                            seed = seed * 6364136223846793005ull + 1442695040888963407ull;
                            const int j = i + (int)((seed >> 33) % (uint64_t)(total - i));
                            std::swap(all[i], all[j]);
                            ids.push_back(all[i]);
                        }
                    }
                    return ids;
                };
                int spread_best = -1, chunked = -1;            // chunked: where the chunked candidate stands among `cands`
                double spread_rate = 0;
                std::vector<int32_t*> dealt;
                auto deal = [&](const std::vector<int>& ids) {  // the candidate's frames, dealt round-robin over its chunks
                    dealt.clear();
                    for (int slot = 0; slot < kChunkFrames; ++slot)
                        for (int ch = 0; ch < nch; ++ch)
                            if (slot < chunk_frames(ch)) dealt.push_back(static_cast<int32_t*>(run[(size_t)ids[(size_t)ch]]) + (size_t)slot * width * height);
                };
                for (int k = 0; k < 4 && (int)p->tried.size() < kCandidates && !run.empty(); ++k) {
                    deal(pick(k));
                    JSP_HIP(hipMemcpy(d_table, dealt.data(), sizeof(uint32_t*) * (size_t)nbuf, hipMemcpyHostToDevice));
                    const double rate = jsp::pool_store_rate(d_table, nbuf, width, height, 0u);
                    p->tried.push_back(rate);
                    if (std::getenv("JSP_POOL_PROBE_LOG")) std::fprintf(stderr, "[jsp_pool] candidate %d (chunks of 16 frames out of a run of %d x the pool: %s; frames dealt over them): %.0f GB/s (plain fill %.0f)\n", k, spread,
                                                                        k == 0 ? "every fourth" : k == 2 ? "every third" : "a pseudo-random choice", rate, yardstick);
                    if (rate > spread_rate) { spread_rate = rate; spread_best = k; }
                    if (yardstick > 0 && rate >= 0.985 * yardstick) break;
                }
                if (spread_best >= 0) {                        // keep the best of them as a candidate like any other, give the other chunks back
                    Candidate c;
                    const std::vector<int> ids = pick(spread_best);
                    deal(ids);
                    c.frames = dealt;
                    std::vector<char> kept(run.size(), 0);
                    for (int id : ids) kept[(size_t)id] = 1;
                    for (size_t q = 0; q < run.size(); ++q) {
                        if (kept[q]) c.allocs.push_back(run[q]);
                        else (void)hipFree(run[q]);
                    }
                    run.clear();
                    c.rate = spread_rate;
                    cands.push_back(std::move(c));
                    chunked = (int)cands.size() - 1;
                    // (an older form — its frames lie densely — must beat the chunked candidate by 3 % to stand before it: the probe's shape does not mind density,
                    // the key-frame kernel's does, profiles/r05_front_lab_frame_order.txt)
                    if (best < 0 || cands[chunked].rate * (cands[best].mapped.empty() ? 1.03 : 1.0) > cands[best].rate) best = chunked;   // (against a mapped candidate: the better probe wins, no allowance)
                } else if (!run.empty()) {                     // no chunked candidate was measured (JSP_POOL_PROBE_MAX used up by the hinted form): the run goes back whole
                    for (void* d : run) (void)hipFree(d);
                    run.clear();
                }
                const bool good_enough = settled || last_resort || (chunked >= 0 && best == chunked && (yardstick <= 0 || cands[chunked].rate >= 0.95 * yardstick));
                for (int a = 0; (int)p->tried.size() < kCandidates && !good_enough; ++a) {
                    if (best >= 0 && (uint64_t)(cands.size() + 1) * one > hold_limit) break;   // holding another candidate would pass the limit
                    Candidate c;
                    const int form = (a + (hint >= 0 ? hint + 1 : 0)) % 3;    // (the hinted form has had its first try above)
                    if (!make_old(form, c)) {                  // the memory ran out while candidates were being held: the best so far it is
                        if (best >= 0) break;
                        throw std::runtime_error("out of device memory for the frame pool");
                    }
                    cands.push_back(std::move(c));
                    p->held_peak = std::max<uint64_t>(p->held_peak, (uint64_t)cands.size() * one);
                    if (best < 0 || cands.back().rate > cands[best].rate * (best == chunked ? 1.03 : 1.0)) best = (int)cands.size() - 1;
                    if (yardstick > 0 && cands[best].rate >= 0.985 * yardstick) break;   // as good as it gets (the fast kind takes what a plain fill takes)
                }
                if (best >= 0) pool_form_hint(device_id)->store(cands[best].form);
            } catch (...) {
                for (auto& c : cands) release(c);
                for (void* d : run) (void)hipFree(d);
                if (d_table) (void)hipFree(d_table);
                throw;
            }
            (void)hipFree(d_table);
            if (best < 0) throw std::runtime_error("out of device memory for the frame pool");   // (no candidate could be made at all)
            for (int i = 0; i < (int)cands.size(); ++i) if (i != best) release(cands[i]);
            p->attempts = (int)p->tried.size();
            p->store_rate = cands[best].rate;
            p->fill_rate = yardstick;
            p->allocs = cands[best].allocs;
            p->mapped = std::move(cands[best].mapped);
            cands[best].mapped = jsp::MappedRange{};
            p->bufs = cands[best].frames;
            if (!p->mapped.empty()) JSP_HIP(hipMemset(p->mapped.va, 0, p->mapped.bytes));   // (one call for the whole range: a memset per frame is a fifth of a millisecond each)
            else for (int32_t* f : p->bufs) JSP_HIP(hipMemset(f, 0, bytes));
            p->probe_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - probe_t0).count();
            return p.release();
        }
        for (int i = 0; i < nbuf; ++i) {
            void* d = nullptr;
            JSP_HIP(hipMalloc(&d, bytes));
            p->allocs.push_back(d);
            p->bufs.push_back(static_cast<int32_t*>(d));
            JSP_HIP(hipMemset(d, 0, bytes));
        }
        return p.release();
    } catch (const std::exception& e) {
        set_error("%s", e.what());
        return nullptr;
    }
}
int32_t* jsp_pool_buffer(jsp_pool* p, int i) {
    return (p && i >= 0 && i < (int)p->bufs.size()) ? p->bufs[i] : nullptr;
}
int jsp_pool_count(jsp_pool* p) { return p ? (int)p->bufs.size() : 0; }
void jsp_pool_destroy(jsp_pool* p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    for (void* d : p->allocs) (void)hipFree(d);
    p->mapped.release();
    delete p;
}
double jsp_pool_store_rate(jsp_pool* p, int* attempts) {
    if (attempts) *attempts = p ? p->attempts : 0;
    return p ? p->store_rate : 0.0;
}
int jsp_pool_probe_info(jsp_pool* p, double* probe_ms, uint64_t* held_peak_bytes, uint64_t* hold_limit_bytes) {
    if (!p) return -1;
    if (probe_ms) *probe_ms = p->probe_ms;
    if (held_peak_bytes) *held_peak_bytes = p->held_peak;
    if (hold_limit_bytes) *hold_limit_bytes = p->hold_limit;
    return 0;
}
int jsp_pool_probe_rates(jsp_pool* p, double* rates, int cap) {
    if (!p) return -1;
    for (int i = 0; rates && i < cap && i < (int)p->tried.size(); ++i) rates[i] = p->tried[(size_t)i];
    return (int)p->tried.size();
}
int jsp_download(const int32_t* device_frame, int32_t* host, size_t npixels) {
    return guarded([&] {
        JSP_HIP(hipMemcpy(host, device_frame, npixels * sizeof(int32_t), hipMemcpyDeviceToHost));
        return 0;
    });
}
int jsp_upload(int32_t* device_frame, const int32_t* host, size_t npixels) {
    return guarded([&] {
        JSP_HIP(hipMemcpy(device_frame, host, npixels * sizeof(int32_t), hipMemcpyHostToDevice));
        return 0;
    });
}

// ---- batched / staged ---------------------------------------------------------------------

int jsp_set_stream(jsp_codec* c, void* hip_stream) {
    if (!c) return JSP_ERROR_OCCURED;
    return guarded([&] {
        c->activate();
        c->async_flush(nullptr);                         // (a frame held for its successor belongs on the stream it was staged for)
        c->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
        return 0;
    });
}

int jsp_prefetch(jsp_codec* c, const void* host, size_t bytes) {
    if (!c || (!host && bytes)) return JSP_ERROR_OCCURED;
    return guarded([&] {
        c->activate();
        return c->prefetch(host, bytes) == 0 ? JSP_ZERO_STATE : JSP_ERROR_OCCURED;
    });
}

int jsp_set_option(jsp_codec* c, const char* key, const char* value) {
    if (!c || !key || !value) return -1;
    if (std::strcmp(key, "key_frame_compare") == 0) {
        if (c->next_ticket != c->oldest_ticket) return -1;
        int row = -1;
        if (std::strcmp(value, "off") != 0) {
            char* end = nullptr;
            const long v = std::strtol(value, &end, 10);
            if (end == value || *end || v < 0 || v > (1 << 24)) return -1;
            row = (int)v;
        }
        return guarded([&] {
            c->activate();
            c->worker_drain();
            if (row >= 0) { c->d_keyflag.reserve(16 * sizeof(uint32_t)); c->h_keyflag.reserve(16 * sizeof(uint32_t)); }
            c->key_compare_row = row;
            c->last_key_differs = -1;
            return 0;
        }, -1);
    }
    if (std::strcmp(key, "async_depth") == 0) {
        char* end = nullptr;
        const long v = std::strtol(value, &end, 10);
        if (end == value || *end || v < 1 || v > 16 || c->next_ticket != c->oldest_ticket) return -1;
        c->async_depth = (int)v;
        return 0;
    }
    // retired launch-plan options (measured slower, removed in round 5): results never depended on them, so a caller that still sets them is not refused
    if (std::strcmp(key, "sp_group_chunk") == 0 || std::strcmp(key, "msv1_parse_pieces") == 0) return 0;
    return guarded([&] {
        c->activate();
        return c->set_option(key, value);
    }, -1);
}

int jsp_key_frame_differs(jsp_codec* c) { return c ? c->last_key_differs : -1; }

long long jsp_counter(jsp_codec* c, const char* name) {
    if (!c || !name) return -1;
    if (std::strcmp(name, "async_reruns") == 0) return c->async_reruns;
    return c->counter(name);
}

// ---- asynchronous per-frame path ---------------------------------------------------------------------------

namespace {

// The frame of job `from` could not be settled by the GPU alone: everything from it on is re-run, in order, through
// the synchronous path (the codec's state is put back to what it was before that frame).
void redo_from(jsp_codec* c, uint64_t from) {
    c->async_flush(nullptr);                             // (frames held for their successors go out as submitted: they find the veto word set and leave their frames alone)
    JSP_HIP(hipStreamSynchronize(c->stream));
    c->async_reset();
    jsp_async_job& first = c->jobs[from % c->async_depth];
    c->prev_caller = first.prev_caller_before;
    c->prev_dev = first.prev_dev_before;
    for (uint64_t t = from; t < c->next_ticket; ++t) {
        jsp_async_job& j = c->jobs[t % c->async_depth];
        int32_t* data = nullptr;
        int sig = 0;
        j.status = decompress_one(c, j.frame.src, j.frame.n, j.frame.dst, j.frame.key, &data, &sig);
        j.significant = sig;
        j.key_differs = c->last_key_differs;
        j.key_compare_queued = false;
        if (j.status != JSP_ZERO_STATE) j.why = last_error_slot();
        j.prev_caller_after = c->prev_caller;
        j.redone = true;
        ++c->async_reruns;
    }
}

// Makes the results of job `t` final: waits for its kernels, reads the verdict of the scout, and re-runs everything from
// it on when the GPU alone could not settle the frame.
void settle(jsp_codec* c, uint64_t t) {
    jsp_async_job& j = c->jobs[t % c->async_depth];
    if (j.redone || j.settled) return;
    if (j.by_worker) {
        c->worker_wait(j);                               // host stage done, uploads and kernels queued, the event recorded
        JSP_HIP(hipEventSynchronize(j.done));
        j.st->finish_results();
        j.status = j.st->status[0];
        j.significant = j.st->significant[0] < 0 ? 0 : j.st->significant[0];
        if (j.status != JSP_ZERO_STATE) j.why = j.st->why.empty() ? "decode aborted: the reference raises on this stream" : j.st->why;
        // what the frame really did to the previous frame (frames are settled in submission order)
        if (!j.st->cleared.empty() && j.st->cleared[0]) c->settled_prev = nullptr;
        if (j.st->adopted[0]) c->settled_prev = j.frame.dst;
        if (j.key_compare_queued) { j.key_differs = c->read_key_compare((int)(t % c->async_depth)); j.key_compare_queued = false; }
        j.prev_caller_after = c->settled_prev;
        if (t + 1 == c->next_ticket) c->prev_caller = c->settled_prev;   // (nothing submitted behind it: the prediction gives way)
        j.settled = true;
        return;
    }
    c->async_flush(&j);                                  // (held for a frame that has not come: launched alone)
    JSP_HIP(hipEventSynchronize(j.done));
    j.st->finish_results();
    if (!c->async_finish(j.st.get())) { redo_from(c, t); return; }
    if (j.key_compare_queued) { j.key_differs = c->read_key_compare((int)(t % c->async_depth)); j.key_compare_queued = false; }
    else if (j.key_differs == -3) j.key_differs = j.st->key_differs.empty() ? -1 : j.st->key_differs[0];
    j.status = j.st->status[0];
    j.significant = j.st->significant[0] < 0 ? 0 : j.st->significant[0];
    if (j.status != JSP_ZERO_STATE) j.why = j.st->why.empty() ? "decode aborted: the reference raises on this stream" : j.st->why;
    j.settled = true;
}


int submit_async(jsp_codec* c, const uint8_t* src, size_t n, int32_t* dst, bool key, uint64_t* ticket) {
    if (!c || !dst || !ticket || (!src && n)) throw std::runtime_error("null argument");
    c->activate();
    if (classify_pointer(dst) != 1) throw std::runtime_error("asynchronous calls take device frame buffers only");
    if (c->ptr_mode == 2) throw std::runtime_error("codec is in host-pointer mode");
    c->ptr_mode = 1;
    if ((int)(c->next_ticket - c->oldest_ticket) >= c->async_depth) throw std::runtime_error("too many frames in flight: jsp_wait for the oldest first");
    // (the ring only ever grows: a smaller `async_depth` changes the modulus, not the vector — the jobs beyond it keep their
    // event and their staged object, whose device buffers the codec may still refer to: Msv1Codec::last_full_dev)
    if ((int)c->jobs.size() < c->async_depth) c->jobs.resize(c->async_depth);
    jsp_async_job& j = c->jobs[c->next_ticket % c->async_depth];
    if (!j.done) JSP_HIP(hipEventCreateWithFlags(&j.done, hipEventDisableTiming));
    j.frame = jsp_frame_in{src, n, key, dst};
    if (c->async_by_workers()) {
        if (c->next_ticket == c->oldest_ticket) c->settled_prev = c->prev_caller;   // nothing in flight: predictions start from the facts
        j.prev_caller_before = c->prev_caller;
        j.prev_dev_before = c->prev_dev;
        j.redone = j.settled = false;
        j.by_worker = true;
        j.key_differs = -1;
        j.key_compare_queued = false;
        j.why.clear();
        c->worker_submit(j);
        j.prev_caller_after = c->prev_caller;
        j.ticket = c->next_ticket++;
        *ticket = j.ticket;
        return JSP_ZERO_STATE;
    }
    j.by_worker = false;
    if (c->async_settle_first(j.frame))
        for (uint64_t t = c->oldest_ticket; t < c->next_ticket; ++t)
            if (c->jobs[t % c->async_depth].st->verdict_pending) settle(c, t);   // the others were settled when they were staged
    j.prev_caller_before = c->prev_caller;
    j.prev_dev_before = c->prev_dev;
    j.redone = j.settled = false;
    j.why.clear();
    j.key_differs = -1;
    j.key_compare_queued = false;
    jsp_staged* st = c->stage_async(j.frame, j.st.get());
    st->device = c->device;
    if (st != j.st.get()) j.st.reset(st);
    const bool wants_compare = key && c->key_compare_row >= 0 && st->status[0] == JSP_ZERO_STATE && st->adopted[0] && j.prev_dev_before;
    if (wants_compare) j.key_differs = st->key_differs.empty() ? -2 : st->key_differs[0];   // (-3: the frame's own kernels compare; async_finish() knows)
    // (a compare pass of the codec's own must follow the frame's kernels at once: such a frame is not handed to async_launch, which may hold it)
    if (j.key_differs == -2 && wants_compare) c->async_flush(nullptr);
    if (!(wants_compare && j.key_differs == -2) && c->async_launch(j)) {
        // the codec launches the frame — now, or together with the next one — and records j.done behind it
    } else {
        st->decode(c->stream);
        // (a frame the GPU may still veto leaves `dst` untouched and is re-run through the synchronous path, which compares again)
        if (wants_compare && j.key_differs == -2) { c->queue_key_compare(dst, j.prev_dev_before, (int)(c->next_ticket % c->async_depth)); j.key_compare_queued = true; }
        JSP_HIP(hipEventRecord(j.done, c->stream));
    }
    if (!st->cleared.empty() && st->cleared[0]) c->prev_caller = nullptr;
    if (st->adopted[0]) c->prev_caller = dst;
    j.prev_caller_after = c->prev_caller;
    j.ticket = c->next_ticket++;
    *ticket = j.ticket;
    return JSP_ZERO_STATE;
}

}  // namespace

extern "C" int jsp_decompress_i_async(jsp_codec* c, const uint8_t* src, size_t n, int32_t* dst, uint64_t* ticket) {
    return guarded([&] { return submit_async(c, src, n, dst, true, ticket); });
}
extern "C" int jsp_decompress_p_async(jsp_codec* c, const uint8_t* src, size_t n, int32_t* dst, uint64_t* ticket) {
    return guarded([&] { return submit_async(c, src, n, dst, false, ticket); });
}
namespace {
int wait_ticket(jsp_codec* c, uint64_t ticket, int32_t** data_pnt, int* significant_changes) {
    if (!c) throw std::runtime_error("null codec");
    if (ticket != c->oldest_ticket || ticket >= c->next_ticket) throw std::runtime_error("tickets are waited for in submission order");
    c->activate();
    jsp_async_job& j = c->jobs[ticket % c->async_depth];
    // from here on the ticket is consumed whatever happens: a caller may let go of the frame's `src` / `dst` exactly when
    // jsp_wait was given the oldest ticket (anything that fails while settling the frame becomes the frame's error)
    try {
        settle(c, ticket);
    } catch (const std::exception& e) {
        j.status = JSP_ERROR_OCCURED;
        j.why = e.what();
        j.settled = true;
    }
    if (j.status != JSP_ZERO_STATE) set_error("%s", j.why.c_str());
    ++c->oldest_ticket;
    j.ticket = 0;
    if (data_pnt) *data_pnt = j.prev_caller_after;
    if (significant_changes) *significant_changes = j.significant;
    if (j.frame.key) {
        c->last_key_differs = c->key_compare_row >= 0 && j.status == JSP_ZERO_STATE ? j.key_differs : -1;
        // ONE mapping, here: jsp_key_frame_differs() says 1 / 0 / -1 (nothing to compare with, or the frame failed); *significant_changes of a
        // key frame THAT DECODED says "changed" for 1 and for -1 (Manager.hx:399-411: the first frame, no previous frame, counts as a
        // change).  A frame that failed reports what the decode reported (0): its status is the news, not a change.
        if (c->key_compare_row >= 0 && significant_changes && j.status == JSP_ZERO_STATE) *significant_changes = j.key_differs != 0 ? 1 : 0;
    }
    return j.status;
}
}  // namespace
// The reference's one synchronous call per frame (Manager.hx:507,511), served by the asynchronous path: device frame buffer,
// nothing in flight, and a codec whose one-frame launch settles the frame by itself (MSVideo1 with the on-GPU parse:
// 0.18 -> 0.07 ms per 1080p key frame against staging a batch of one).
bool sync_call_takes_async_path(jsp_codec* c, const int32_t* dst) {
    return c && dst && c->sync_through_async() && c->ptr_mode != 2 && c->next_ticket == c->oldest_ticket &&
           classify_pointer(dst) == 1;
}
int submit_and_wait(jsp_codec* c, const uint8_t* src, size_t n, int32_t* dst, bool key, int32_t** data_pnt, int* significant) {
    uint64_t ticket = 0;
    submit_async(c, src, n, dst, key, &ticket);
    return wait_ticket(c, ticket, data_pnt, significant);
}
extern "C" int jsp_wait(jsp_codec* c, uint64_t ticket, int32_t** data_pnt, int* significant_changes) {
    if (data_pnt) *data_pnt = nullptr;
    if (significant_changes) *significant_changes = 0;
    return guarded([&] { return wait_ticket(c, ticket, data_pnt, significant_changes); });
}
extern "C" void* jsp_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
extern "C" void jsp_host_free(void* p) { if (p) (void)hipHostFree(p); }

int jsp_sync(jsp_codec* c) {
    if (!c) { set_error("null codec"); return JSP_ERROR_OCCURED; }
    return guarded([&] {
        c->activate();
        c->worker_drain();
        JSP_HIP(hipStreamSynchronize(c->stream));
        if (c->side_stream) JSP_HIP(hipStreamSynchronize(c->side_stream));   // (what was queued beside the launches counts as the caller's work too)
        return 0;
    });
}

namespace {
jsp_staged* stage_batch_into(jsp_codec* c, jsp_staged* reuse, int nframes, const uint8_t* const* srcs, const size_t* lens,
                             const uint8_t* is_key, int32_t* const* dsts);
}
jsp_staged* jsp_stage_batch(jsp_codec* c, int nframes, const uint8_t* const* srcs, const size_t* lens,
                            const uint8_t* is_key, int32_t* const* dsts) {
    return stage_batch_into(c, nullptr, nframes, srcs, lens, is_key, dsts);
}
jsp_staged* jsp_restage_batch(jsp_codec* c, jsp_staged* reuse, int nframes, const uint8_t* const* srcs, const size_t* lens,
                              const uint8_t* is_key, int32_t* const* dsts) {
    return stage_batch_into(c, reuse, nframes, srcs, lens, is_key, dsts);
}
namespace {
jsp_staged* stage_batch_into(jsp_codec* c, jsp_staged* reuse, int nframes, const uint8_t* const* srcs, const size_t* lens,
                             const uint8_t* is_key, int32_t* const* dsts) {
    try {
        if (!c || nframes < 0 || (nframes && (!srcs || !lens || !dsts))) throw std::runtime_error("null argument");
        c->activate();
        std::vector<jsp_frame_in> frames(nframes);
        for (int i = 0; i < nframes; ++i) {
            if (!dsts[i]) throw std::runtime_error("null dst in batch");
            if (classify_pointer(dsts[i]) != 1) throw std::runtime_error("batch entry points take device frame buffers only");
            frames[i] = jsp_frame_in{srcs[i], lens[i], is_key ? is_key[i] != 0 : true, dsts[i]};
        }
        if (c->ptr_mode == 2) throw std::runtime_error("codec is in host-pointer mode");
        if (nframes) c->ptr_mode = 1;
        c->worker_drain();
        jsp_staged* st = c->stage(frames, reuse);
        st->device = c->device;
        if (reuse && st != reuse) delete reuse;       // (a batch object of another kind: replaced)
        for (int i = 0; i < nframes; ++i) {
            if (!st->cleared.empty() && st->cleared[i]) c->prev_caller = nullptr;
            if (st->adopted[i]) c->prev_caller = dsts[i];
        }
        return st;
    } catch (const std::exception& e) {
        set_error("%s", e.what());
        return nullptr;
    }
}
}  // namespace

int jsp_staged_decode(jsp_codec* c, jsp_staged* s) {
    return guarded([&] {
        if (!c || !s) throw std::runtime_error("null argument");
        c->activate();
        s->decode(c->stream);
        return 0;
    });
}

void jsp_staged_destroy(jsp_staged* s) { delete s; }

int jsp_staged_get_info(const jsp_staged* s, jsp_staged_info* out) {
    if (!s || !out) return JSP_ERROR_OCCURED;
    *out = s->info;
    return 0;
}

const char* jsp_staged_kernels(const jsp_staged* s) { return s ? s->kernels.c_str() : ""; }

int jsp_staged_results(jsp_staged* s, int* status, int* adopted, int* significant) {
    if (!s) return JSP_ERROR_OCCURED;
    return guarded([&] {
        s->finish_results();
        for (size_t i = 0; i < s->status.size(); ++i) {
            if (status) status[i] = s->status[i];
            if (adopted) adopted[i] = s->adopted[i];
            if (significant) significant[i] = s->significant[i] < 0 ? 0 : s->significant[i];
        }
        return 0;
    });
}

int jsp_decompress_i_batch(jsp_codec* c, int nframes, const uint8_t* const* srcs, const size_t* lens,
                           int32_t* const* dsts) {
    jsp_staged* st = jsp_stage_batch(c, nframes, srcs, lens, nullptr, dsts);
    if (!st) return JSP_ERROR_OCCURED;
    int rc = jsp_staged_decode(c, st);
    if (rc == 0) rc = jsp_sync(c);
    if (rc == 0) {
        st->finish_results();
        for (int s : st->status)
            if (s != JSP_ZERO_STATE) rc = s;
        if (rc != 0 && !st->why.empty()) set_error("%s", st->why.c_str());
    }
    jsp_staged_destroy(st);
    return rc;
}

}  // extern "C"
