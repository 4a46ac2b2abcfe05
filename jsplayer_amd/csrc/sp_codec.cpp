// ScreenPressor behind the IVideoCodec-shaped C ABI: host entropy stage (sp_host.cpp) + HIP
// reconstruction (sp_kernels.hip).
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <unordered_set>

#include "codec.h"
#include "sp.h"

namespace jsp {
namespace {
using namespace jsp::sp;

struct SpStaged : jsp_staged {
    Geometry geo{};
    struct Op {
        enum Kind { Intra, Inter, InterGroup } kind;
        int first, count;      // Intra: range in the IFrameArgs array; InterGroup: range in the PGroupFrame array
        int band_rows;         // Intra: rows per band (0 = one band)
        bool tiles;            // Intra: tile layout (one wave per band x span), else row-major
        int32_t* dst;          // Inter
        const int32_t* prev;   // Inter, InterGroup: the frame before the (first) frame
        size_t block_off, payload_off;
    };
    std::vector<Op> ops;
    DeviceBuffer d_runs, d_rows, d_seeds, d_tileidx, d_left, d_iargs, d_blocks, d_payload, d_gframes;
    PinnedBuffer h_pack;   // every table of the batch, back to back: the uploads read from here, nobody waits for them
    // the batch's tables while the host stage builds them (kept from use to use of this object: no per-frame allocation)
    std::vector<IRun> runs;
    std::vector<uint32_t> rows, seeds, tileidx, left, payload;
    std::vector<IFrameArgs> iargs;
    std::vector<size_t> iarg_run_off, iarg_row_off, iarg_seed_off, iarg_tile_off, iarg_left_off;
    std::vector<PBlock> blocks;
    std::vector<PGroupFrame> gframes;

    void decode(hipStream_t stream) override {
        for (const Op& op : ops) {
            if (op.kind == Op::Intra && op.tiles)
                launch_iframe_tiles(geo, static_cast<const IFrameArgs*>(d_iargs.p) + op.first, op.count, op.band_rows, stream);
            else if (op.kind == Op::Intra)
                launch_iframes(geo, static_cast<const IFrameArgs*>(d_iargs.p) + op.first, op.count, op.band_rows, stream);
            else if (op.kind == Op::InterGroup)
                launch_pframe_group(geo, static_cast<const PGroupFrame*>(d_gframes.p) + op.first, op.count, op.prev,
                                    static_cast<const PBlock*>(d_blocks.p), static_cast<const uint32_t*>(d_payload.p),
                                    geo.aligned16, stream);
            else
                launch_pframe(geo, op.dst, op.prev, static_cast<const PBlock*>(d_blocks.p) + op.block_off,
                              static_cast<const uint32_t*>(d_payload.p) + op.payload_off, stream);
        }
        JSP_HIP(hipGetLastError());
        decoded = true;
    }
};

std::atomic<int> g_sp_async_streams{0};   // ScreenPressor codec instances of this process whose asynchronous calls run on worker threads

struct SpCodec : jsp_codec, DstColumns {
    HostDecoder host;
    // ---- what the caller's frame buffers hold in their last column (DstColumns) ------------------------------------------------
    // The reference's inter frames read ONE kind of pixel from their destination before writing it (HostDecoder::
    // set_destination_column); to hand back what the reference hands back whatever the caller's buffer rotation, the codec remembers
    // the last column of every picture it has decoded into a buffer, and fetches the column of a buffer it has never written
    // (the caller's own content) from the device the first time it is asked for.  4 bytes per row and buffer.
    std::mutex col_mu;
    std::unordered_map<const void*, std::vector<int32_t>> last_col;
    const int32_t* before(const HostFrame& f) override {
        std::lock_guard<std::mutex> lk(col_mu);
        std::vector<int32_t>& col = last_col[f.dst_host ? static_cast<const void*>(f.dst_host) : f.dst];
        if (f.dst_host) {                              // host-pointer mode: the caller's buffer is right there (read afresh: it is the caller's to change)
            col.resize((size_t)Y);
            for (int y = 0; y < Y; ++y) col[y] = f.dst_host[(size_t)y * X + X - 1];
            return col.data();
        }
        if (col.size() != (size_t)Y) {                 // never decoded into by this codec: the caller's content, as it stands in HBM
            col.assign((size_t)Y, 0);
            if (hipSetDevice(device) != hipSuccess ||
                hipMemcpy2D(col.data(), sizeof(int32_t), static_cast<const int32_t*>(f.dst) + (X - 1), sizeof(int32_t) * (size_t)X, sizeof(int32_t), (size_t)Y,
                            hipMemcpyDeviceToHost) != hipSuccess) {
                (void)hipGetLastError();
                col.clear();
                return nullptr;                        // (not known: the decoder reads its own shadow of the position)
            }
        }
        return col.data();
    }
    void after(const HostFrame& f, const HostDecoder& d, const FrameOut& out) override {
        if (f.dst_host) return;                        // (host mode: the buffer is read afresh every time)
        if (!out.adopted) {
            // nothing was adopted.  A frame that FAILED part-way may still have painted: what the buffer holds is then neither the picture
            // remembered here nor a picture of the decoder's — forget it, the next use asks the device.  (An unchanged frame wrote nothing.)
            if (out.status != 0 && f.dst) { std::lock_guard<std::mutex> lk(col_mu); last_col.erase(f.dst); }
            return;
        }
        std::lock_guard<std::mutex> lk(col_mu);
        if (last_col.size() > 8192) last_col.clear();  // (a caller that keeps handing in new buffers: forget, fetch again when asked)
        std::vector<int32_t>& col = last_col[f.dst];
        col.resize((size_t)Y);
        d.last_column(col.data());
    }
    std::vector<FrameOut> outs;                          // what the host stage says about the frames in hand (their tables keep their memory)
    std::vector<std::unique_ptr<HostDecoder>> spare;     // decoders for the groups of pictures of a batch decoded side by side
    int opt_host_threads = 0;                            // 0 = auto
    SpCodec(int w, int h, int bpp) : host(w, h, bpp) {
        kind = JSP_CODEC_SCREENPRESSOR;
        X = w;
        Y = h;
        if (w > kMaxIntraWidth) throw std::runtime_error("ScreenPressor frames wider than 8192 pixels are not supported");
        if (iframe_lds_bytes(host.geo()) > 160 * 1024) throw std::runtime_error("ScreenPressor frame too large for the LDS plan of the I-frame kernel");
    }
    int preinit(int lines) override { worker_drain(); host.preinit(lines); return JSP_ZERO_STATE; }

    // ---- asynchronous path on worker threads: groups of pictures side by side ------------------------------------------------
    // A group = the frames from one coded key frame up to the next.  The frames handed to the asynchronous calls since the
    // last drain form groups in submission order; the first continues whatever the stream's decoder (`host`) holds, every
    // later one gets a decoder of its own (kept in `spare` between uses).  A group is run from start to finish by ONE worker
    // thread: host entropy stage of a frame, its tables packed and uploaded, its kernels queued on the codec's stream, the
    // job's event recorded — frame after frame, as they are submitted.  Frames of different groups depend on nothing of each
    // other (neither decoder state nor pixels), so their host stages overlap; inside a group everything stays in order.
    // If a group's key frame does not decode, older state shows through in the reference (the models are only renewed by a
    // key frame that decodes): the group then waits for the group before it and goes on with THAT group's decoder.
    struct Group {
        std::unique_ptr<HostDecoder> own;        // null: the stream's decoder
        HostDecoder* dec = nullptr;
        std::shared_ptr<Group> before;           // the group in front, until this group's first frame has decoded
        std::deque<jsp_async_job*> tasks;
        bool closed = false, finished = false, first_done = false, successor_settled = false;
        // the previous frame as the group's frames REALLY left it (device pointer), once its first frame is through: what the next
        // frame of the group is decoded against.  The submission-time prediction (jsp_async_job::prev_dev_before: from the frame's
        // first byte) only serves a group's first frame — an inter frame that aborts, or one that finds no entropy coder yet, adopts
        // nothing whatever its first byte promised, and the frames behind it must copy from the picture that is really there.
        int32_t* prev = nullptr;
        bool have_prev = false;
    };
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::deque<std::shared_ptr<Group>> ready;     // groups no worker has taken yet
    std::vector<std::shared_ptr<Group>> groups;   // every group since the last drain, in order
    std::vector<std::thread> workers;
    bool quitting = false;
    int opt_async_threads = 0;                    // 0 = auto; 1 = the calling thread does everything (no workers)
    int stream_version = 0;                       // entropy coder the stream is pinned to (0: none yet), as far as submission has got
    bool seen_key = false;                        // (prediction) a key frame has been decoded
    int32_t* pred_prev_dev = nullptr;             // (prediction) device pointer of the previous frame after the last submitted frame
    struct JobExtra {
        bool done = false;
        uint64_t seq = 0;                         // submission order (0: the slot was never used)
    };
    uint64_t submit_seq = 0;
    static constexpr size_t kAfterRing = 64;      // (more than frames can be in flight)
    std::pair<uint64_t, int32_t*> after_ring[kAfterRing] = {};   // [seq mod 64] = {seq, the previous frame once that frame was through}: what the next frame really follows
    std::vector<JobExtra> extra;                  // per ring slot: the worker is through with the slot's job

    // auto: the host's threads are shared by the ScreenPressor streams of the process that use the asynchronous calls — one
    // stream gets up to 8 workers, sixteen streams on sixteen cores get none (each then runs its host stage inside the call,
    // as before: sixteen streams already keep sixteen cores busy)
    int async_threads() const {
        if (opt_async_threads > 0) return opt_async_threads;
        const int streams = std::max(1, g_sp_async_streams.load());
        int t = usable_cpus() / streams;
        return t < 2 ? 1 : (t > 8 ? 8 : t);
    }
    bool async_by_workers() override {
        if (!counted_async) { counted_async = true; g_sp_async_streams.fetch_add(1); }   // (a stream that uses the asynchronous calls)
        if (async_threads() > 1) return true;
        worker_drain();                            // (the budget changed under a stream that had workers: back to the calling thread)
        return false;
    }
    bool counted_async = false;

    ~SpCodec() override {
        try { worker_drain(); } catch (...) {}
        {
            std::lock_guard<std::mutex> lk(mu);
            quitting = true;
        }
        cv_work.notify_all();
        for (auto& t : workers) t.join();
        if (counted_async) g_sp_async_streams.fetch_sub(1);
    }

    void worker_main() {
        (void)hipSetDevice(device);
        std::vector<FrameOut> outs_local(1);
        for (;;) {
            std::shared_ptr<Group> g;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return quitting || !ready.empty(); });
                if (ready.empty()) return;        // quitting
                g = ready.front();
                ready.pop_front();
            }
            for (;;) {
                jsp_async_job* j = nullptr;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv_work.wait(lk, [&] { return !g->tasks.empty() || g->closed; });
                    if (g->tasks.empty()) {
                        g->finished = true;
                        cv_done.notify_all();
                        break;
                    }
                    j = g->tasks.front();
                    g->tasks.pop_front();
                }
                run_job(*g, *j, outs_local);
            }
        }
    }

    // (worker thread) one frame: host stage on the group's decoder, tables up, kernels queued, event recorded
    void run_job(Group& g, jsp_async_job& j, std::vector<FrameOut>& outs_local) {
        SpStaged* st = dynamic_cast<SpStaged*>(j.st.get());
        std::string failed;
        try {
            const bool opens_here = !g.have_prev;      // the group's first frame: what it follows was left by ANOTHER worker
            int32_t* prev = g.have_prev ? g.prev : j.prev_dev_before;
            const int32_t* prev_in = prev;             // what a key frame is compared with (option "key_frame_compare")
            jsp_staged* got = stage_impl(std::vector<jsp_frame_in>{j.frame}, st, g.dec, &prev, &outs_local);
            if (got != j.st.get()) j.st.reset(got);
            if (!g.first_done && g.own && j.st->status[0] != JSP_ZERO_STATE) {
                // the group's key frame did not decode: what the stream held before shows through — go on with the decoder
                // of the group in front, once that group is through
                std::shared_ptr<Group> b = g.before;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv_done.wait(lk, [&] { return b->finished; });
                }
                g.dec = b->dec;
                prev = b->have_prev ? b->prev : j.prev_dev_before;   // (that group is through: what it really left)
                prev_in = prev;
                got = stage_impl(std::vector<jsp_frame_in>{j.frame}, j.st.get(), g.dec, &prev, &outs_local);
                if (got != j.st.get()) j.st.reset(got);
            } else if (!g.first_done && g.before) {
                std::lock_guard<std::mutex> lk(mu);
                g.before->successor_settled = true;     // its decoder will not be asked for again
                g.before.reset();
            }
            g.first_done = true;
            g.prev = prev;
            g.have_prev = true;
            j.st->device = device;
            j.st->decode(stream);
            if (j.frame.key && key_compare_row >= 0 && j.st->status[0] == JSP_ZERO_STATE && j.st->adopted[0]) {
                if (opens_here) {
                    // The picture before this one belongs to the group in front, whose worker queues its kernels when ITS host stage is
                    // through — possibly after this point.  The compare must come behind them on the stream, and against the picture that
                    // group really left (the submission-time guess may be off by a frame that adopted nothing): wait until every frame
                    // submitted earlier has queued its work.  (Earlier frames only: groups are handed to workers in order, so whoever is
                    // waited for here has a worker and waits, if at all, for still earlier ones.)
                    const size_t me = (size_t)(&j - jobs.data());
                    std::unique_lock<std::mutex> lk(mu);
                    const uint64_t my = extra[me].seq;
                    cv_done.wait(lk, [&] {
                        for (size_t k = 0; k < extra.size(); ++k)
                            if (k != me && extra[k].seq != 0 && extra[k].seq < my && !extra[k].done) return false;
                        return true;
                    });
                    if (my > 1 && after_ring[(my - 1) % kAfterRing].first == my - 1) prev_in = after_ring[(my - 1) % kAfterRing].second;
                }
                if (prev_in) {
                    queue_key_compare(j.frame.dst, prev_in, (int)(&j - jobs.data()));
                    j.key_compare_queued = true;
                }
            }
            JSP_HIP(hipEventRecord(j.done, stream));
        } catch (const std::exception& e) {
            failed = e.what();
        }
        if (!failed.empty()) {                     // (out of memory, a HIP error: the frame reports it; nothing is left half-queued that a wait could hang on)
            if (!j.st) j.st.reset(new SpStaged());
            j.st->status.assign(1, JSP_ERROR_OCCURED);
            j.st->adopted.assign(1, 0);
            j.st->significant.assign(1, 0);
            j.st->cleared.assign(1, 0);
            j.st->why = failed;
            (void)hipEventRecord(j.done, stream);
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            const uint64_t seq = extra[&j - jobs.data()].seq;
            after_ring[seq % kAfterRing] = {seq, g.have_prev ? g.prev : j.prev_dev_before};
            extra[&j - jobs.data()].done = true;
        }
        cv_done.notify_all();
    }

    void worker_submit(jsp_async_job& j) override {
        if (extra.size() < jobs.size()) extra.resize(jobs.size());
        const HostFrame hf{j.frame.src, j.frame.n, j.frame.key};
        std::shared_ptr<Group> cur = groups.empty() ? nullptr : groups.back();
        if (!cur) {                                 // first frame since the last drain: predictions start from the codec's state
            // (and the stream's settings are noted HERE, on the caller's thread with nothing in flight: from now on a worker may be inside `host` —
            // the first group decodes on it and sets its key-frame layout there — and this thread must not read it.  ThreadSanitizer, round 6.)
            settings_at_submit = host.settings();
            stream_version = host.pinned_version();
            seen_key = host.has_prev() || stream_version != 0;
            pred_prev_dev = prev_dev;
        }
        const bool opens = starts_group(hf);
        const int version = opens ? (j.frame.src[0] >> 4) + 1 : 0;
        // a new group of its own needs the stream's entropy coder to be known (pinned by the first coded key frame, which
        // therefore always runs in the group in hand)
        bool new_group = !cur;
        std::unique_ptr<HostDecoder> fresh;
        if (cur && opens && stream_version >= 2 && stream_version <= 4) {
            const Geometry g = host.geo();
            {
                std::lock_guard<std::mutex> lk(mu);
                // decoders of groups that are through and whose successor has started well go back to the shelf, and the groups
                // themselves go: a caller that only ever uses the asynchronous calls never drains, and a stream of key frames would
                // otherwise keep (and scan, under the lock the workers need) one group per key frame for as long as it runs
                size_t kept = 0;
                for (size_t i = 0; i < groups.size(); ++i) {
                    auto& old = groups[i];
                    const bool gone = old != cur && old->finished && old->successor_settled;
                    if (gone && old->own) spare.push_back(std::move(old->own));
                    if (!gone) { if (kept != i) groups[kept] = std::move(old); ++kept; }
                }
                groups.resize(kept);
                while (spare.size() > 16) spare.pop_back();      // (the shelf need not grow with the stream either)
            }
            if (!spare.empty()) { fresh = std::move(spare.back()); spare.pop_back(); }
            if (fresh && fresh->pinned_version() != 0 && fresh->pinned_version() != stream_version) fresh.reset();   // it served another coder
            if (!fresh) fresh = std::make_unique<HostDecoder>(g.X, g.Y, g.bpp);
            fresh->adopt_settings(settings_at_submit);
            new_group = fresh->pin_version(stream_version);
            if (!new_group) spare.push_back(std::move(fresh));
        }
        if (opens && stream_version == 0) stream_version = version;   // (the coder is chosen before the frame is decoded: ScreenPressor.hx:130-131)
        if (new_group) {
            auto g = std::make_shared<Group>();
            if (cur) {
                g->own = std::move(fresh);
                g->dec = g->own.get();
                g->before = cur;
            } else {
                g->dec = &host;
            }
            std::lock_guard<std::mutex> lk(mu);
            if (cur) cur->closed = true;
            groups.push_back(g);
            ready.push_back(g);
            cur = g;
        }
        // what the frame will do to the previous frame, from its first byte (ScreenPressor.hx:130-159, 308-313)
        bool adopts;
        if (j.frame.key) adopts = j.frame.n > 0 && ((j.frame.src[0] & 0xF) == 1 || (j.frame.src[0] & 0xF) == 2);
        else adopts = j.frame.n > 0 && seen_key && j.frame.src[0] != 0;
        if (j.frame.key && adopts) seen_key = true;
        j.prev_dev_before = pred_prev_dev;
        if (adopts) { pred_prev_dev = j.frame.dst; prev_caller = j.frame.dst; prev_dev = j.frame.dst; }
        if ((int)workers.size() < async_threads()) workers.emplace_back([this] { worker_main(); });
        {
            std::lock_guard<std::mutex> lk(mu);
            extra[&j - jobs.data()].done = false;
            extra[&j - jobs.data()].seq = ++submit_seq;
            cur->tasks.push_back(&j);
        }
        cv_work.notify_all();
    }
    HostDecoder::Settings settings_at_submit;   // the stream decoder's Preinit / key-frame layout as they stood when the first frame since the last drain was submitted

    void worker_wait(jsp_async_job& j) override {
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return extra[&j - jobs.data()].done; });
    }

    void worker_drain() override {
        if (groups.empty()) return;
        std::shared_ptr<Group> last = groups.back();
        {
            std::unique_lock<std::mutex> lk(mu);
            last->closed = true;
            cv_work.notify_all();
            cv_done.wait(lk, [&] {
                for (auto& g : groups) if (!g->finished) return false;
                return true;
            });
        }
        // the stream goes on from the decoder the last group ended on
        if (last->dec != &host) std::swap(host, *last->dec);
        for (auto& g : groups)
            if (g->own) spare.push_back(std::move(g->own));
        prev_dev = last->have_prev ? last->prev : pred_prev_dev;   // what the frames really left (the prediction where nothing ran)
        groups.clear();
    }
    // jsp_counter: "sp_groups_held" = groups of pictures the asynchronous path still keeps a record of (bounded however long a stream
    // runs without a drain), "sp_spare_decoders" = decoders on the shelf
    long long counter(const char* name) override {
        std::lock_guard<std::mutex> lk(mu);
        if (std::strcmp(name, "sp_groups_held") == 0) return (long long)groups.size();
        if (std::strcmp(name, "sp_spare_decoders") == 0) return (long long)spare.size();
        return -1;
    }
    int is_key_frame(const uint8_t* src, size_t n) override { return HostDecoder::is_key_frame(src, n) ? 1 : 0; }
    int needs_index() override { return 0; }
    bool may_leave_pixels(const jsp_frame_in&) override { return false; }
    int opt_band_rows = -1;   // -1: chosen per batch (choose_band_rows); 0: one band per frame; n: n rows per band
    bool opt_inter_fusion = true;   // consecutive inter frames of a staged batch share one launch
    int set_option(const char* key, const char* value) override {
        if (std::strcmp(key, "sp_host_threads") == 0) {
            if (std::strcmp(value, "auto") == 0) { opt_host_threads = 0; return 0; }
            char* end = nullptr;
            const long v = std::strtol(value, &end, 10);
            if (end == value || *end || v < 1 || v > 64) return -1;
            opt_host_threads = (int)v;
            return 0;
        }
        if (std::strcmp(key, "sp_async_threads") == 0) {   // worker threads of the asynchronous per-frame calls (1: none)
            if (next_ticket != oldest_ticket) return -1;
            worker_drain();
            if (std::strcmp(value, "auto") == 0) { opt_async_threads = 0; return 0; }
            char* end = nullptr;
            const long v = std::strtol(value, &end, 10);
            if (end == value || *end || v < 1 || v > 64) return -1;
            opt_async_threads = (int)v;
            return 0;
        }
        if (std::strcmp(key, "sp_forget_buffers") == 0) {   // the caller has written into frame buffers itself: what they hold is asked for again
            if (next_ticket != oldest_ticket) return -1;
            worker_drain();
            std::lock_guard<std::mutex> lk(col_mu);
            last_col.clear();
            return 0;
        }
        if (std::strcmp(key, "sp_inter_fusion") == 0) {
            if (std::strcmp(value, "on") == 0) { opt_inter_fusion = true; return 0; }
            if (std::strcmp(value, "off") == 0) { opt_inter_fusion = false; return 0; }
            return -1;
        }
        if (std::strcmp(key, "sp_band_rows") == 0) {
            if (std::strcmp(value, "auto") == 0) { opt_band_rows = -1; return 0; }
            char* end = nullptr;
            const long v = std::strtol(value, &end, 10);
            if (end == value || *end || v < 0 || v > 1 << 20) return -1;
            opt_band_rows = (int)v;
            return 0;
        }
        return -1;
    }

    jsp_staged* stage(const std::vector<jsp_frame_in>& frames, jsp_staged* reuse) override {
        return stage_impl(frames, reuse, nullptr, nullptr, nullptr);
    }
    // `one` (worker threads of the asynchronous path): the single frame of `frames` goes through THAT decoder, the previous
    // frame's device pointer comes from / goes to *prev_io and the frame's tables are built in *outs_one — nothing of the
    // codec that another thread may be using is touched.
    jsp_staged* stage_impl(const std::vector<jsp_frame_in>& frames, jsp_staged* reuse, HostDecoder* one, int32_t** prev_io,
                           std::vector<FrameOut>* outs_one) {
        activate();
        HostDecoder& host = one ? *one : this->host;
        int32_t*& prev_dev = prev_io ? *prev_io : this->prev_dev;
        std::vector<FrameOut>& outs = outs_one ? *outs_one : this->outs;
        const double t0 = now_ms();
        auto* st = dynamic_cast<SpStaged*>(reuse);
        std::unique_ptr<SpStaged> guard;
        if (!st) { st = new SpStaged(); guard.reset(st); }
        const int nf = (int)frames.size();
        const Geometry& g = host.geo();
        st->geo = g;
        for (const auto& f : frames)
            if (reinterpret_cast<uintptr_t>(f.dst) & 15) st->geo.aligned16 = false;
        st->ops.clear();
        st->decoded = false;
        st->status.assign(nf, JSP_ZERO_STATE);
        st->adopted.assign(nf, 0);
        st->significant.assign(nf, 0);
        st->cleared.assign(nf, 0);
        st->key_differs.assign(nf, -2);
        st->info = jsp_staged_info{};

        auto &runs = st->runs;
        auto &rows = st->rows, &seeds = st->seeds, &tileidx = st->tileidx, &left = st->left, &payload = st->payload;
        auto &iargs = st->iargs;
        auto &iarg_run_off = st->iarg_run_off, &iarg_row_off = st->iarg_row_off, &iarg_seed_off = st->iarg_seed_off,
             &iarg_tile_off = st->iarg_tile_off, &iarg_left_off = st->iarg_left_off;
        auto &blocks = st->blocks;
        auto &gframes = st->gframes;
        runs.clear(); rows.clear(); seeds.clear(); tileidx.clear(); left.clear(); payload.clear(); iargs.clear();
        iarg_run_off.clear(); iarg_row_off.clear(); iarg_seed_off.clear(); iarg_tile_off.clear(); iarg_left_off.clear();
        blocks.clear(); gframes.clear();
        int nkey = 0;
        for (const auto& f : frames) nkey += f.key ? 1 : 0;
        int band_rows = opt_band_rows >= 0 ? opt_band_rows : choose_band_rows(g, nkey);   // one cut for the whole batch
        const bool tiles = iframe_tiles_ok(st->geo);   // key frames as independent tiles (needs aligned buffers)
        if (tiles && g.Y > iframe_tile_max_band_rows() && (band_rows <= 0 || band_rows > iframe_tile_max_band_rows())) band_rows = iframe_tile_max_band_rows();   // a tile's row index and left pixels live in LDS, 12 bytes per row of the band (tile_plan, sp_kernels.hip: bands taller than ~550 rows share a workgroup among fewer waves)
        host.set_iframe_layout(band_rows, tiles ? iframe_tile_span(st->geo) : 0);
        // the key-frame compare by the host stage: only where this decoder also decoded the frame before — the stream's own decoder
        // taking one frame at a time (a group of pictures on a decoder of its own does not hold the picture before its key frame)
        host.set_key_compare_row(!one && nf == 1 ? key_compare_row : -1);
        // Inter frames are fused per launch when the batch has several of them; a frame that moves more
        // than a quarter of its pixels keeps its motion blocks (literal pixels for them would rival the
        // frame in size) and gets a launch of its own.
        const bool fuse_inter = opt_inter_fusion && nf - nkey >= 2;
        std::unordered_set<const void*> group_dsts;
        // The host stage runs over the batch in waves: up to `threads` groups of pictures (a coded key frame and what follows it)
        // side by side, at most 64 frames, then their tables are taken into the batch in stream order.
        std::vector<HostFrame> hf(nf);
        for (int i = 0; i < nf; ++i) hf[i] = HostFrame{frames[i].src, frames[i].n, frames[i].key, frames[i].dst, frames[i].caller_host_dst};
        int threads = opt_host_threads;
        if (threads <= 0) { threads = usable_cpus(); threads = threads < 1 ? 1 : (threads > 8 ? 8 : threads); }
        if (nf > 1) {   // a buffer used twice in the batch: what it holds when its second frame is decoded is what the first left — in order only
            std::unordered_set<const void*> seen;
            for (const auto& f : frames)
                if (!seen.insert(f.dst).second) { threads = 1; break; }
        }
        for (int w0 = 0; w0 < nf;) {
            int w1 = w0 + 1, groups = 1;
            while (w1 < nf && w1 - w0 < 64) {
                if (starts_group(hf[w1])) { if (groups == threads) break; ++groups; }
                ++w1;
            }
            if ((int)outs.size() < w1 - w0) outs.resize(w1 - w0);
            if (one) decode_single(host, hf[w0], outs[0], fuse_inter, this);
            else decode_frames(host, spare, hf.data() + w0, w1 - w0, outs.data(), threads, fuse_inter, this);
        for (int i = w0; i < w1; ++i) {
            const jsp_frame_in& f = frames[i];
            FrameOut& fo = outs[i - w0];
            st->status[i] = fo.status;
            st->adopted[i] = fo.adopted ? 1 : 0;
            st->significant[i] = fo.significant ? 1 : 0;
            st->cleared[i] = fo.prev_cleared ? 1 : 0;
            st->key_differs[i] = fo.key_differs;
            if (fo.status != JSP_ZERO_STATE && fo.error) { set_error("%s", fo.error); st->why = fo.error; }
            if (fo.prev_cleared) prev_dev = nullptr;
            st->info.stream_bytes += fo.stream_bytes;
            const uint64_t npx = (uint64_t)g.X * g.Y;
            switch (fo.kind) {
                case FrameKind::Flat:
                case FrameKind::Intra: {
                    IFrameArgs a{};
                    a.dst = f.dst;
                    a.flat = fo.kind == FrameKind::Flat;
                    a.colour = fo.flat_colour;
                    a.nruns = (uint32_t)fo.runs.size();
                    iarg_run_off.push_back(runs.size());
                    iarg_row_off.push_back(rows.size());
                    iarg_seed_off.push_back(seeds.size());
                    iarg_tile_off.push_back(tileidx.size());
                    iarg_left_off.push_back(left.size());
                    runs.insert(runs.end(), fo.runs.begin(), fo.runs.end());
                    if (!tiles) rows.insert(rows.end(), fo.row_run.begin(), fo.row_run.end());
                    seeds.insert(seeds.end(), fo.seeds.begin(), fo.seeds.end());
                    tileidx.insert(tileidx.end(), fo.tile_idx.begin(), fo.tile_idx.end());
                    left.insert(left.end(), fo.left.begin(), fo.left.end());
                    const bool join = !st->ops.empty() && st->ops.back().kind == SpStaged::Op::Intra &&
                                      !group_dsts.count(f.dst);
                    if (join) st->ops.back().count++;
                    else {
                        st->ops.push_back({SpStaged::Op::Intra, (int)iargs.size(), 1, band_rows, tiles, nullptr, nullptr, 0, 0});
                        group_dsts.clear();
                    }
                    group_dsts.insert(f.dst);
                    iargs.push_back(a);
                    const uint64_t r = fo.stream_runs;
                    st->info.runs += r;
                    st->info.units_coded += npx;
                    st->info.algorithmic_bytes += 8 * r + 4 * npx;  // A = 8R + 4P
                    break;
                }
                case FrameKind::Inter: {
                    group_dsts.clear();
                    if (fuse_inter && fo.literalised && blocks.size() < (1u << 31) && payload.size() < (1u << 31)) {
                        // (the group kernel relies on the block tables of a group's frames following each other)
                        const bool extend = !st->ops.empty() && st->ops.back().kind == SpStaged::Op::InterGroup && st->ops.back().count < kGroupMaxFrames &&
                                            (size_t)gframes.back().block_off + fo.blocks.size() == blocks.size();
                        if (!extend)
                            st->ops.push_back({SpStaged::Op::InterGroup, (int)gframes.size(), 0, 0, false, nullptr, prev_dev, 0, 0});
                        st->ops.back().count++;
                        gframes.push_back({f.dst, (uint32_t)blocks.size(), (uint32_t)payload.size()});
                    } else {
                        st->ops.push_back({SpStaged::Op::Inter, 0, 0, 0, false, f.dst, prev_dev, blocks.size(), payload.size()});
                    }
                    blocks.insert(blocks.end(), fo.blocks.begin(), fo.blocks.end());
                    payload.insert(payload.end(), fo.payload.begin(), fo.payload.end());
                    payload.resize((payload.size() + 3) & ~size_t(3), 0u);   // every frame's literals start on a 16-byte boundary of the batch's table (so does every rectangle inside it)
                    st->info.units_coded += fo.data_pixels;
                    st->info.units_copied += fo.prev_pixels;
                    // A = 4P written + 4 P_prev fetched + 16 N_blk + literal payload of the data rectangles
                    st->info.algorithmic_bytes += 4 * npx + 4 * fo.prev_pixels + 16 * (uint64_t)fo.blocks.size() +
                                                  4 * fo.data_pixels;
                    break;
                }
                case FrameKind::None: break;
            }
            if (fo.adopted) prev_dev = f.dst;
        }
            w0 = w1;
        }
        if (!one && outs.size() > 1) outs.resize(1);   // (a wave's worth of frame tables is hundreds of MB: only the per-frame calls' one stays)
        st->info.frames = nf;
        st->info.pixels = (uint64_t)g.X * g.Y * nf;
        st->info.kernel_launches = st->ops.size();
        st->info.descriptor_bytes = runs.size() * sizeof(IRun) + (rows.size() + seeds.size() + tileidx.size() + left.size()) * 4 +
                                    iargs.size() * sizeof(IFrameArgs) +
                                    blocks.size() * sizeof(PBlock) + payload.size() * 4 + gframes.size() * sizeof(PGroupFrame);
        // what the plan moves: every table read once, every frame written once, the previous frame read once per
        // inter launch (the group kernel carries pixels in registers from frame to frame)
        {
            const uint64_t npx = (uint64_t)g.X * g.Y;
            uint64_t moved = st->info.descriptor_bytes;
            st->kernels.clear();
            for (const auto& op : st->ops) {
                if (op.kind == SpStaged::Op::Intra) {
                    moved += 4 * npx * op.count;
                    st->note_kernel(op.tiles ? "sp_iframe_tile_kernel" : "sp_iframe_rows_search_kernel");
                } else if (op.kind == SpStaged::Op::InterGroup) {
                    moved += 4 * npx * op.count + 4 * npx;
                    st->note_kernel("sp_pframe_group_kernel");
                } else {
                    moved += 8 * npx;
                    st->note_kernel("sp_pframe_kernel");
                }
            }
            st->info.moved_bytes = moved;
        }
        st->info.host_stage_ms = now_ms() - t0;

        const double t1 = now_ms();
        st->d_runs.reserve(std::max<size_t>(runs.size(), 1) * sizeof(IRun));
        st->d_rows.reserve(std::max<size_t>(rows.size(), 1) * 4);
        st->d_seeds.reserve(std::max<size_t>(seeds.size(), 1) * 4);
        st->d_tileidx.reserve(std::max<size_t>(tileidx.size(), 1) * 4);
        st->d_left.reserve(std::max<size_t>(left.size(), 1) * 4);
        st->d_iargs.reserve(std::max<size_t>(iargs.size(), 1) * sizeof(IFrameArgs));
        st->d_blocks.reserve(std::max<size_t>(blocks.size(), 1) * sizeof(PBlock));
        st->d_payload.reserve(std::max<size_t>(payload.size(), 1) * 4 + 16);
        st->d_gframes.reserve(std::max<size_t>(gframes.size(), 1) * sizeof(PGroupFrame));
        for (size_t k = 0; k < iargs.size(); ++k) {
            iargs[k].runs = static_cast<const IRun*>(st->d_runs.p) + iarg_run_off[k];
            iargs[k].row_run = static_cast<const uint32_t*>(st->d_rows.p) + iarg_row_off[k];
            iargs[k].seeds = static_cast<const uint32_t*>(st->d_seeds.p) + iarg_seed_off[k];
            iargs[k].tile_idx = static_cast<const uint32_t*>(st->d_tileidx.p) + iarg_tile_off[k];
            iargs[k].left = static_cast<const uint32_t*>(st->d_left.p) + iarg_left_off[k];
        }
        // the tables go up from pinned memory that belongs to the staged batch: nothing waits for the copies (a batch
        // is decoded on the same stream, behind them)
        struct Part { DeviceBuffer* d; const void* h; size_t bytes; };
        const Part parts[] = {{&st->d_runs, runs.data(), runs.size() * sizeof(IRun)}, {&st->d_rows, rows.data(), rows.size() * 4},
                              {&st->d_seeds, seeds.data(), seeds.size() * 4}, {&st->d_tileidx, tileidx.data(), tileidx.size() * 4},
                              {&st->d_left, left.data(), left.size() * 4}, {&st->d_iargs, iargs.data(), iargs.size() * sizeof(IFrameArgs)},
                              {&st->d_blocks, blocks.data(), blocks.size() * sizeof(PBlock)}, {&st->d_payload, payload.data(), payload.size() * 4},
                              {&st->d_gframes, gframes.data(), gframes.size() * sizeof(PGroupFrame)}};
        size_t total = 0;
        for (const Part& p : parts) total += (p.bytes + 15) & ~size_t(15);
        st->h_pack.reserve(total + 16);
        size_t at = 0;
        for (const Part& p : parts) {
            if (!p.bytes) continue;
            uint8_t* h = static_cast<uint8_t*>(st->h_pack.p) + at;
            std::memcpy(h, p.h, p.bytes);
            JSP_HIP(hipMemcpyAsync(p.d->p, h, p.bytes, hipMemcpyHostToDevice, stream));
            at += (p.bytes + 15) & ~size_t(15);
        }
        st->info.h2d_ms = now_ms() - t1;
        guard.release();
        return st;
    }
};

}  // namespace
}  // namespace jsp

jsp_codec* jsp_make_screenpressor(int w, int h, int bpp) { return new jsp::SpCodec(w, h, bpp); }
