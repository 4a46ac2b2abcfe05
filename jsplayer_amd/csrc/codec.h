// Internal C++ shape of the C ABI objects (include/jsplayer_amd.h).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "../../include/jsplayer_amd.h"
#include "common.h"

// One frame of a batch handed to the host stage.
struct jsp_frame_in {
    const uint8_t* src;
    size_t n;
    bool key;      // DecompressI (true) or DecompressP (false) semantics
    int32_t* dst;  // DEVICE pointer
    const int32_t* caller_host_dst = nullptr;   // host-pointer mode: the caller's own buffer (`dst` is then an internal HBM frame)
};

// A staged batch: descriptor tables resident in HBM + the launch plan.
struct jsp_staged {
    virtual ~jsp_staged() = default;
    virtual void decode(hipStream_t stream) = 0;  // asynchronous
    jsp_staged_info info{};
    std::string kernels;       // jsp_staged_kernels(): names of the kernels decode() launches
    void note_kernel(const char* name) {   // appends `name` unless it is already listed
        if (kernels.find(name) != std::string::npos) return;
        if (!kernels.empty()) kernels += " + ";
        kernels += name;
    }
    std::vector<int> status, adopted, significant;
    std::string why;           // what made a frame fail, for jsp_last_error()
    std::vector<int> cleared;  // frame i ended with prevFrame == null (ScreenPressor RenewI + failure)
    // key-frame compare (jsp_codec::key_compare_row): what the host stage already knows about frame i — 1 / 0: differs / does not from
    // the previous frame at or after that row, -1: there was no previous frame; empty or -2: not known (the GPU pass answers)
    std::vector<int> key_differs;
    // significance words written by the kernels (one per frame); -1 in `significant`
    // marks "take it from the device word"
    jsp::DeviceBuffer d_signif;
    jsp::PinnedBuffer h_signif;
    bool decoded = false;
    int device = -1;                // HIP device the batch's buffers live on (set when it is staged): what a re-run must activate
    bool verdict_pending = false;   // asynchronous staging whose results are final only after async_finish()
    // After the stream has been synchronised.  Never throws: when the codec-specific step fails (a HIP error while a batch is re-run
    // through the descriptor kernels) every frame of the batch reports JSP_ERROR_OCCURED and `why` says what happened.
    void finish_results() noexcept;
    virtual void after_sync() {}   // codec-specific checks of what the kernels reported (called by finish_results; may throw)
};

// One frame in flight on the asynchronous path (jsp_decompress_i_async / _p_async ... jsp_wait).
struct jsp_async_job {
    uint64_t ticket = 0;                 // 0 = slot free
    std::unique_ptr<jsp_staged> st;      // kept from frame to frame: its buffers are recycled
    hipEvent_t done = nullptr;
    jsp_frame_in frame{};                // src stays valid until the ticket has been waited for (caller's contract)
    int32_t* prev_caller_before = nullptr;   // codec state before this frame: what a synchronous re-run starts from
    int32_t* prev_dev_before = nullptr;
    int32_t* prev_caller_after = nullptr;    // PreviousFrame() once this frame is done
    bool redone = false;                 // results already final (the frame was re-run through the synchronous path)
    bool settled = false;                // results already final (its kernels were waited for ahead of its jsp_wait)
    int status = 0, significant = 0;
    std::string why;                     // error text of a final non-zero status
    bool by_worker = false;              // the frame's host stage and launches run on a worker thread of the codec (async_by_workers)
    int key_differs = -1;                // key-frame compare of this frame (jsp_wait hands it back as *significant_changes)
    bool key_compare_queued = false;     // ... is being worked out by the pass queued behind the frame's kernels
};

struct jsp_codec {
    int kind = 0;
    int X = 0, Y = 0;
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipStream_t side_stream = nullptr;   // a second stream of the codec's own (made on first use): work that runs BESIDE the launches on `stream` — jsp_sync waits for both

    // Caller-visible previous frame (identity) and where its pixels live in HBM.
    int32_t* prev_caller = nullptr;
    int32_t* prev_dev = nullptr;

    // Host-pointer compatibility mode: two internal HBM frames used alternately.
    int ptr_mode = 0;  // 0 unknown, 1 device pointers, 2 host pointers
    jsp::DeviceBuffer compat[2];

    std::unique_ptr<jsp_staged> scratch;  // reused by the per-frame entry points

    // asynchronous path: a ring of jobs, tickets count up from 1, frames complete in submission order
    std::vector<jsp_async_job> jobs;
    int async_depth = 4;
    uint64_t next_ticket = 1, oldest_ticket = 1;   // [oldest_ticket, next_ticket) are in flight

    // Key-frame compare (option "key_frame_compare"; Manager.hx:392-421, the pixel loop :413-419): with a first row set, every key
    // frame is also compared with the previous frame as the call found it — by the host stage where it holds both pictures anyway,
    // else by a pass queued on the codec's stream right behind the frame's kernels (no extra wait, the frame still in the Infinity Cache).
    int key_compare_row = -1;                      // -1: off
    int last_key_differs = -1;                     // jsp_key_frame_differs(): the last synchronous key frame (1 / 0; -1: nothing to compare with / off)
    jsp::DeviceBuffer d_keyflag;                   // one word per frame in flight
    jsp::PinnedBuffer h_keyflag;

    virtual ~jsp_codec();
    virtual int preinit(int lines) = 0;
    virtual int is_key_frame(const uint8_t* src, size_t n) = 0;
    virtual int needs_index() = 0;
    // Host stage for `frames` (advances prev_dev / models), returns a staged batch.
    // `reuse` may be a previous result of this codec whose buffers can be recycled.
    virtual jsp_staged* stage(const std::vector<jsp_frame_in>& frames, jsp_staged* reuse) = 0;
    // True when a frame of this batch may leave some dst pixels unwritten (so a host-mode dst
    // has to be uploaded first to keep them as the caller had them).
    virtual bool may_leave_pixels(const jsp_frame_in& f) = 0;
    virtual int set_option(const char*, const char*) { return -1; }
    // jsp_prefetch: a range of host memory the next frames' bytes lie in may be taken to the device in one copy (0: accepted or nothing to do)
    virtual int prefetch(const void*, size_t) { return 0; }
    long long async_reruns = 0;            // jsp_counter("async_reruns")
    virtual long long counter(const char*) { return -1; }   // the codec's own counters (jsp_counter)
    // Host stage of ONE frame for the asynchronous path: like stage(), but it must not wait for the GPU (uploads come
    // from pinned memory owned by the returned object).  What cannot be known without the GPU's answer is settled in
    // async_finish(), called after the frame's event: false = the frame has to be re-run through the synchronous path.
    virtual jsp_staged* stage_async(const jsp_frame_in& f, jsp_staged* reuse) {
        return stage(std::vector<jsp_frame_in>{f}, reuse);
    }
    virtual bool async_finish(jsp_staged*) { return true; }
    // The frame of job `j` is staged (j.st): the codec may take its launch in hand — queue its kernels and record j.done now, or HOLD it
    // until the next frame is submitted and launch the two together (async_flush(&j) or async_flush(nullptr) launches what is held; async_reset
    // drops it: the frame is about to be re-run synchronously).  false: the caller queues st->decode() and records the event itself.
    virtual bool async_launch(jsp_async_job&) { return false; }
    virtual void async_flush(const jsp_async_job* /*only_if_held*/) {}
    // True when stage_async() would stage this frame with kernels that cannot be vetoed afterwards (the synchronous
    // staging): every earlier frame still in flight is then settled first — re-run, if the GPU could not settle it —
    // because the caller may already have handed this frame a buffer an earlier frame's re-run still reads.
    virtual bool async_settle_first(const jsp_frame_in&) { return false; }
    // True when the synchronous calls are better served by submit + wait on the asynchronous path (one launch that settles
    // the frame by itself) than by the batch staging of one frame (parse at staging, a wait, then the decode launch).
    virtual bool sync_through_async() const { return false; }
    // Called (stream idle) before frames are re-run through the synchronous path: undo whatever made the frames in
    // flight behind the failed one stand still.
    virtual void async_reset() {}
    // Asynchronous path served by worker threads of the codec's own (ScreenPressor: a coded key frame renews every bit of
    // decoder state, so the host entropy stage of the frames from one coded key frame to the next runs on a thread and a
    // decoder of its own, beside the groups of pictures before it).  worker_submit() returns at once — it hands the frame to
    // a worker and PREDICTS what the frame does to the previous frame (prev_caller / prev_dev; decidable from the frame's
    // first byte for every stream that decodes) —, worker_wait() blocks until the frame's host stage and launches are
    // through (its event is recorded), worker_drain() until every submitted frame is and puts the stream's state back where
    // the synchronous entry points expect it.
    virtual bool async_by_workers() { return false; }
    virtual void worker_submit(jsp_async_job&) {}
    virtual void worker_wait(jsp_async_job&) {}
    virtual void worker_drain() {}
    int32_t* settled_prev = nullptr;     // (worker path) the previous frame as the frames waited for so far really left it

    void init_device(int device_id);
    void activate();
    // queues the compare of key frame `dst` with `prev` (rows from key_compare_row on) behind what is queued on `stream`; the answer
    // lands in word `slot` of h_keyflag (valid once the stream has got there).  Both buffers exist from the moment the option is set.
    void queue_key_compare(const int32_t* dst, const int32_t* prev, int slot);
    int read_key_compare(int slot) const { return static_cast<const uint32_t*>(h_keyflag.p)[slot] ? 1 : 0; }
};

// Factories (one per codec family).
jsp_codec* jsp_make_msv1(int bits, int w, int h, const uint8_t* palette, int palette_bytes);
jsp_codec* jsp_make_screenpressor(int w, int h, int bpp);
