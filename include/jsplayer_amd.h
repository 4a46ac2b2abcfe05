/*
 * jsplayer_amd — MI355X-native block-video decode path behind jsplayer's IVideoCodec.
 *
 * C ABI (plain pointers and sizes, no C++/torch types).  Every entry point cites the
 * reference interface it replaces; paths are relative to /root/reference/src.
 *
 * Frame buffers follow the reference's contract (MSVideo1.hx:211-214, ScreenPressor.hx:189,
 * Manager.hx:114-118,379): one int32 per pixel, value 0x00RRGGBB, stride = width ints,
 * BOTTOM-UP rows, length >= width*height ints, allocated and owned by the CALLER.  The codec
 * borrows `dst` and keeps it as its "previous frame" until a later frame replaces it; the caller
 * never passes the current previous frame as `dst` (Manager.hx:424-443,477).
 *
 * `dst` may be either
 *   - a DEVICE pointer (HBM; e.g. from jsp_pool_create): the frame stays on the GPU, or
 *   - a HOST pointer: the codec decodes into an internal HBM frame and copies the result back
 *     (compatibility mode for an unmodified Manager; PCIe-bound).
 * The two kinds must not be mixed on one codec instance.
 *
 * Ordering: a codec's own HIP stream is a blocking stream, i.e. ordered after work the caller queued on the legacy
 * default stream; work the caller has pending on other non-blocking streams that touches `dst` or the previous
 * frame must have finished before the call (or give the codec that stream with jsp_set_stream).
 *
 * Threading (reference: single-threaded, not re-entrant): one thread per codec instance at a
 * time; distinct instances may be used concurrently from distinct threads.
 * No exceptions cross this boundary; failures are reported by status + jsp_last_error().
 */
#ifndef JSPLAYER_AMD_H
#define JSPLAYER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct jsp_codec jsp_codec;
typedef struct jsp_pool jsp_pool;
typedef struct jsp_staged jsp_staged;

/* Codec kinds = the three classes Manager.video_info_cb constructs (Manager.hx:105-110). */
enum {
    JSP_CODEC_MSVIDEO1_16 = 1,   /* new MSVideo1_16bit(w,h)            MSVideo1.hx:20-31   */
    JSP_CODEC_MSVIDEO1_8 = 2,    /* new MSVideo1_8bit(w,h,palette)     MSVideo1.hx:267-274 */
    JSP_CODEC_SCREENPRESSOR = 3  /* new ScreenPressor(w,h,bpp)         ScreenPressor.hx:53-64 */
};

/* enum DecoderState (IVideoCodec.hx:5-9) */
enum { JSP_ZERO_STATE = 0, JSP_IN_PROGRESS = 1, JSP_ERROR_OCCURED = 2 };

/* ---- IVideoCodec (IVideoCodec.hx:16-29) ------------------------------------------------- */

/* Constructors (Manager.hx:105-110).  `palette` = the strf bytes after the 40-byte
 * BITMAPINFOHEADER (AVIParser.hx:79-85), used by JSP_CODEC_MSVIDEO1_8 only; `bpp` is used by
 * JSP_CODEC_SCREENPRESSOR only.  `device_id` = HIP device ordinal.  NULL on failure. */
jsp_codec* jsp_codec_create(int kind, int width, int height, int bpp,
                            const uint8_t* palette, int palette_bytes, int device_id);

/* StopAndClean() (IVideoCodec.hx:28; MSVideo1.hx:33-35; ScreenPressor.hx:81-84) + release. */
void jsp_codec_destroy(jsp_codec* c);

/* Preinit(insignificant_lines) (IVideoCodec.hx:18; MSVideo1.hx:37-41,281-291;
 * ScreenPressor.hx:86-89).  Called once by Manager with 36 (Manager.hx:61,128). */
int jsp_preinit(jsp_codec* c, int insignificant_lines);

/* PreviousFrame() (IVideoCodec.hx:20): the caller-owned buffer last adopted, or NULL.
 * Compared BY IDENTITY by the caller (Manager.hx:470-475). */
int32_t* jsp_previous_frame(jsp_codec* c);

/* IsKeyFrame(data) (IVideoCodec.hx:21; MSVideo1.hx:226-259,395-427; ScreenPressor.hx:96-101).
 * Pure host-side scan; returns 0/1. */
int jsp_is_key_frame(jsp_codec* c, const uint8_t* src, size_t n);

/* State() / ContinueI() (IVideoCodec.hx:22,25).  The reference never reports in_progress
 * (resumable I-decode is disabled, ScreenPressor.hx:210-215,277-285); both return zero_state. */
int jsp_state(jsp_codec* c);
int jsp_continue_i(jsp_codec* c);

/* DecompressI(src,dst):DecoderState (IVideoCodec.hx:24; MSVideo1.hx:62-67;
 * ScreenPressor.hx:117-295).  Returns JSP_ZERO_STATE or JSP_ERROR_OCCURED. */
int jsp_decompress_i(jsp_codec* c, const uint8_t* src, size_t n, int32_t* dst);

/* DecompressP(src,dst):PFrameResult (IVideoCodec.hx:26; MSVideo1.hx:106-209,293-393;
 * ScreenPressor.hx:302-484).  *data_pnt = the old previous frame ("no change", dst not adopted)
 * or dst; *significant_changes = 0/1.  Returns JSP_ZERO_STATE, or JSP_ERROR_OCCURED where the
 * reference would raise (e.g. an MSVideo1 skip code before any frame was decoded). */
int jsp_decompress_p(jsp_codec* c, const uint8_t* src, size_t n, int32_t* dst,
                     int32_t** data_pnt, int* significant_changes);

/* NeedsIndex() (IVideoCodec.hx:27; MSVideo1.hx:221-224 -> 1; ScreenPressor.hx:486-489 -> 0). */
int jsp_needs_index(jsp_codec* c);

/* Thread-local description of the last failure on this thread ("" if none). */
const char* jsp_last_error(void);

/* ---- frame pool in HBM (Manager.hx:114-118: num_buffers+1 frame buffers) ---------------- */

/* A pool of 32 frames or more (what batch decoding writes into: tile j of every frame at about the same time, or every frame of a clip one after
 * the other from the same workgroups) is PLACED.  What the decode kernels' stores get from a pool depends on where its frames lie relative to each
 * other: frames that are neighbours in memory cost the frame-walking kernels a sixth and can cost the batch kernels a quarter (no query
 * reveals it).  The pool therefore allocates, in one run, four times the 16-frame chunks it needs, takes every fourth (candidate k: chunks k, k + 4, ...)
 * and DEALS its frames round-robin over them — buffer i and buffer i + 1 never lie in the same chunk —, measures the candidate with the kernels' store
 * shape (a few milliseconds), keeps the first that takes what a plain fill takes and gives the other chunks back; only when none of the four comes
 * near do the older forms (one allocation, two frames per allocation, one per frame — frames taken in a strided order) get a try; boards differ in
 * which form their memory likes, so the form that won the last probe of the process is tried first by the next pool.  Decode consecutive
 * frames into consecutive buffers of the pool and they are far apart.  JSP_POOL_PROBE=0 in the environment: one allocation per frame, first come — what
 * smaller pools (a player's num_buffers + 1) always get.  The probe's appetite is bounded (jsp_pool_probe_info; while it chooses it holds four times
 * the pool, or what JSP_POOL_PROBE_HOLD_GB / a quarter of the free memory allows). */
jsp_pool* jsp_pool_create(int device_id, int width, int height, int nbuf);
int32_t* jsp_pool_buffer(jsp_pool* p, int i); /* device pointer, width*height ints, zeroed */
/* GB/s the chosen allocation took from the probe (0: a pool that is not probed); *attempts = allocations tried. */
double jsp_pool_store_rate(jsp_pool* p, int* attempts);
/* What placing the pool cost: wall time of the probe, the most device memory it held at one time (rejected candidates are kept until it
 * has chosen) and what it was allowed to hold — a quarter of the device memory free when it began, JSP_POOL_PROBE_HOLD_GB (GB) if lower,
 * never more than JSP_POOL_PROBE_MAX candidates (default 16).  All 0 for a pool that is not probed.  Returns 0, -1 for a null pool. */
int jsp_pool_probe_info(jsp_pool* p, double* probe_ms, uint64_t* held_peak_bytes, uint64_t* hold_limit_bytes);
/* What every candidate the probe measured took (GB/s, in the order tried: first the chunked candidates, then the older forms): up to `cap` of them
 * into `rates`; returns how many were measured (0 for a pool that is not probed, -1 for a null pool). */
int jsp_pool_probe_rates(jsp_pool* p, double* rates, int cap);
int jsp_pool_count(jsp_pool* p);
void jsp_pool_destroy(jsp_pool* p);
/* Copy one frame between a device frame buffer and host memory (parity checks, display). */
int jsp_download(const int32_t* device_frame, int32_t* host, size_t npixels);
int jsp_upload(int32_t* device_frame, const int32_t* host, size_t npixels);

/* ---- batched / resident-input entry points (extension; same semantics as the calls above) -- */

/* Run the codec's HIP work on `hip_stream` (a hipStream_t, e.g. torch's current stream) instead
 * of the codec's own stream.  NULL restores the codec's own stream. */
int jsp_set_stream(jsp_codec* c, void* hip_stream);
/* Codec options.  Returns 0 when accepted, -1 for an unknown key/value.
 *   "msv1_parse" = "gpu" (default) | "host" : MSVideo1 only.  "gpu" builds the per-block descriptor
 *       table with the on-GPU parse kernels (raw frame bytes are all the device needs; a replay of a
 *       staged batch re-runs the parse); frames the parse flags as special fall back to the host
 *       parser one by one, so results are identical either way.
 *   "msv1_parse_ahead" = "on" (default) | "off" : MSVideo1 only, replays of a staged batch of inter frames (jsp_staged_decode called again on the
 *       same batch).  Such a replay is a table-writing parse launch and the reconstruction launches that read the tables; "on" queues the NEXT
 *       replay's parse on a second stream of the codec, into a second set of tables, beside this replay's reconstruction (jsp_sync waits for both
 *       streams).  Costs a second table set (4 bytes per block and frame).  Results do not depend on it.
 *   "msv1_compact_tables" = "off" (default) | "on" : MSVideo1 only, the same replays.  "on": the table-writing parse leaves 2 bytes per block and a
 *       4-byte base per 256 blocks (the block's code offset modulo 32 768; a group's codes lie within 4 608 bytes of each other) instead of 4 bytes per
 *       block, and the temporal launch expands them as it stages them — half the table bytes written and read.  Used when every launch of the batch that
 *       reads tables is a temporal launch.  Off by default: it saves a twentieth of the step's memory traffic and half the table memory, and no time.
 *       Results do not depend on it.
 *   ("sp_group_chunk" and "msv1_parse_pieces", launch plans of round 4 that measured slower and were removed in round 5, are still accepted and do
 *   nothing — results never depended on them; their environment twins JSP_SP_GROUP_CHUNK / JSP_MSV1_PARSE_PIECES are no longer read.)
 *   "sp_band_rows" = "auto" (default) | "0" | "<n>" : ScreenPressor only.  Key frames are rebuilt by one
 *       workgroup per band of n rows (0 = the whole frame is one band; auto = sized so a batch fills
 *       the GPU); the host stage hands each band the row above it.  Results do not depend on it.
 *   "sp_host_threads" = "auto" (default: up to 8) | "1".."64" : ScreenPressor only, jsp_stage_batch.  A coded key frame renews
 *       every bit of decoder state, so the frames from one coded key frame up to the next depend on nothing before them:
 *       the host entropy stage takes up to this many such groups of a batch side by side, a decoder and a host thread each
 *       (one-frame calls have nothing to split).  Results do not depend on it.
 *   "sp_inter_fusion" = "on" (default) | "off" : ScreenPressor only.  In a staged batch, consecutive inter
 *       frames are rebuilt by ONE launch (pixels carried in registers from frame to frame; the host stage
 *       hands motion rectangles over as literal pixels; a frame that moves more than a quarter of its
 *       pixels keeps its motion blocks and a launch of its own).  "off": one launch per frame.
 *   "sp_forget_buffers" = "1" : ScreenPressor only; nothing may be in flight.  CONTRACT behind it: an inter frame may read ONE pixel per row of
 *       its destination before writing it (ScreenPressor.hx:436-449: "left of column 0" is the last pixel of the row above, which this
 *       frame has not reached), so the codec remembers the last column of every picture it decoded into a buffer and asks the device for the
 *       column of a buffer it has never written — a synchronous copy, ordered after work on blocking streams only: the caller's own writes
 *       to that buffer must be complete.  A caller that writes into frame buffers itself BETWEEN decodes (jsp_upload, a clear, another codec
 *       sharing the pool, a buffer freed and allocated again at the same address) says so with this option and every buffer is asked for
 *       again; without such writes nothing needs to be said.  Buffers of a frame that failed are forgotten by the codec itself.
 *   "msv1_async_pairs" = "on" (default) | "off" : MSVideo1 with "msv1_parse" = "gpu", asynchronous calls, frames of up to 128 parse tiles.
 *       on: such a frame is not launched at once but HELD until half of "async_depth" frames (at most 4) have been submitted, and they go
 *       out in ONE launch: all of them load, parse and reach their verdicts side by side, each paints when the frame in front is through
 *       (it may copy from its pixels and be compared with them) and is left unpainted — for the host's re-run — when the frame in front
 *       was.  Whatever is held goes out at once when one of the held frames is waited for, or when anything else needs the stream
 *       (jsp_sync, a synchronous call, jsp_prefetch, jsp_set_stream).  One player stream is bound by the chain of its frames' kernels:
 *       64 -> 94 Gpixels/s at 1080p with 8 frames in flight.  jsp_counter(c, "paired_frames") counts the frames that shared a launch.
 *       Results do not depend on it. */
/*   "msv1_async" = "auto" (default) | "one_launch_dma" | "one_launch" | "two_launches" : MSVideo1 with "msv1_parse" = "gpu",
 *       asynchronous calls only; frames of up to 128 parse tiles (2 MiB).  auto: one_launch_dma while at most 3 codec instances
 *       of the process use this path, one_launch beyond (many streams: the copy queues are the bottleneck).  one_launch_dma: the copy engine brings the frame's bytes up on a
 *       stream of its own (next to the previous frame's kernel), ONE launch parses, waits until every tile of the frame
 *       has reported what the host parser would have found, and rebuilds the frame — or leaves `dst` untouched for the
 *       synchronous re-run.  one_launch: the same launch reads the bytes from the caller's pinned memory itself (no copy
 *       queued at all; the bus transfer then sits inside the kernel).  two_launches: a scout launch, then the decode
 *       launch it may veto (what larger frames always get).  Results do not depend on it. */
/*   "key_frame_compare" = "off" (default) | "<first row>" : any codec; nothing may be in flight.  With a first row set (Manager uses
 *       INSIGNIFICANT_LINES = 36), every key frame is also compared with the previous frame as the call found it — the pixel loop of
 *       frames_differ_significantly (Manager.hx:413-419: any dst[i] != prev[i], i >= row * width) — without a pass of the caller's
 *       own: ScreenPressor's host stage holds both pictures and answers itself (synchronous calls), otherwise the compare is queued on
 *       the codec's stream right behind the frame's kernels (no extra wait; the frame is still in the Infinity Cache).  The answer:
 *       jsp_key_frame_differs() after a synchronous DecompressI; *significant_changes of jsp_wait for an asynchronous one (a key
 *       frame that decoded and has no previous frame to be compared with counts as a change: Manager.hx:399-411; for a frame that FAILED
 *       *significant_changes stays what the decode reported and jsp_key_frame_differs() says -1).  Staged batches are not compared. */
/*   "async_depth" = "1".."16" (default "4") : any codec.  Frames that may be in flight between jsp_decompress_*_async and
 *       jsp_wait. */
int jsp_set_option(jsp_codec* c, const char* key, const char* value);
/* Block until everything queued by this codec has finished (frames in flight on the asynchronous path stay to be
 * collected with jsp_wait). */
int jsp_sync(jsp_codec* c);
/* The last key frame decoded through jsp_decompress_i, or collected with jsp_wait, against the frame before it (option
 * "key_frame_compare"): 1 differs, 0 does not, -1 nothing to compare with (no previous frame, the frame failed, option off). */
int jsp_key_frame_differs(jsp_codec* c);
/* Diagnostics (no reference counterpart): how often this codec instance took one of its slow paths since it was created.
 *   "async_reruns"       frames of the asynchronous per-frame calls that were re-run through the synchronous path (the GPU
 *                        alone could not settle the stream — short streams, end markers, skip codes without a previous
 *                        frame — or the frame's tiles did not report in time);
 *   "lookback_fallbacks" staged MSVideo1 batches re-run through the descriptor kernels after a tile gave up waiting for
 *                        the tiles before it;
 *   "sp_groups_held", "sp_spare_decoders"  (ScreenPressor) groups of pictures the asynchronous path keeps a record of, and
 *                        host decoders on its shelf: both stay bounded however long a stream runs without jsp_sync.
 * Unknown names and null arguments answer -1.  Results never depend on either path having been taken. */
long long jsp_counter(jsp_codec* c, const char* name);

/* ---- asynchronous per-frame calls: the `_async` variant SURVEY.md 8(b) allows for the one-call-per-tick surface
 * (Manager.hx:507,511) --------------------------------------------------------------------------------------------
 * jsp_decompress_i_async / jsp_decompress_p_async run the frame's HOST stage, queue its uploads and kernels and return
 * at once with a ticket; jsp_wait(ticket) blocks until that frame is complete and hands back exactly what the
 * synchronous call would have returned (DecoderState, *data_pnt, *significant_changes).  So the host stage of frame
 * n+1 (entropy decode, parse pre-scan, copy into pinned memory) overlaps the uploads and kernels of frame n.
 *   - device frame buffers only; tickets must be waited for in submission order, and a frame is complete when its ticket has been
 *     waited for.  The GPU work of DIFFERENT groups of pictures (ScreenPressor with worker threads: a coded key frame opens a
 *     group) is queued in the order their host stages finish, not in submission order — so the buffer a frame in flight is decoded
 *     AGAINST (the previous frame at its submission) must not be handed out as `dst` of a later frame until the frame reading it
 *     has been waited for, just like its own `dst` (examples/jsp_play and jsplayer_amd/player.py keep both out of circulation);
 *   - at most "async_depth" frames (jsp_set_option, default 4, 1..16) may be in flight: a further submission fails;
 *   - `src` must stay valid and unchanged, and `dst` untouched, until the frame's ticket has been waited for; if `src`
 *     lies in memory from jsp_host_alloc (pinned), it is uploaded from where it is, without a staging copy;
 *   - jsp_previous_frame() answers for the last SUBMITTED frame (adoption is decided by the host stage);
 *   - a frame the GPU cannot settle alone (MSVideo1: truncated stream, 8-bit end marker, skip code with nothing to
 *     copy from) is transparently re-run through the synchronous path inside jsp_wait, together with the frames
 *     submitted after it. */
int jsp_decompress_i_async(jsp_codec* c, const uint8_t* src, size_t n, int32_t* dst, uint64_t* ticket);
int jsp_decompress_p_async(jsp_codec* c, const uint8_t* src, size_t n, int32_t* dst, uint64_t* ticket);
int jsp_wait(jsp_codec* c, uint64_t ticket, int32_t** data_pnt, int* significant_changes);
/* Pinned host memory for compressed frames (what an AVI reader fills): uploads from it need no staging copy. */
void* jsp_host_alloc(size_t bytes);
void jsp_host_free(void* p);
/* The next frames' bytes lie in [host, host + bytes) (a stretch of the file the reader holds, chunk headers and all): the codec may
 * take the whole range to the device in ONE copy on a stream of its own, and asynchronous frames submitted afterwards whose `src`
 * lies inside it then queue no upload of their own (MSVideo1 with "msv1_parse" = "gpu"; everything else
 * accepts the call and does nothing).  No counterpart in the reference: its Manager hands the decoder slices of the one ArrayBuffer
 * the loader filled (DataLoader.hx), and this is that buffer crossing the bus in pieces sized for the bus instead of frame by frame
 * (a megabyte per copy goes at 24 - 39 GB/s here, 64 MB at 57).  Returns at once.  The codec keeps the 4 most recent ranges; a
 * range must stay unchanged in host memory while it is kept (frames are pre-scanned on the host from `src` itself); host == NULL,
 * bytes == 0 gives every range up (do so before the memory is reused for other bytes).  Results never depend on it.
 * jsp_counter(c, "prefetched_frames") counts the frames that found their bytes on the device. */
int jsp_prefetch(jsp_codec* c, const void* host, size_t bytes);

/* Equivalent to calling DecompressI on frames 0..n-1 in order (Manager.hx:507 in a loop); device
 * `dsts` only.  Key-frame-only MSVideo1 batches decode in ONE launch (grid.y = frame). */
int jsp_decompress_i_batch(jsp_codec* c, int nframes, const uint8_t* const* srcs,
                           const size_t* lens, int32_t* const* dsts);

/* Two-step form of the batch above, for measuring with inputs resident in HBM:
 *   jsp_stage_batch   host stage (parse / entropy -> descriptor tables) + H2D, untimed by callers;
 *   jsp_staged_decode queues the reconstruction kernels for the whole batch (asynchronous:
 *                     follow with jsp_sync or events on the stream given to jsp_set_stream);
 * `is_key[i]` selects DecompressI (non-zero) or DecompressP (zero) semantics for frame i
 * (NULL = all key frames).  Staging advances the codec's host-side state (entropy models,
 * previous-frame chain) exactly as the per-frame calls would; a staged batch may be decoded
 * any number of times into the same `dsts`. */
jsp_staged* jsp_stage_batch(jsp_codec* c, int nframes, const uint8_t* const* srcs,
                            const size_t* lens, const uint8_t* is_key, int32_t* const* dsts);
/* The same into an existing batch object, whose pinned and device buffers are taken over (a caller that stages batch after
 * batch: no allocation per batch).  Every decode of `reuse` must have finished (jsp_sync).  Returns the batch to use from
 * now on — `reuse` itself, or a new object when `reuse` was of a kind that cannot hold this batch (then it has been
 * destroyed) —, NULL on error (`reuse` stays valid). */
jsp_staged* jsp_restage_batch(jsp_codec* c, jsp_staged* reuse, int nframes, const uint8_t* const* srcs,
                              const size_t* lens, const uint8_t* is_key, int32_t* const* dsts);
int jsp_staged_decode(jsp_codec* c, jsp_staged* s);
void jsp_staged_destroy(jsp_staged* s);

/* Accounting for a staged batch (bench.py roofline): */
typedef struct jsp_staged_info {
    uint64_t frames;
    uint64_t pixels;           /* width*height*frames */
    uint64_t stream_bytes;     /* compressed bytes consumed (S in SURVEY.md 8d) */
    uint64_t descriptor_bytes; /* bytes of host-built tables resident in HBM */
    uint64_t units_coded;      /* MSVideo1: coded 4x4 blocks; ScreenPressor: data pixels */
    uint64_t units_copied;     /* MSVideo1: skipped blocks; ScreenPressor: pixels fetched from prev */
    uint64_t runs;             /* ScreenPressor run descriptors (R) */
    uint64_t algorithmic_bytes;/* SURVEY.md 8(d) formula for this batch */
    uint64_t kernel_launches;  /* launches jsp_staged_decode issues */
    double host_stage_ms;      /* wall time of the host parse / entropy stage */
    double h2d_ms;             /* wall time of the uploads */
    double device_parse_ms;    /* wall time of the on-GPU parse at staging (0 with the host parser) */
    uint64_t moved_bytes;      /* bytes the launch plan has to move through HBM at the least: every table and stream
                                  byte read once, every destination pixel written once, the previous frame read once
                                  per launch that carries pixels in registers from frame to frame.  Below
                                  algorithmic_bytes for those launches (the SURVEY.md formula charges a previous-frame
                                  read per frame), above it where host-built tables add bytes the formula leaves out. */
} jsp_staged_info;
int jsp_staged_get_info(const jsp_staged* s, jsp_staged_info* out);

/* Names of the kernels jsp_staged_decode launches for this batch, in launch order, each name once, separated by
 * " + " (what a rocprofv3 kernel trace of the decode shows).  Valid until the batch is destroyed. */
const char* jsp_staged_kernels(const jsp_staged* s);

/* Per-frame results of a staged batch: status[i] (DecoderState), adopted[i] (1 if dsts[i] became
 * the previous frame), significant[i] (valid after jsp_staged_decode + jsp_sync). */
int jsp_staged_results(jsp_staged* s, int* status, int* adopted, int* significant);

/* ---- what sits right after the codec in the reference's Manager, on the GPU --------------- */

/* Manager.fill_bitmap_data (Manager.hx:325-390): RGB32 frame -> canvas pixels.  Modes: */
enum {
    JSP_DISPLAY_CANVAS = 0,        /* 0xFF000000 | B<<16 | G<<8 | R              (Manager.hx:379) */
    JSP_DISPLAY_CANVAS_RGB15 = 1,  /* 0xFF000000 | c << 3  (ScreenPressor 16 bpp)  (Manager.hx:370) */
    JSP_DISPLAY_SETPIXELS = 2,     /* 0xFF000000 | c                               (Manager.hx:351) */
    JSP_DISPLAY_SETPIXELS_RGB15 = 3 /* c << 11                                     (Manager.hx:340) */
};
/* `frame`, `out`: device pointers, width*height ints.  flip_rows != 0 also undoes the bottom-up row
 * order (the reference leaves that to its display matrix, Main.hx:318).  Asynchronous on `hip_stream`. */
int jsp_display_convert(const int32_t* frame, int32_t* out, int width, int height, int mode, int flip_rows,
                        void* hip_stream);
/* The pixel loop of frames_differ_significantly (Manager.hx:413-419): *differ = any a[i] != b[i] for
 * first_pixel <= i < npixels.  Device pointers; synchronous. */
int jsp_frames_differ(const int32_t* a, const int32_t* b, size_t first_pixel, size_t npixels, int* differ,
                      void* hip_stream);

/* ---- streams sharded one per GPU inside ONE process (the caller side of Manager.hx:97-142: a Manager and a decoder per
 * stream).  Streams are independent — a codec instance, its previous-frame chain and its entropy models each — so stream i simply
 * lives on devices[i mod ndev] (jsp_codec_create / jsp_pool_create take the device; every call activates its codec's device), one
 * host thread per stream, and no frame ever crosses xGMI.  The one collective is the sum of the per-device counters. */
int jsp_device_count(void);                                             /* HIP devices visible to the process (0: none) */
int jsp_assign_stream(int stream_index, const int* devices, int ndev);  /* devices[stream_index mod ndev]; -1 on bad arguments */
/* per_device = ndev pairs (frames, pixels), entry i belonging to devices[i]; total[0..1] = their sums.  When librccl can be loaded
 * the sums are ALSO computed on the GPUs — one RCCL communicator per distinct device, ncclAllReduce(ncclSum) of the two counters
 * over xGMI — and must agree with the host's (*via_rccl = 1; a mismatch is JSP_ERROR_OCCURED); without RCCL, or when it declines
 * (jsp_shard_last_error says why), *via_rccl = 0 and the host's sums stand.  Returns JSP_ZERO_STATE or JSP_ERROR_OCCURED. */
int jsp_reduce_counters(const int* devices, int ndev, const uint64_t* per_device, uint64_t* total, int* via_rccl);
const char* jsp_shard_last_error(void);

/* Library/build identification: "jsplayer_amd <version> gfx950". */
const char* jsp_version(void);

#ifdef __cplusplus
}
#endif
#endif /* JSPLAYER_AMD_H */
