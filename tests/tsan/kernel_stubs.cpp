// TEST INFRASTRUCTURE (see hip/hip_runtime.h in this directory): what the .hip files give the host layers, without a GPU.  Launches paint
// nothing; what the HOST PROTOCOL reads back from a launch is produced here so that frames settle the way they do on a GPU: the on-GPU parse's
// per-frame counters come from the product's own host parser, a one-launch frame's report says "go" — or, every seventh frame, "the host parser
// has to settle it", which sends the frame and the frames queued behind it through the re-run path.
#include <atomic>
#include <cstring>
#include <vector>

#include "../../jsplayer_amd/csrc/codec.h"
#include "../../jsplayer_amd/csrc/msv1.h"
#include "../../jsplayer_amd/csrc/sp.h"

namespace jsp {

uint32_t msv1_parse_tile_bytes() { return 16384; }
uint32_t msv1_small_tile_bytes() { return 8192; }

void msv1_launch_parse(const Msv1Geometry& geo, const uint8_t* d_stream, const Msv1ParseFrame* d_frames, int nframes, const uint32_t*, int, int, uint32_t*,
                       uint32_t*, uint32_t*, uint32_t* d_desc, Msv1FrameInfo* d_info, int insignificant_blocks, hipStream_t stream) {
    std::vector<uint8_t> rows((size_t)geo.nby + 1);
    for (int i = 0; i < nframes; ++i) {
        const Msv1ParseFrame& f = d_frames[i];
        if (f.host_parsed) continue;
        Msv1Parse out;
        msv1_parse(geo, d_stream + f.beg, f.end - f.beg, true, 0, insignificant_blocks, f.beg, d_desc + f.desc_base, rows, out);
        Msv1FrameInfo fi{};
        fi.n_coded = (uint32_t)out.n_coded;
        fi.n_skip_codes = (uint32_t)out.n_skipped;
        fi.total_blocks = (uint32_t)(out.n_coded + out.n_skipped);
        fi.flags = out.s1 ? MSV1_INFO_S1 : 0u;
        fi.consumed = (uint32_t)out.consumed;
        d_info[i] = fi;
    }
    stub_stream_work(stream);
}

void msv1_launch_fused(const Msv1Geometry&, const uint8_t*, const Msv1TileRec*, const int32_t*, unsigned long long*, uint32_t, uint32_t, int, uint32_t*,
                       hipStream_t stream, Msv1AsyncInfo* d_info, int, int mode, uint32_t bad_mask, uint32_t* d_poison, const Msv1TileRec*, Msv1AsyncInfo* h_info,
                       uint32_t want, uint8_t*, bool, const Msv1Riders* riders) {
    static std::atomic<unsigned> n{0};
    auto settle = [&](Msv1AsyncInfo* dev, Msv1AsyncInfo* host, uint32_t w, uint32_t bad) {
        if (!dev) return;
        const bool vetoed_before = d_poison && *d_poison;
        const uint32_t flags = (n.fetch_add(1, std::memory_order_relaxed) % 7 == 6) ? MSV1_ASYNC_SHORT : 0u;
        dev->arrived = dev->finished = w;
        if (host) {
            Msv1AsyncInfo r{};
            r.flags = flags;
            *host = r;
        } else
            dev->flags |= flags;
        if (((flags & bad) || vetoed_before) && d_poison) *d_poison = 1u;
    };
    if (mode == 3) {
        settle(d_info, h_info, want, bad_mask);
        if (riders)
            for (uint32_t i = 0; i < riders->count; ++i) settle(riders->f[i].info, riders->f[i].host_info, riders->f[i].want, riders->f[i].bad_mask);
    } else if (mode == 1 && d_info) {
        settle(d_info, nullptr, want, bad_mask);
    }
    stub_stream_work(stream);
}

void msv1_launch_blocks(const Msv1Geometry&, const uint8_t*, const uint32_t*, const Msv1FrameArgs*, int, const int32_t*, bool, hipStream_t s) { stub_stream_work(s); }
void msv1_launch_blocks_temporal(const Msv1Geometry&, const uint8_t*, const uint32_t*, const Msv1FrameArgs*, int, const int32_t*, hipStream_t s, const uint16_t*, const uint32_t*) { stub_stream_work(s); }
void msv1_launch_tables_compact(const Msv1Geometry&, const uint32_t*, uint16_t*, uint32_t*, const uint32_t*, int, hipStream_t s) { stub_stream_work(s); }
void msv1_launch_edge_compare(const Msv1Geometry&, const Msv1FrameArgs*, int, hipStream_t s) { stub_stream_work(s); }
void launch_frames_differ(const int32_t*, const int32_t*, size_t, size_t, uint32_t* d_flag, hipStream_t s) { if (d_flag) *d_flag = 0; stub_stream_work(s); }
double pool_store_rate(uint32_t* const*, int, int, int, uint32_t) { return 6900.0; }
double pool_fill_rate(uint32_t*, size_t) { return 6900.0; }

namespace sp {
int choose_band_rows(const Geometry& g, int nframes) {
    if (nframes <= 0) return 0;
    const int want = (3072 + nframes - 1) / nframes;
    int rows = (g.Y + want - 1) / (want > 0 ? want : 1);
    if (rows < 24) rows = 24;
    return rows >= g.Y ? 0 : rows;
}
size_t iframe_lds_bytes(const Geometry&, int) { return 32768; }
int iframe_tile_max_band_rows() { return 4096; }
int iframe_tile_span(const Geometry&) { return 256; }
bool iframe_tiles_ok(const Geometry& g) { return (g.X & 3) == 0 && g.aligned16; }
void launch_iframe_tiles(const Geometry&, const IFrameArgs*, int, int, hipStream_t s) { stub_stream_work(s); }
void launch_iframes(const Geometry&, const IFrameArgs*, int, int, hipStream_t s) { stub_stream_work(s); }
void launch_pframe(const Geometry&, int32_t*, const int32_t*, const PBlock*, const uint32_t*, hipStream_t s) { stub_stream_work(s); }
void launch_pframe_group(const Geometry&, const PGroupFrame*, int, const int32_t*, const PBlock*, const uint32_t*, bool, hipStream_t s) { stub_stream_work(s); }
}  // namespace sp
}  // namespace jsp
