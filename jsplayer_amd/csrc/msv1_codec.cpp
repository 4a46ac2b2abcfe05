// MSVideo1_16bit / MSVideo1_8bit behind the IVideoCodec-shaped C ABI.
// Host side = control flow of MSVideo1.hx:106-209 / 293-393 (early-outs, changes, block_changes,
// adoption of dst as prevFrame); pixels are produced by msv1_kernels.hip.
#include <algorithm>
#include <unordered_set>

#include "codec.h"
#include "msv1.h"

namespace jsp {
namespace {

struct Msv1Codec;

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
    };
    std::vector<Group> groups;
    bool need_signif = false;  // some frame asked for the stage-2 compare

    void decode(hipStream_t stream) override {
        if (nframes == 0) return;
        if (need_signif) JSP_HIP(hipMemsetAsync(d_signif.p, 0, sizeof(uint32_t) * nframes, stream));
        const auto* frames = static_cast<const Msv1FrameArgs*>(d_frames.p);
        for (const Group& g : groups) {
            msv1_launch_blocks(geo, static_cast<const uint8_t*>(d_stream.p),
                               static_cast<const uint32_t*>(d_desc.p), frames + g.first, g.count,
                               d_palette, vec_ok, stream);
            if (g.edge_compare) msv1_launch_edge_compare(geo, frames + g.first, g.count, stream);
        }
        JSP_HIP(hipGetLastError());
        if (need_signif)
            JSP_HIP(hipMemcpyAsync(h_signif.p, d_signif.p, sizeof(uint32_t) * nframes, hipMemcpyDeviceToHost,
                                   stream));
        decoded = true;
    }
};

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
        return (X & 3) || (Y & 3) || geo.bits == 8;
    }

    jsp_staged* stage(const std::vector<jsp_frame_in>& frames, jsp_staged* reuse) override {
        activate();
        const double t0 = now_ms();
        auto* st = dynamic_cast<Msv1Staged*>(reuse);
        if (!st) st = new Msv1Staged();
        std::unique_ptr<Msv1Staged> guard(reuse ? nullptr : st);
        const int nf = (int)frames.size();
        st->geo = geo;
        st->nframes = nf;
        st->decoded = false;
        st->groups.clear();
        st->status.assign(nf, JSP_ZERO_STATE);
        st->adopted.assign(nf, 0);
        st->significant.assign(nf, 0);
        st->info = jsp_staged_info{};
        if (geo.bits == 8 && !d_palette.p) {  // Preinit not called: all-zero palette
            d_palette.reserve(sizeof palette);
            JSP_HIP(hipMemcpy(d_palette.p, palette, sizeof palette, hipMemcpyHostToDevice));
        }
        st->d_palette = static_cast<const int32_t*>(d_palette.p);

        size_t total_stream = 0;
        for (const auto& f : frames) total_stream += (f.n + 1) & ~size_t(1);
        if (total_stream + 64 > 0xFFFFFFF0u) throw std::runtime_error("batch stream exceeds 4 GiB");
        st->h_stream.reserve(total_stream + 64);
        st->h_desc.reserve(sizeof(uint32_t) * (size_t)std::max(geo.nblocks, 1) * nf);
        st->h_frames.reserve(sizeof(Msv1FrameArgs) * std::max(nf, 1));
        st->d_signif.reserve(sizeof(uint32_t) * std::max(nf, 1));
        st->h_signif.reserve(sizeof(uint32_t) * std::max(nf, 1));
        auto* h_stream = static_cast<uint8_t*>(st->h_stream.p);
        auto* h_desc = static_cast<uint32_t*>(st->h_desc.p);
        auto* h_frames = static_cast<Msv1FrameArgs*>(st->h_frames.p);
        auto* d_signif = static_cast<uint32_t*>(st->d_signif.p);

        bool vec_ok = (X & 3) == 0;
        size_t off = 0;
        std::unordered_set<const void*> group_dsts;
        bool group_closed = true;  // true: the next frame must open a new launch group
        for (int i = 0; i < nf; ++i) {
            const jsp_frame_in& f = frames[i];
            if (f.n) std::memcpy(h_stream + off, f.src, f.n);
            uint32_t* desc = h_desc + (size_t)i * geo.nblocks;
            Msv1Parse pr;
            msv1_parse(geo, f.src, f.n, prev_dev != nullptr, size_of_just_skips, insignificant_blocks,
                       (uint32_t)off, desc, block_changes, pr);
            Msv1FrameArgs& fa = h_frames[i];
            fa.dst = f.dst;
            fa.prev = prev_dev;
            fa.signif = d_signif + i;
            fa.stream_end = (uint32_t)(off + f.n);
            fa.desc_base = (uint32_t)((size_t)i * geo.nblocks);
            fa.cmp_row_lo = 0xFFFFFFFFu;
            fa.pad = 0;
            off += (f.n + 1) & ~size_t(1);

            if ((reinterpret_cast<uintptr_t>(f.dst) & 15) || (reinterpret_cast<uintptr_t>(prev_dev) & 15))
                vec_ok = false;
            bool dependent = pr.n_skipped != 0;  // reads its predecessor
            if (pr.early_out) {
                // nothing to paint: every block untouched, the frame rides along as a no-op
                std::fill(desc, desc + geo.nblocks, MSV1_DESC_UNTOUCHED);
            } else if (pr.aborted) {
                st->status[i] = JSP_ERROR_OCCURED;  // the reference raises out of DecompressP here
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
                if (pr.changes) {
                    st->adopted[i] = 1;
                }
            }
            st->info.units_coded += pr.n_coded;
            st->info.units_copied += pr.n_skipped;
            st->info.stream_bytes += pr.consumed;

            // launch grouping: frames that read nothing join the running group unless their dst
            // is already written by it; a frame that reads its predecessor gets its own launch
            const bool edge = fa.cmp_row_lo != 0xFFFFFFFFu && ((X & 3) || (Y & 3));
            const bool writes = !pr.early_out;
            if (dependent || group_closed || (writes && group_dsts.count(f.dst))) {
                st->groups.push_back({i, 1, edge});
                group_dsts.clear();
                group_closed = dependent;  // nothing may share a launch with a frame that reads
            } else {
                st->groups.back().count++;
            }
            if (writes) group_dsts.insert(f.dst);
            if (st->adopted[i]) prev_dev = f.dst;
        }
        st->vec_ok = vec_ok;
        st->need_signif = false;
        for (int v : st->significant) st->need_signif |= v < 0;
        st->info.frames = nf;
        st->info.pixels = (uint64_t)X * Y * nf;
        st->info.descriptor_bytes = sizeof(uint32_t) * (uint64_t)geo.nblocks * nf + sizeof(Msv1FrameArgs) * nf;
        // A = S + 64*N_coded + 128*N_skipped
        st->info.algorithmic_bytes = st->info.stream_bytes + 64 * st->info.units_coded + 128 * st->info.units_copied;
        st->info.kernel_launches = st->groups.size();
        st->info.host_stage_ms = now_ms() - t0;

        const double t1 = now_ms();
        st->d_stream.reserve(total_stream + 64);
        st->d_desc.reserve(sizeof(uint32_t) * (size_t)std::max(geo.nblocks, 1) * nf);
        st->d_frames.reserve(sizeof(Msv1FrameArgs) * std::max(nf, 1));
        if (nf) {
            JSP_HIP(hipMemcpyAsync(st->d_stream.p, h_stream, total_stream, hipMemcpyHostToDevice, stream));
            JSP_HIP(hipMemcpyAsync(st->d_desc.p, h_desc, sizeof(uint32_t) * (size_t)geo.nblocks * nf,
                                   hipMemcpyHostToDevice, stream));
            JSP_HIP(hipMemcpyAsync(st->d_frames.p, h_frames, sizeof(Msv1FrameArgs) * nf, hipMemcpyHostToDevice,
                                   stream));
            JSP_HIP(hipStreamSynchronize(stream));
        }
        st->info.h2d_ms = now_ms() - t1;
        guard.release();
        return st;
    }
};

}  // namespace
}  // namespace jsp

jsp_codec* jsp_make_msv1(int bits, int w, int h, const uint8_t* palette, int palette_bytes) {
    return new jsp::Msv1Codec(bits, w, h, palette, palette_bytes);
}
