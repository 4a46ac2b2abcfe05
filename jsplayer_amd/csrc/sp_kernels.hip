// ScreenPressor reconstruction kernels for gfx950 (MI355X).  Integer work, HBM-bound, no MFMA.
//
// sp_iframe_rows_kernel — one workgroup per I-frame (grid.x = frame; many frames fill the chip).
//   The run table resolves every pixel to either a constant or "the pixel one row up (same column or
//   one to the left), plus a per-run delta" (ScreenPressor.hx:242-273; the gradient predictor
//   telescopes inside a run).  Rows are produced top of the buffer downwards; the previous row lives
//   in LDS, so the row-to-row dependency never touches HBM: per row one coalesced read of the run
//   records that intersect it and one 16-byte store per lane.
// sp_pframe_kernel — P-frame: a workgroup covers 4 horizontally adjacent 16x16 blocks (64 px =
//   256 contiguous bytes per row), lane = 16-byte chunk of a row: unchanged / base copy / motion
//   from the previous frame in HBM, literal payload for data rectangles (ScreenPressor.hx:361-475).
#include "sp.h"

namespace jsp::sp {
namespace {

constexpr int IWG = 512;

__device__ __forceinline__ uint32_t add_bytes(uint32_t u, uint32_t d) {  // per byte, bytes 0..2
    return (((u & 0x00FF00FFu) + (d & 0x00FF00FFu)) & 0x00FF00FFu) | (((u & 0x0000FF00u) + (d & 0x0000FF00u)) & 0x0000FF00u);
}

__global__ __launch_bounds__(IWG) void sp_iframe_rows_kernel(const IFrameArgs* __restrict__ args, int X, int Y) {
    extern __shared__ __align__(16) uint32_t lds[];
    const IFrameArgs fa = args[blockIdx.x];
    uint32_t* __restrict__ dst = reinterpret_cast<uint32_t*>(fa.dst);
    const int tid = threadIdx.x;
    const size_t npx = (size_t)X * Y;
    if (fa.flat) {  // flat key frame: one colour (ScreenPressor.hx:132-155)
        for (size_t i = tid; i < npx; i += IWG) dst[i] = fa.colour;
        return;
    }
    // LDS plan: two row buffers (with one guard word in front for the x-1 access) + run staging
    const int rowcap = X + 4;
    uint32_t* rowbuf0 = lds;
    uint32_t* rowbuf1 = lds + rowcap;
    uint32_t* rstart = lds + 2 * rowcap;          // X + 2 run starts
    uint32_t* rword = rstart + (X + 2);           // X + 2 run words
    uint32_t* lastpix = rword + (X + 2);          // last pixel of each of the 4 most recent rows
    if (tid < 4) lastpix[tid] = 0;
    const bool vec = (X & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
    for (int y = 0; y < Y; ++y) {
        uint32_t* cur = (y & 1) ? rowbuf1 : rowbuf0;
        const uint32_t* up = (y & 1) ? rowbuf0 : rowbuf1;
        const uint32_t r0 = fa.row_run[y], r1 = fa.row_run[y + 1];
        const int nr = (int)(r1 - r0) + 1;        // runs r0..r1 (the last one may start in the next row)
        for (int k = tid; k < nr; k += IWG) {
            const IRun r = fa.runs[r0 + k];
            rstart[k] = r.start;
            rword[k] = r.word;
        }
        // pixel (X-1, y-2) for the x == 0 case of the above-left predictor (linear index i-X-1)
        const uint32_t wrap_left = y >= 2 ? lastpix[(y - 2) & 3] : 0u;
        __syncthreads();
        const uint32_t row0 = (uint32_t)((size_t)y * X);
        for (int x0 = tid * 4; x0 < X; x0 += IWG * 4) {
            // run holding pixel x0: last k with rstart[k] <= row0 + x0
            const uint32_t i0 = row0 + x0;
            int lo = 0, hi = nr - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (rstart[mid] <= i0) lo = mid; else hi = mid - 1;
            }
            int k = lo;
            uint32_t px[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int x = x0 + j;
                if (x < X) {
                    const uint32_t i = row0 + x;
                    while (k + 1 < nr && rstart[k + 1] <= i) ++k;
                    const uint32_t w = rword[k];
                    const uint32_t kind = w >> 24, val = w & 0xFFFFFFu;
                    uint32_t v;
                    if (kind == RUN_CONST) v = val;
                    else if (y == 0) v = 0;                                  // above the buffer: undefined -> 0
                    else if (kind == RUN_ABOVE) v = up[x];
                    else if (kind == RUN_ABOVE_PLUS) v = add_bytes(up[x], val);
                    else v = x > 0 ? up[x - 1] : wrap_left;               // RUN_ABOVE_LEFT
                    px[j] = v;
                    cur[x] = v;
                    if (x == X - 1) lastpix[y & 3] = v;
                } else
                    px[j] = 0;
            }
            if (vec) *reinterpret_cast<uint4*>(dst + row0 + x0) = make_uint4(px[0], px[1], px[2], px[3]);
            else
                for (int j = 0; j < 4 && x0 + j < X; ++j) dst[row0 + x0 + j] = px[j];
        }
        __syncthreads();
    }
}

constexpr int PWG = 256;  // 16 rows x 16 chunks of 4 pixels = 4 blocks side by side

__global__ __launch_bounds__(PWG) void sp_pframe_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ prev,
                                                        const PBlock* __restrict__ blocks,
                                                        const uint32_t* __restrict__ payload, int X, int Y, int nbx,
                                                        int vec) {
    const int ly = threadIdx.x >> 4;          // row inside the block row
    const int chunk = threadIdx.x & 15;       // 4-pixel chunk inside the 64-pixel span
    const int bx = blockIdx.x * 4 + (chunk >> 2);
    const int by = blockIdx.y;
    if (bx >= nbx) return;
    const int y = by * 16 + ly;
    const int x0 = bx * 16 + (chunk & 3) * 4;
    if (y >= Y || x0 >= X) return;
    const PBlock pb = blocks[(size_t)by * nbx + bx];
    const size_t npx = (size_t)X * Y;
    const size_t i0 = (size_t)y * X + x0;
    uint32_t px[4];
    const int cx0 = (chunk & 3) * 4;          // chunk origin relative to the block
    const bool row_in = ly >= pb.y1 && ly < pb.y2;
    const bool touched = pb.flags != 0 && row_in && cx0 < pb.x2 && cx0 + 4 > pb.x1;
    if (!touched) {
        if (vec && x0 + 4 <= X) {
            *reinterpret_cast<uint4*>(dst + i0) = *reinterpret_cast<const uint4*>(prev + i0);
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (x0 + j < X) dst[i0 + j] = prev[i0 + j];
        return;
    }
    const int w = pb.x2 - pb.x1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rx = cx0 + j;               // column relative to the block
        uint32_t v = 0;
        if (x0 + j < X) {
            if (rx >= pb.x1 && rx < pb.x2) {
                if (pb.flags & PB_MOTION) {
                    // linear index into the previous frame, no per-axis clipping (ScreenPressor.hx:400-405);
                    // outside the buffer reads as 0
                    const long jdx = (long)(y + pb.my) * X + (x0 + j + pb.mx);
                    v = (jdx >= 0 && (size_t)jdx < npx) ? prev[jdx] : 0u;
                } else {
                    v = payload[pb.payload + (uint32_t)((ly - pb.y1) * w + (rx - pb.x1))];
                }
            } else
                v = prev[i0 + j];             // base copy around a sub-rectangle
        }
        px[j] = v;
    }
    if (vec && x0 + 4 <= X) *reinterpret_cast<uint4*>(dst + i0) = make_uint4(px[0], px[1], px[2], px[3]);
    else
        for (int j = 0; j < 4; ++j)
            if (x0 + j < X) dst[i0 + j] = px[j];
}

}  // namespace

size_t iframe_lds_bytes(const Geometry& g) { return sizeof(uint32_t) * (2 * (size_t)(g.X + 4) + 2 * (size_t)(g.X + 2) + 4); }

void launch_iframes(const Geometry& g, const IFrameArgs* d_args, int nframes, hipStream_t stream) {
    if (nframes <= 0) return;
    const size_t lds = iframe_lds_bytes(g);
    static bool attr_set = false;
    if (!attr_set && lds > 48 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sp_iframe_rows_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(sp_iframe_rows_kernel, dim3(nframes), dim3(IWG), lds, stream, d_args, g.X, g.Y);
}

void launch_pframe(const Geometry& g, int32_t* dst, const int32_t* prev, const PBlock* d_blocks,
                   const uint32_t* d_payload, hipStream_t stream) {
    const int vec = ((g.X & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(prev) & 15) == 0) ? 1 : 0;
    dim3 grid((g.nbx + 3) / 4, g.nby);
    hipLaunchKernelGGL(sp_pframe_kernel, grid, dim3(PWG), 0, stream, reinterpret_cast<uint32_t*>(dst),
                       reinterpret_cast<const uint32_t*>(prev), d_blocks, d_payload, g.X, g.Y, g.nbx, vec);
}

}  // namespace jsp::sp
