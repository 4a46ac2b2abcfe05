// LAB build of msv1_fused_kernel (never part of the library): thread 0 of every tile of the batch form adds the cycles of each phase to
// counters of its own behind the tile tables (`want` = tiles of the batch); tools/lab/fused_clocks.sh builds with
//   make -C jsplayer_amd/csrc HOOKS=-I../../tools/lab/hooks_clocks
// -DJSP_FUSED_STOP=k instead: the batch form returns after phase k (instruction counts per phase, by subtraction).
#pragma once

#ifndef JSP_BATCH_LS
#define JSP_BATCH_LS 32
#endif
#ifndef JSP_FUSED_ALIGN
#define JSP_FUSED_ALIGN 1
#endif
#ifndef JSP_FUSED_VMCNT
#define JSP_FUSED_VMCNT 4
#endif
#ifndef JSP_FUSED_WAVES
#define JSP_FUSED_WAVES 4
#endif
#ifndef JSP_FUSED_LDS_PAD
#define JSP_FUSED_LDS_PAD 0
#endif
#ifndef JSP_FUSED_WAVES_TABLES
#define JSP_FUSED_WAVES_TABLES 4
#endif
#if defined(JSP_FUSED_STOP)
#define JSP_CLOCK_BEGIN() do { } while (0)
#define JSP_CLOCK(k) do { if ((MODE == 0 || MODE == 4) && (k) == JSP_FUSED_STOP) return; } while (0)
#else
#define JSP_FUSED_CLOCKS 1
#define JSP_CLOCK_BEGIN() unsigned long long clk_ = __builtin_readcyclecounter()
#define JSP_CLOCK(k) do { if ((MODE == 0 || MODE == 4) && threadIdx.x == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); \
    agg[(size_t)want * 9u + (size_t)(tile0 + blockIdx.x) * 8u + (k)] += now_ - clk_; clk_ = now_; } } while (0)
#endif
#if defined(JSP_FUSED_STOP)
constexpr bool kFusedClocks = false;
#else
constexpr bool kFusedClocks = true;
#endif
