// msv1_fused_kernel's build constants and its (empty) measurement hooks.  The product is built from THIS file.  The lab builds of
// tools/lab/ (phase clocks, per-phase instruction counts) put a directory with their own msv1_fused_hooks.h in front of this one on
// the include path; none of that is compiled into the library.
#pragma once

#define JSP_BATCH_LS 32          // slots (2 bytes) per lane of a staged batch's tiles: 32 = 16 KiB tiles (one-frame launches of small frames use 16)
#define JSP_FUSED_ALIGN 1        // staging windows start on multiples of 256 blocks: a wave's row store begins on a 512-byte boundary
#define JSP_FUSED_VMCNT 4        // row stores of earlier blocks a wave may have in flight when it issues a block's four
#define JSP_FUSED_WAVES 4        // __launch_bounds__: waves per SIMD the register allocation must leave room for
#define JSP_FUSED_WAVES_TABLES 4 // ... of the table-writing form (MODE 4).  5 (96 VGPRs, 15 spilled) is SLOWER: 1.02 - 1.04 against 1.004 - 1.008 ms per 511 inter frames, profiles/r05_msv1_tables_5waves_ab.txt
#define JSP_FUSED_LDS_PAD 0      // words added to the kernel's LDS arena (lab: occupancy experiments)
#define JSP_CLOCK_BEGIN() do { } while (0)
#define JSP_CLOCK(k) do { } while (0)
constexpr bool kFusedClocks = false;     // (msv1_codec.cpp: the lab build prints the clocks at every sync)
