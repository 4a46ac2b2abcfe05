#!/usr/bin/env python3
"""lab: sp_pframe_group_kernel taken apart (variants of sp_kernels.hip written next to the tree's, built into copies of the library by
sp_group_parts.sh; pixels of the variants are WRONG, they are timed with --no-verify):
  nolit   the loader wave fetches no literal pixels (records and frame records still travel)
  norec   the worker waves treat every block as unchanged (no record read in the frame loop; the loader still works)
  idle    both, and the loader fetches no block records either (zero records): the frame loop + the hand-over alone
usage: sp_group_parts.py <sp_kernels.hip> <variant> <out.hip>"""
import sys

src, variant, out = sys.argv[1:4]
s = open(src).read()


def once(old, new):
    global s
    assert s.count(old) == 1, (variant, old[:60], s.count(old))
    s = s.replace(old, new)


if variant in ("nolit", "idle"):
    once("                while (m) {\n                    const int l = __ffsll((long long)m) - 1;", "                while (false && m) {\n                    const int l = __ffsll((long long)m) - 1;")
if variant in ("norec", "idle"):
    once("                if (pb.flags != 0 && cx0 < pb.x2 && cx0 + 4 > pb.x1) {\n                    const int w = pb.x2 - pb.x1;\n                    const uint32_t* lit0 = ck.lits + ck.lit_at[f * G2_BLOCKS + kb] - pb.x1;",
         "                if (false && pb.flags != 0 && cx0 < pb.x2 && cx0 + 4 > pb.x1) {\n                    const int w = pb.x2 - pb.x1;\n                    const uint32_t* lit0 = ck.lits + ck.lit_at[f * G2_BLOCKS + kb] - pb.x1;")
if variant == "idle":
    once("                if (q * 64 < nf_try * G2_BLOCKS && f < nf_try && k < nb_here)\n                    __builtin_amdgcn_global_load_lds(",
         "                ck.pb[q * 64 + lane] = PBlock{};\n                if (false && q * 64 < nf_try * G2_BLOCKS && f < nf_try && k < nb_here)\n                    __builtin_amdgcn_global_load_lds(")
open(out, "w").write(s)
