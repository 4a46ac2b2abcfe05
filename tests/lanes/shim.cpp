// TEST-ONLY shim: the lane-private part of the on-GPU MSVideo1 parse (jsplayer_amd/csrc/msv1_lanes.h) compiled for the
// host — the few GPU instructions it is written in are emulated by the header under JSP_LANES_HOST — so that masks,
// tables and visited sets can be checked against a sequential walk without a GPU (tests/test_msv1_lanes_cpu.py).
#define JSP_LANES_HOST 1
#include "../../jsplayer_amd/csrc/msv1_lanes.h"

using namespace jsp::lanes;

template <int BITS, int LS>
static void one_lane(const uint32_t* wp, uint32_t nvalid, int zw_all, uint32_t* out) {
    uint32_t w[LS / 2 + 1];
    for (int i = 0; i < LS / 2 + 1; ++i) w[i] = wp[i];
    const Masks m = build_masks<BITS, LS>(w, nvalid);
    out[0] = m.M; out[1] = m.L; out[2] = m.Z; out[3] = m.K; out[4] = m.EM; out[5] = m.valid;
    uint32_t tab[9];
    lane_table<BITS, LS>(w, m, zw_all ? 0xFFFFFFFFu : m.Z, tab);
    for (int e = 0; e < 9; ++e) out[6 + e] = tab[e];
    for (int e = 0; e < 9; ++e) out[15 + e] = visited<BITS, LS>(m, (uint32_t)e);
}

extern "C" {
// out: [M, L, Z, K, EM, valid, tab[9], visited[9]]
void lanes_one(int bits, int ls, const uint32_t* w, uint32_t nvalid, int zw_all, uint32_t* out) {
    if (bits == 16 && ls == 32) one_lane<16, 32>(w, nvalid, zw_all, out);
    else if (bits == 16) one_lane<16, 16>(w, nvalid, zw_all, out);
    else if (ls == 32) one_lane<8, 32>(w, nvalid, zw_all, out);
    else one_lane<8, 16>(w, nvalid, zw_all, out);
}
uint32_t lanes_perm(uint32_t s0, uint32_t s1, uint32_t sel) { return perm_b32(s0, s1, sel); }
}
