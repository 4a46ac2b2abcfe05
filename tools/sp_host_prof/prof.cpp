#include "sp.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace jsp::sp;
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); int reps = argc > 2 ? atoi(argv[2]) : 3;
    uint32_t n; fread(&n, 4, 1, f);
    std::vector<std::vector<uint8_t>> fr(n); std::vector<uint8_t> key(n);
    for (uint32_t i = 0; i < n; ++i) { uint32_t l; fread(&l, 4, 1, f); fread(&key[i], 1, 1, f); fr[i].resize(l); fread(fr[i].data(), 1, l, f); }
    HostDecoder hd(1920, 1080, 24);
    hd.set_iframe_layout(8, 256);
    FrameOut out;
    double bi = 1e9, bp = 1e9;
    for (int r = 0; r < reps; ++r) {
        auto t0 = std::chrono::steady_clock::now();
        double ti = 0, tp = 0; int ni = 0, np = 0;
        for (uint32_t i = 0; i < n; ++i) {
            auto a = std::chrono::steady_clock::now();
            if (key[i]) hd.decode_i(fr[i].data(), fr[i].size(), out); else hd.decode_p(fr[i].data(), fr[i].size(), out);
            auto b = std::chrono::steady_clock::now();
            double d = std::chrono::duration<double, std::milli>(b - a).count();
            if (key[i]) { ti += d; ++ni; } else { tp += d; ++np; }
            if (out.status) { printf("status %d %s\n", out.status, out.error ? out.error : ""); return 1; }
        }
        if (ni && ti / ni < bi) bi = ti / ni;
        if (np && tp / np < bp) bp = tp / np;
        if (r == reps - 1) printf("best: I %.2f  P %.2f ms/frame\n", bi, bp);
        if (0) printf("rep %d: I %.2f ms/frame (%d)  P %.2f ms/frame (%d)\n", r, ni ? ti / ni : 0, ni, np ? tp / np : 0, np);
    }
}
