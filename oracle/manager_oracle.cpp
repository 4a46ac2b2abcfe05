// ORACLE (test infrastructure, not product): CPU restatement of the two Manager passes that touch every pixel right
// after the codec — what jsp_display_convert / jsp_frames_differ are checked against.  Only tests/, smoke() and
// bench.py's cpu_baseline leg may load this library.
//
// Parity unpinned by the reference (it ships no vectors): these are line-by-line restatements.
#include <cstddef>
#include <cstdint>

extern "C" {

// Manager.fill_bitmap_data, /root/reference/src/Manager.hx:325-390.  One 32-bit word per pixel in, one out; JS `<<` and
// `|` work on int32, the typed-array store keeps the low 32 bits.  `mode`:
//   0  canvas data present, !convert_fromRGB15   :379  dst[i] = 0xFF000000 | ((c & 0xFF) << 16) | (c & 0xFF00) | ((c >> 16) & 0xFF)
//   1  canvas data present, convert_fromRGB15    :370  dst[i] = 0xFF000000 | (src[i] << 3)
//   2  setPixels path, !convert_fromRGB15        :351  conv_buffer[i] = 0xFF000000 | c
//   3  setPixels path, convert_fromRGB15         :340  conv_buffer[i] = src[i] << 11
// The reference never flips rows (its display matrix does, Main.hx:318); `flip_rows` restates that flip: output row r
// = input row height-1-r.
void orc_display_convert(const int32_t* src, int32_t* dst, int width, int height, int mode, int flip_rows) {
    for (int y = 0; y < height; ++y) {
        const int32_t* in = src + (size_t)(flip_rows ? height - 1 - y : y) * width;
        int32_t* out = dst + (size_t)y * width;
        for (int x = 0; x < width; ++x) {
            const int32_t c = in[x];                                   // JS number holding an int32
            uint32_t v;
            switch (mode) {
                case 0: v = 0xFF000000u | (((uint32_t)c & 0xFFu) << 16) | ((uint32_t)c & 0xFF00u) | ((uint32_t)(c >> 16) & 0xFFu); break;   // >> is arithmetic, the mask makes it moot
                case 1: v = 0xFF000000u | ((uint32_t)c << 3); break;
                case 2: v = 0xFF000000u | (uint32_t)c; break;
                default: v = (uint32_t)c << 11; break;
            }
            out[x] = (int32_t)v;
        }
    }
}

// The pixel loop of Manager.frames_differ_significantly, Manager.hx:413-419: any pnt1[i] != pnt2[i] for
// first_pixel <= i < npixels (first_pixel = INSIGNIFICANT_LINES * X there).
int orc_frames_differ(const int32_t* a, const int32_t* b, size_t first_pixel, size_t npixels) {
    for (size_t i = first_pixel; i < npixels; ++i)
        if (a[i] != b[i]) return 1;
    return 0;
}

}  // extern "C"
