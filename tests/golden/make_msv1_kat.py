"""Writes msv1_kat.json: hand-worked MSVideo1 known-answer vectors (no decoder is run here).

Every expectation below is spelled out literally from the CRAM bit layout as the reference decodes
it: codes are LE16 words (a = low byte, b = high byte), buffer row 0 is the
first row of a block, flag bit k = 4*y + x, RGB555 -> (R<<19)|(G<<11)|(B<<3).
"""
import json
import os

RED, GREEN, BLUE = 0x00F80000, 0x0000F800, 0x000000F8

kats = []

# 1. 16-bit solid: word 0xFC00 = 0x8000 | 0x7C00 (R=31) -> all 16 pixels red
kats.append(dict(name="16_solid_red", bits=16, w=4, h=4, frames=[dict(src=[0x00, 0xFC], key=True,
            expect=[RED] * 16, adopted=True)]))

# 2. 16-bit 2-colour: stored flags 0x000F (bits 0..3 set => first colour on row 0),
#    c0 = 0x001F (blue), c1 = 0x03E0 (green)
kats.append(dict(name="16_two_colour", bits=16, w=4, h=4, frames=[dict(
    src=[0x0F, 0x00, 0x1F, 0x00, 0xE0, 0x03], key=True,
    expect=[BLUE] * 4 + [GREEN] * 12, adopted=True)]))

# 3. 16-bit 8-colour: stored flags 0 => every pixel takes the SECOND colour of its quadrant pair
#    colours 0x8001,2,3,4,5,6,7,8 -> B = value -> rgb = value << 3
q_tl, q_tr, q_bl, q_br = 2 << 3, 4 << 3, 6 << 3, 8 << 3
rows01 = [q_tl, q_tl, q_tr, q_tr]
rows23 = [q_bl, q_bl, q_br, q_br]
kats.append(dict(name="16_eight_colour_flags0", bits=16, w=4, h=4, frames=[dict(
    src=[0x00, 0x00, 0x01, 0x80, 2, 0, 3, 0, 4, 0, 5, 0, 6, 0, 7, 0, 8, 0], key=True,
    expect=rows01 * 2 + rows23 * 2, adopted=True)]))

# 4. 16-bit 8-colour with stored flags 0x7FFF: bits 0..14 set => FIRST colour of the pair,
#    pixel (3,3) (bit 15, cannot be stored set) => second colour of the bottom-right pair
f_tl, f_tr, f_bl, f_br = 1 << 3, 3 << 3, 5 << 3, 7 << 3
kats.append(dict(name="16_eight_colour_flags7fff", bits=16, w=4, h=4, frames=[dict(
    src=[0xFF, 0x7F, 0x01, 0x80, 2, 0, 3, 0, 4, 0, 5, 0, 6, 0, 7, 0, 8, 0], key=True,
    expect=[f_tl, f_tl, f_tr, f_tr] * 2 + [f_bl, f_bl, f_br, f_br] + [f_bl, f_bl, f_br, q_br],
    adopted=True)]))

# 5. 8x4, two frames: [solid red, solid green] then [skip 1, solid blue].
#    Buffer rows are 8 ints: block 0 = columns 0..3, block 1 = columns 4..7.
row_a = [RED] * 4 + [GREEN] * 4
row_b = [RED] * 4 + [BLUE] * 4
kats.append(dict(name="16_skip_then_solid", bits=16, w=8, h=4, lines=0, frames=[
    dict(src=[0x00, 0xFC, 0xE0, 0x83], key=True, expect=row_a * 4, adopted=True),
    dict(src=[0x01, 0x84, 0x1F, 0x80], key=False, expect=row_b * 4, adopted=True, signif=True),
]))

# 6. same, but the second frame repaints identical pixels: adopted (a coded block was seen) yet
#    not significant (stage-2 compare finds no difference)
kats.append(dict(name="16_repaint_identical", bits=16, w=8, h=4, lines=0, frames=[
    dict(src=[0x00, 0xFC, 0xE0, 0x83], key=True, expect=row_a * 4, adopted=True),
    dict(src=[0x01, 0x84, 0xE0, 0x83], key=False, expect=row_a * 4, adopted=True, signif=False),
]))

# 7. all-skip short stream: early-out, dst untouched, nothing adopted
kats.append(dict(name="16_all_skip_early_out", bits=16, w=8, h=4, lines=0, frames=[
    dict(src=[0x00, 0xFC, 0xE0, 0x83], key=True, expect=row_a * 4, adopted=True),
    dict(src=[0x02, 0x84], key=False, expect=None, adopted=False, signif=False),
]))

# 8. 8-bit: palette entry i = RGBQUAD (B=i, G=2i, R=3i, 0) -> 0x00RRGGBB
def pal(i):
    return ((3 * i) & 0xFF) << 16 | ((2 * i) & 0xFF) << 8 | i
palette = []
for i in range(256):
    palette += [i, (2 * i) & 0xFF, (3 * i) & 0xFF, 0]
#    2-colour: flags 0x0001 (not inverted): bit set => first index (5), clear => second (9)
kats.append(dict(name="8_two_colour", bits=8, w=4, h=4, palette=palette, frames=[dict(
    src=[0x01, 0x00, 5, 9], key=True, expect=[pal(5)] + [pal(9)] * 15, adopted=True)]))
#    solid: b = 0x80, a = index 7
kats.append(dict(name="8_solid", bits=8, w=4, h=4, palette=palette, frames=[dict(
    src=[7, 0x80], key=True, expect=[pal(7)] * 16, adopted=True)]))
#    8-colour: b = 0x90 | ..., stored flags 0x9000: inverted -> 0x6FFF: bits 0..11 set, 12 clear,
#    13,14 set, 15 clear ; bit set (after inversion) => second index of the pair
i8 = [10, 11, 12, 13, 14, 15, 16, 17]
exp = []
inv = 0x9000 ^ 0xFFFF
for y in range(4):
    for x in range(4):
        q = ((y & 2) << 1) + (x & 2)
        exp.append(pal(i8[q + ((inv >> (4 * y + x)) & 1)]))
kats.append(dict(name="8_eight_colour", bits=8, w=4, h=4, palette=palette, frames=[dict(
    src=[0x00, 0x90] + i8, key=True, expect=exp, adopted=True)]))
#    end marker 0,0 before the second block: second block keeps the caller's buffer content
kats.append(dict(name="8_end_marker", bits=8, w=8, h=4, palette=palette, prefill=0x00123456, frames=[dict(
    src=[7, 0x80, 0, 0], key=True, expect=([pal(7)] * 4 + [0x00123456] * 4) * 4, adopted=True)]))

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "msv1_kat.json"), "w") as f:
    json.dump(kats, f, indent=1)
print(len(kats), "vectors")
