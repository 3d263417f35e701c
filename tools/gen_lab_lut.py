#!/usr/bin/env python3
"""L channel of cv2.cvtColor(BGR2LAB) for 8-bit grey-replicated pixels (util_cylinder.py:1840-1841),
restated from OpenCV 4.5.5 color_lab.cpp (RGB2Lab_b: sRGB gamma table, cube-root table, fixed point).
[ext], parity unpinned.  Prints the 256-entry LUT embedded in oracle/src/orc_blob.c and csrc/region.hip."""
import numpy as np
f32 = np.float32
gamma_shift, lab_shift, lab_shift2 = 3, 12, 15
def apply_gamma(x):
    x = f32(x)
    if x <= f32(0.04045):
        return x / f32(12.92)
    return f32(np.power((x + f32(0.055)) / f32(1.055), f32(2.4), dtype=np.float32))
gam = [int(np.rint(f32(255 * (1 << gamma_shift)) * apply_gamma(f32(i) / f32(255)))) for i in range(256)]
lthresh = f32(216) / f32(24389); lscale = f32(841) / f32(108); lbias = f32(16) / f32(116)
scale = f32(1) / (f32(255) * f32(1 << gamma_shift))
cb = []
for i in range(256 * 3 // 2 * (1 << gamma_shift)):
    x = scale * f32(i)
    v = (x * lscale + lbias) if x < lthresh else f32(np.cbrt(x, dtype=np.float32))
    cb.append(int(np.rint(f32(1 << lab_shift2) * v)))
Lscale = (116 * 255 + 50) // 100
Lshift = -((16 * 255 * (1 << lab_shift2) + 50) // 100)
lut = []
for v in range(256):
    fY = cb[gam[v]]              # Y = descale(tab[v] * 4096, 12) = tab[v] (the Y coefficients sum to 4096)
    L = (Lscale * fY + Lshift + (1 << (lab_shift2 - 1))) >> lab_shift2
    lut.append(min(255, max(0, L)))
print(', '.join(map(str, lut)))
# true-colour input: L = LY[(gam[R] * 871 + gam[G] * 2929 + gam[B] * 296 + 2048) >> 12]   (Y row of sRGB -> XYZ (D65) in 12-bit
# fixed point: cvRound(4096 * {0.212671, 0.715160, 0.072169}), sum 4096, so R = G = B = v gives Y = gam[v] and L = lut[v])
import sys
if len(sys.argv) > 1 and sys.argv[1] == 'colour':
    LY = []
    for y in range(255 * (1 << gamma_shift) + 1):
        L = (Lscale * cb[y] + Lshift + (1 << (lab_shift2 - 1))) >> lab_shift2
        LY.append(min(255, max(0, L)))
    assert all(LY[gam[v]] == lut[v] for v in range(256))
    print('GAMMA[256] =', ', '.join(map(str, gam)))
    print(f'LY[{len(LY)}] =', ', '.join(map(str, LY)))
