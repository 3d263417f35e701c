/*
 * ORACLE (test infrastructure) -- binary morphology as OpenCV 4.5.5 defines it ([ext], parity unpinned):
 *   dst(x,y) = min/max over kernel non-zeros (kx,ky) of src(x + kx - ax, y + ky - ay), anchor = (kw/2, kh/2),
 *   the SAME formula for erode and dilate (no reflection), border value +inf for erode / -inf for dilate
 *   (morphologyDefaultBorderValue) -- so an opening with an even-length bar shifts runs by +1 px.
 * Call sites: util_cylinder.py:1810-1814 (open 20x1, 1x20), :2003-2005 (open 3x3), :150 (close 3x3),
 *             :126-127 (dilate with the rotated line kernel, erode 3x3).
 * Masks are u8 with any non-zero = foreground; outputs are 0/255.
 */
#include "orc_common.h"

ORC_API void orc_erode_rect(const uint8_t *src, int h, int w, int kw, int kh, uint8_t *dst)
{
    int ax = kw / 2, ay = kh / 2;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int all = 1;
            for (int ky = 0; ky < kh && all; ky++) {
                int yy = y + ky - ay;
                if (yy < 0 || yy >= h) continue;
                for (int kx = 0; kx < kw; kx++) {
                    int xx = x + kx - ax;
                    if (xx < 0 || xx >= w) continue;
                    if (!src[(size_t)yy * w + xx]) { all = 0; break; }
                }
            }
            dst[(size_t)y * w + x] = all ? 255 : 0;
        }
}

ORC_API void orc_dilate_rect(const uint8_t *src, int h, int w, int kw, int kh, uint8_t *dst)
{
    int ax = kw / 2, ay = kh / 2;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int any = 0;
            for (int ky = 0; ky < kh && !any; ky++) {
                int yy = y + ky - ay;
                if (yy < 0 || yy >= h) continue;
                for (int kx = 0; kx < kw; kx++) {
                    int xx = x + kx - ax;
                    if (xx < 0 || xx >= w) continue;
                    if (src[(size_t)yy * w + xx]) { any = 1; break; }
                }
            }
            dst[(size_t)y * w + x] = any ? 255 : 0;
        }
}

ORC_API void orc_open_rect(const uint8_t *src, int h, int w, int kw, int kh, uint8_t *dst)
{
    uint8_t *t = (uint8_t *)malloc((size_t)h * w);
    orc_erode_rect(src, h, w, kw, kh, t);
    orc_dilate_rect(t, h, w, kw, kh, dst);
    free(t);
}

ORC_API void orc_close_rect(const uint8_t *src, int h, int w, int kw, int kh, uint8_t *dst)
{
    uint8_t *t = (uint8_t *)malloc((size_t)h * w);
    orc_dilate_rect(src, h, w, kw, kh, t);
    orc_erode_rect(t, h, w, kw, kh, dst);
    free(t);
}

/* dilate with an arbitrary ks x ks binary structuring element, anchor (ks/2, ks/2) */
ORC_API void orc_dilate_se(const uint8_t *src, int h, int w, const uint8_t *se, int ks, uint8_t *dst)
{
    int a = ks / 2;
    memset(dst, 0, (size_t)h * w);
    /* scatter form: every foreground source pixel p turns on p - (k - a) for every SE non-zero k */
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            if (!src[(size_t)y * w + x]) continue;
            for (int ky = 0; ky < ks; ky++)
                for (int kx = 0; kx < ks; kx++) {
                    if (!se[ky * ks + kx]) continue;
                    int yy = y - (ky - a), xx = x - (kx - a);
                    if (yy >= 0 && yy < h && xx >= 0 && xx < w) dst[(size_t)yy * w + xx] = 255;
                }
        }
}
