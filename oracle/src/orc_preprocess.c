/*
 * ORACLE (test infrastructure) -- stage a-1: load_and_preprocess_image
 *   reference: utils/util_cylinder.py:1769-1802 (+ detect_ridges :1734-1738,
 *              sauvola_threshold_fast :1740-1766)
 *
 * [ext] restated here:
 *   cv2.GaussianBlur((5,5),0) u8      -> fixed-point [1 4 6 4 1]/16 per axis, REFLECT_101
 *   skimage.feature.hessian_matrix    -> img_as_float (x * (1/255)), scipy.ndimage
 *       gaussian_filter(sigma=3, mode='constant', truncate=4): axis 0 first, then axis 1,
 *       symmetric correlate1d summation order (centre tap, then pairs from the
 *       outermost inwards), np.gradient twice (central, one-sided at the border)
 *   hessian_matrix_eigvals            -> (a+c)/2 - sqrt(4 b^2 + (a-c)^2)/2   (smaller one)
 *   cv2.boxFilter 15x15 normalised, BORDER_REPLICATE on f64: RowSum then ColumnSum in
 *       OpenCV's update forms; the column sums run down the whole column as OpenCV's do,
 *       the row sums restart every ORC_BOX_BX columns -- documented deviation, parity unpinned
 *
 * The Hessian half is PINNED bit-for-bit against the real skimage/scipy
 * (tests/golden/ridge_*.npz, tools/gen_golden.py).
 */
#include "orc_common.h"

/* scipy.ndimage _gaussian_kernel1d(3.0, 0, 12): w[0] = centre, w[j] = tap at +-j
 * (values printed from the real scipy; hex-exact) */
static const double ORC_GW[13] = {
    0x1.105a329f98197p-3, 0x1.01a25f86eb137p-3, 0x1.b42a57d56c0bep-4,
    0x1.4a614d1afd337p-4, 0x1.bfde9c12bec92p-5, 0x1.0fa58939b5290p-5,
    0x1.26defcaeb0201p-6, 0x1.1e6bccad344bap-7, 0x1.f1e9915139407p-9,
    0x1.8345966f69519p-10, 0x1.0d8a5ad43c165p-11, 0x1.4fbe39149e277p-13,
    0x1.763a210dfb305p-15
};

/* cv2.GaussianBlur(gray,(5,5),0) -- util_cylinder.py:1790 */
ORC_API void orc_blur5(const uint8_t *src, int h, int w, uint8_t *dst)
{
    static const int k[5] = {1, 4, 6, 4, 1};
    int *tmp = (int *)malloc((size_t)h * w * sizeof(int));
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int j = -2; j <= 2; j++)
                s += k[j + 2] * src[(size_t)y * w + orc_reflect101(x + j, w)];
            tmp[(size_t)y * w + x] = s;
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int j = -2; j <= 2; j++)
                s += k[j + 2] * tmp[(size_t)orc_reflect101(y + j, h) * w + x];
            dst[(size_t)y * w + x] = (uint8_t)((s + 128) >> 8);
        }
    free(tmp);
}

/* img_as_float + ndi.gaussian_filter(sigma=3, mode='constant') -- skimage hessian_matrix */
ORC_API void orc_gauss_sigma3(const uint8_t *img, int h, int w, double *G)
{
    const double inv255 = 1.0 / 255;
    double *f = (double *)malloc((size_t)h * w * sizeof(double));
    double *v = (double *)malloc((size_t)h * w * sizeof(double));
    for (size_t i = 0; i < (size_t)h * w; i++) f[i] = (double)img[i] * inv255;
    /* axis 0 (rows / y) first */
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            double t = f[(size_t)y * w + x] * ORC_GW[0];
            for (int j = 12; j >= 1; j--) {
                double a = (y - j >= 0) ? f[(size_t)(y - j) * w + x] : 0.0;
                double b = (y + j < h) ? f[(size_t)(y + j) * w + x] : 0.0;
                double s = a + b;
                double p = s * ORC_GW[j];
                t = t + p;
            }
            v[(size_t)y * w + x] = t;
        }
    /* then axis 1 (x) */
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            double t = v[(size_t)y * w + x] * ORC_GW[0];
            for (int j = 12; j >= 1; j--) {
                double a = (x - j >= 0) ? v[(size_t)y * w + x - j] : 0.0;
                double b = (x + j < w) ? v[(size_t)y * w + x + j] : 0.0;
                double s = a + b;
                double p = s * ORC_GW[j];
                t = t + p;
            }
            G[(size_t)y * w + x] = t;
        }
    free(f);
    free(v);
}

/* np.gradient along one axis at (y,x) of array A (stride-aware) */
static inline double grad_x(const double *A, int h, int w, int y, int x)
{
    (void)h;
    const double *r = A + (size_t)y * w;
    if (w == 1) return 0.0;
    if (x == 0) return r[1] - r[0];
    if (x == w - 1) return r[w - 1] - r[w - 2];
    return (r[x + 1] - r[x - 1]) / 2.0;
}
static inline double grad_y(const double *A, int h, int w, int y, int x)
{
    if (h == 1) return 0.0;
    if (y == 0) return A[(size_t)w + x] - A[x];
    if (y == h - 1) return A[(size_t)(h - 1) * w + x] - A[(size_t)(h - 2) * w + x];
    return (A[(size_t)(y + 1) * w + x] - A[(size_t)(y - 1) * w + x]) / 2.0;
}

/* hessian_matrix(order='rc') + hessian_matrix_eigvals: writes the SMALLER eigenvalue
 * (util_cylinder.py:1793 keeps the second output); optionally the larger one too. */
ORC_API void orc_hessian_eigs(const double *G, int h, int w, double *emin, double *emax)
{
    double *gx = (double *)malloc((size_t)h * w * sizeof(double));
    double *gy = (double *)malloc((size_t)h * w * sizeof(double));
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            gx[(size_t)y * w + x] = grad_x(G, h, w, y, x);
            gy[(size_t)y * w + x] = grad_y(G, h, w, y, x);
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            double m00 = grad_x(gx, h, w, y, x); /* d2/dx2 */
            double m01 = grad_y(gx, h, w, y, x); /* d/dy (dG/dx) */
            double m11 = grad_y(gy, h, w, y, x); /* d2/dy2 */
            double t1 = m01 * m01;
            double t2 = 4.0 * t1;
            double t3 = m00 - m11;
            double t4 = t3 * t3;
            double t5 = t2 + t4;
            double t7 = sqrt(t5) / 2.0;
            double t9 = (m00 + m11) / 2.0;
            if (emin) emin[(size_t)y * w + x] = t9 - t7;
            if (emax) emax[(size_t)y * w + x] = t9 + t7;
        }
    free(gx);
    free(gy);
}

/* Row-sum block of the box filter restatement below: cv2's RowSum<double,double> keeps ONE running sum along a whole
 * image row (s += S[i + ksz] - S[i]), a sequential rounding history that no parallel evaluation reproduces.  The
 * restatement restarts that sum every ORC_BOX_BX columns (counted from the image's left edge): the first output of a
 * block is the direct 15-term sum, the next ORC_BOX_BX - 1 are OpenCV's update.  A constant of the specification, not of
 * any kernel's thread blocking (deviation 1); tests/test_preprocess_cpu.py evaluates the literal
 * whole-row form beside it and counts the mask pixels that differ. */
#define ORC_BOX_BX 8

/* sauvola_threshold_fast(b, 15, 0.5, 128) and the compare / invert of :1798-1800.
 * mask = 255 where b <= T (ridges), 0 where b > T. */
ORC_API void orc_sauvola_mask(const double *b, int h, int w, uint8_t *mask)
{
    /* cv2.boxFilter(b, CV_64F, (15,15), normalize=True, borderType=BORDER_REPLICATE), [ext] OpenCV 4.5.5 box_filter:
     * RowSum (restarted every ORC_BOX_BX columns, see above), then ColumnSum exactly as OpenCV runs it: SUM starts at 0,
     * takes the first ksize - 1 = 14 rows (7 replicas of row 0, rows 0..6) one after the other, then per output row
     * s0 = SUM + Sp (row y + 7), D = s0 * scale, SUM = s0 - Sm (row y - 7), down the whole column.  parity unpinned. */
    const int r = 7, BX = ORC_BOX_BX;
    const double scale = 1.0 / 225.0; /* 1./(ksize.width*ksize.height) */
    double *rs = (double *)malloc((size_t)h * w * sizeof(double));
    double *rs2 = (double *)malloc((size_t)h * w * sizeof(double));
    for (int y = 0; y < h; y++) {
        const double *row = b + (size_t)y * w;
        double s = 0.0, s2 = 0.0;
        for (int x = 0; x < w; x++) {
            if (x % BX == 0) {
                s = 0.0; s2 = 0.0;
                for (int j = -r; j <= r; j++) {
                    double v = row[orc_clampi(x + j, 0, w - 1)];
                    s = s + v;
                    s2 = s2 + v * v;
                }
            } else {           /* RowSum: s += S[i + ksz] - S[i] */
                double vin = row[orc_clampi(x + r, 0, w - 1)], vout = row[orc_clampi(x - r - 1, 0, w - 1)];
                s = s + (vin - vout);
                s2 = s2 + (vin * vin - vout * vout);
            }
            rs[(size_t)y * w + x] = s;
            rs2[(size_t)y * w + x] = s2;
        }
    }
    for (int x = 0; x < w; x++) {
        double c = 0.0, c2 = 0.0;
        for (int j = -r; j < r; j++) {      /* ColumnSum, sumCount == 0: SUM += Sp over the first ksize - 1 rows */
            size_t o = (size_t)orc_clampi(j, 0, h - 1) * w + x;
            c = c + rs[o];
            c2 = c2 + rs2[o];
        }
        for (int y = 0; y < h; y++) {
            size_t oin = (size_t)orc_clampi(y + r, 0, h - 1) * w + x, oout = (size_t)orc_clampi(y - r, 0, h - 1) * w + x;
            double s0 = c + rs[oin], s1 = c2 + rs2[oin];     /* s0 = SUM + Sp */
            double mean = s0 * scale;
            double mean_sq = s1 * scale;
            c = s0 - rs[oout];                                /* SUM = s0 - Sm */
            c2 = s1 - rs2[oout];
            double var = mean_sq - mean * mean;
            if (var < 0) var = 0;
            double sd = sqrt(var);
            double T = mean * (1 + 0.5 * ((sd / 128) - 1));
            mask[(size_t)y * w + x] = (b[(size_t)y * w + x] > T) ? 0 : 255;
        }
    }
    free(rs);
    free(rs2);
}

/* the whole stage: gray u8 -> blurred u8, binary mask u8 */
ORC_API void orc_preprocess(const uint8_t *gray, int h, int w, uint8_t *blurred, uint8_t *mask,
                            double *b_out /* optional h*w */)
{
    double *G = (double *)malloc((size_t)h * w * sizeof(double));
    double *b = b_out ? b_out : (double *)malloc((size_t)h * w * sizeof(double));
    orc_blur5(gray, h, w, blurred);
    orc_gauss_sigma3(blurred, h, w, G);
    orc_hessian_eigs(G, h, w, b, NULL);
    orc_sauvola_mask(b, h, w, mask);
    free(G);
    if (!b_out) free(b);
}

/* cv2.cvtColor(img, COLOR_BGR2GRAY) on 8-bit input (load_and_preprocess_image, util_cylinder.py:1781-1789), [ext] OpenCV
 * 4.5.5 RGB2Gray<uchar>: 15-bit fixed point, coefficients 0.114 / 0.587 / 0.299 -> 3735 / 19235 / 9798 (sum 2^15), rounded */
ORC_API void orc_bgr2gray(const uint8_t *bgr, size_t npx, uint8_t *gray)
{
    for (size_t p = 0; p < npx; p++)
        gray[p] = (uint8_t)((bgr[3 * p] * 3735u + bgr[3 * p + 1] * 19235u + bgr[3 * p + 2] * 9798u + (1u << 14)) >> 15);
}
