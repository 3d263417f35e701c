/*
 * ORACLE (test infrastructure) -- stages a-2, a-4, a-5, a-6, a-7:
 *   extract_joints (util_cylinder.py:1805-1827), find_cylinder_centroids_and_center (:1902-1941, the
 *   joint filter only: its centre guess is overwritten at :2047), mask_roi_around_center (:1944-2007),
 *   expands_line_roi / expand_line_roi / process_contour_info / create_rotated_line_kernel /
 *   get_pca_endpoints (:35-237), label_and_color_masks (:24-33).
 * [ext] OpenCV pieces (GaussianBlur 19x19 fixed point, threshold, warpAffine INTER_NEAREST,
 * connectedComponents) and numpy.linalg.eig of a 2x2 (LAPACK dgeev -> dlanv2) are restated; the
 * latter is PINNED by tests/golden/pca_endpoints.json (real get_pca_endpoints), the rest unpinned.
 */
#include "orc_common.h"

typedef struct orc_contours orc_contours;
orc_contours *orc_find_contours(const uint8_t *src, int h, int w, int mode, int method);
void orc_contours_free(orc_contours *cs);
int orc_contours_count(const orc_contours *cs);
int orc_contour_size(const orc_contours *cs, int i);
const orc_pt *orc_contour_points(const orc_contours *cs, int i);
void orc_contour_moments(const orc_pt *p, int n, double *m00, double *m10, double *m01);
double orc_contour_area(const orc_pt *p, int n);
void orc_min_enclosing_circle(const orc_pt *p, int n, float *cx, float *cy, float *r);
void orc_ellipse_fill(uint8_t *img, int h, int w, int cx, int cy, int a, int b, uint8_t c);
void orc_open_rect(const uint8_t *src, int h, int w, int kw, int kh, uint8_t *dst);
void orc_close_rect(const uint8_t *src, int h, int w, int kw, int kh, uint8_t *dst);
void orc_erode_rect(const uint8_t *src, int h, int w, int kw, int kh, uint8_t *dst);

/* ------------------------------------------------------------------ a-2 extract_joints */
/* returns the number of centroids written to cent (x,y pairs, contour order) */
ORC_API int orc_extract_joints(const uint8_t *binary, int h, int w, uint8_t *hmask, uint8_t *vmask, int *cent, int cap)
{
    orc_open_rect(binary, h, w, 20, 1, hmask);
    orc_open_rect(binary, h, w, 1, 20, vmask);
    uint8_t *j = (uint8_t *)malloc((size_t)h * w);
    for (size_t i = 0; i < (size_t)h * w; i++) j[i] = (hmask[i] && vmask[i]) ? 255 : 0;
    orc_contours *cs = orc_find_contours(j, h, w, 0, 2);
    int n = 0, nc = orc_contours_count(cs);
    for (int i = 0; i < nc; i++) {
        double m00, m10, m01;
        orc_contour_moments(orc_contour_points(cs, i), orc_contour_size(cs, i), &m00, &m10, &m01);
        if (m00 != 0) {
            if (n < cap) { cent[2 * n] = (int)(m10 / m00); cent[2 * n + 1] = (int)(m01 / m00); }
            n++;
        }
    }
    orc_contours_free(cs);
    free(j);
    return n;
}

/* ------------------------------------------------------------------ fixed-point Gaussian blurs (u8) */
static void blur_sep_u8(const uint8_t *src, int h, int w, const int *k, int r, int shift_total, uint8_t *dst)
{
    /* rows then columns with integer taps whose sum is 2^(shift_total/2) per axis; REFLECT_101 */
    int *tmp = (int *)malloc((size_t)h * w * sizeof(int));
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int j = -r; j <= r; j++) s += k[j + r] * src[(size_t)y * w + orc_reflect101(x + j, w)];
            tmp[(size_t)y * w + x] = s;
        }
    int half = 1 << (shift_total - 1);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int j = -r; j <= r; j++) s += k[j + r] * tmp[(size_t)orc_reflect101(y + j, h) * w + x];
            dst[(size_t)y * w + x] = (uint8_t)((s + half) >> shift_total);
        }
    free(tmp);
}

/* cv2.GaussianBlur(img,(19,19),0): sigma 3.2, 8.8 fixed-point taps with error diffusion (sum 256) */
ORC_API void orc_blur19(const uint8_t *src, int h, int w, uint8_t *dst)
{
    static const int k[19] = {1, 1, 3, 5, 10, 15, 20, 27, 30, 32, 30, 27, 20, 15, 10, 5, 3, 1, 1};
    blur_sep_u8(src, h, w, k, 9, 16, dst);
}

/* cv2.GaussianBlur(img,(7,7),0): [2 7 14 18 14 7 2]/64 */
ORC_API void orc_blur7(const uint8_t *src, int h, int w, uint8_t *dst)
{
    static const int k[7] = {2, 7, 14, 18, 14, 7, 2};
    blur_sep_u8(src, h, w, k, 3, 12, dst);
}

/* ------------------------------------------------------------------ a-5 mask_roi_around_center */
/* returns 0 ok, 2 = no saturated spot (UnboundLocalError in the reference).  r0 = circle_radius0 */
void orc_circle_fill(uint8_t *img, int h, int w, int cx, int cy, int radius, uint8_t c);
int orc_mask_roi_around_center_ex(const uint8_t *hmask, const uint8_t *vmask, const uint8_t *mask_contour,
                                  const uint8_t *gray, int h, int w, uint8_t *roi_h, uint8_t *roi_v, int *r0,
                                  int *spot /* optional: cx, cy, a, b */, int planar);
ORC_API int orc_mask_roi_around_center(const uint8_t *hmask, const uint8_t *vmask, const uint8_t *mask_contour,
                                       const uint8_t *gray, int h, int w, uint8_t *roi_h, uint8_t *roi_v, int *r0,
                                       int *spot /* optional: cx, cy, a, b */)
{
    return orc_mask_roi_around_center_ex(hmask, vmask, mask_contour, gray, h, w, roi_h, roi_v, r0, spot, 0);
}

/* planar != 0: util_plane.py:2733-2792 -- the spot is erased with a filled cv2.circle of radius int(r) (no ellipse, no
 * radius offsets) and that radius is what the function returns */
ORC_API int orc_mask_roi_around_center_ex(const uint8_t *hmask, const uint8_t *vmask, const uint8_t *mask_contour,
                                          const uint8_t *gray, int h, int w, uint8_t *roi_h, uint8_t *roi_v, int *r0,
                                          int *spot /* optional: cx, cy, a, b */, int planar)
{
    size_t N = (size_t)h * w;
    uint8_t *bl = (uint8_t *)malloc(N), *bin = (uint8_t *)malloc(N), *cm = (uint8_t *)malloc(N);
    orc_blur19(gray, h, w, bl);
    for (size_t i = 0; i < N; i++) bin[i] = bl[i] > 240 ? 240 : 0;
    orc_contours *cs = orc_find_contours(bin, h, w, 0, 2);
    int nc = orc_contours_count(cs);
    memset(cm, 255, N);
    int st = 2;
    if (nc > 0) {
        int best = 0;
        double ba = -1;
        for (int i = 0; i < nc; i++) { /* max(contours, key=contourArea): first maximum */
            double a = orc_contour_area(orc_contour_points(cs, i), orc_contour_size(cs, i));
            if (a > ba) { ba = a; best = i; }
        }
        float cx, cy, rad;
        orc_min_enclosing_circle(orc_contour_points(cs, best), orc_contour_size(cs, best), &cx, &cy, &rad);
        int icx = (int)cx, icy = (int)cy;
        int cr0 = (int)rad;
        int cr = rad < 30 ? cr0 + 20 : cr0 + 5;
        int minor = cr + 20 > 1 ? cr + 20 : 1;
        int a = (int)lrint((cr + 40) / 2.0), b = (int)lrint(minor / 2.0); /* python round(): half to even */
        if (planar) { a = b = cr0; orc_circle_fill(cm, h, w, icx, icy, cr0, 0); }
        else orc_ellipse_fill(cm, h, w, icx, icy, a, b, 0);
        *r0 = cr0;
        if (spot) { spot[0] = icx; spot[1] = icy; spot[2] = a; spot[3] = b; }
        st = 0;
    }
    orc_contours_free(cs);
    if (st == 0) {
        uint8_t *t = (uint8_t *)malloc(N);
        for (size_t i = 0; i < N; i++) t[i] = (hmask[i] & cm[i]) & mask_contour[i];
        orc_open_rect(t, h, w, 3, 3, roi_h);
        for (size_t i = 0; i < N; i++) t[i] = (vmask[i] & cm[i]) & mask_contour[i];
        orc_open_rect(t, h, w, 3, 3, roi_v);
        free(t);
    }
    free(bl); free(bin); free(cm);
    return st;
}

/* ------------------------------------------------------------------ a-6 line expansion */
/* numpy.linalg.eig of the symmetric 2x2 [[a,b],[b,d]] as LAPACK dgeev produces it for n = 2
 * (dhseqr -> dlanv2): eigenvalues w[2], eigenvectors as columns V (row-major 2x2). */
static double sign1(double a, double b) { return b >= 0 ? fabs(a) : -fabs(a); } /* Fortran SIGN (b = -0 ignored) */
static void eig2_lapack(double a, double b, double c, double d, double *w, double *V)
{
    double cs, sn;
    const double eps = 2.220446049250313e-16 / 2; /* dlamch('P') = eps*base = 2.2e-16; LAPACK's EPS = dlamch('P') */
    (void)eps;
    if (c == 0) {
        cs = 1; sn = 0;
    } else if (b == 0) {
        cs = 0; sn = 1;
        double t = d; d = a; a = t; b = -c; c = 0;
    } else if ((a - d) == 0 && sign1(1, b) != sign1(1, c)) {
        cs = 1; sn = 0;
    } else {
        double temp = a - d, p = 0.5 * temp;
        double bcmax = fmax(fabs(b), fabs(c));
        double bcmis = fmin(fabs(b), fabs(c)) * sign1(1, b) * sign1(1, c);
        double scale = fmax(fabs(p), bcmax);
        double z = (p / scale) * p + (bcmax / scale) * bcmis;
        if (z >= 4.0 * 2.220446049250313e-16) {
            z = p + sign1(sqrt(scale) * sqrt(z), p);
            a = d + z;
            d = d - (bcmax / z) * bcmis;
            double tau = hypot(c, z);
            cs = z / tau;
            sn = c / tau;
            b = b - c;
            c = 0;
        } else {
            /* complex or nearly equal eigenvalues: make diagonal elements equal (symmetric input: sigma = 0) */
            double sigma = b + c;
            double tau = hypot(sigma, temp);
            cs = sqrt(0.5 * (1 + fabs(sigma) / tau));
            sn = -(p / (tau * cs)) * sign1(1, sigma);
            double aa = a * cs + b * sn, bb = -a * sn + b * cs, cc = c * cs + d * sn, dd = -c * sn + d * cs;
            a = aa * cs + cc * sn; b = bb * cs + dd * sn; c = -aa * sn + cc * cs; d = -bb * sn + dd * cs;
            temp = 0.5 * (a + d);
            a = temp; d = temp;
        }
    }
    w[0] = a; w[1] = d;
    /* Schur vectors Z = [cs -sn; sn cs]; T upper triangular [a b; 0 d] -> eigenvectors of A */
    double v1x = cs, v1y = sn;
    double x0 = (a != d) ? -b / (a - d) : 0.0, x1 = 1.0; /* eigenvector of T for d */
    double v2x = cs * x0 - sn * x1, v2y = sn * x0 + cs * x1;
    double n2 = sqrt(v2x * v2x + v2y * v2y);
    v2x /= n2; v2y /= n2;
    V[0] = v1x; V[1] = v2x; V[2] = v1y; V[3] = v2y;
}

/* get_pca_endpoints(pts) for float32 points (util_cylinder.py:35-55); returns 0 if (None, None) */
ORC_API int orc_pca_endpoints(const float *pts, int n, float *p1, float *p2)
{
    if (n < 2) return 0;
    /* np.mean(pts, axis=0) in float32 (pairwise summation as numpy does for a strided column reduce:
     * axis-0 reduce of an (n,2) C array adds rows in order -- plain sequential f32 accumulation) */
    float mx = 0, my = 0;
    for (int i = 0; i < n; i++) { mx += pts[2 * i]; my += pts[2 * i + 1]; }
    mx = mx / (float)n; my = my / (float)n;
    /* np.cov(centered.T): float64, re-centred, / (n-1) */
    double ax = 0, ay = 0;
    for (int i = 0; i < n; i++) { ax += (double)(pts[2 * i] - mx); ay += (double)(pts[2 * i + 1] - my); }
    ax /= n; ay /= n;
    double sxx = 0, sxy = 0, syy = 0;
    for (int i = 0; i < n; i++) {
        double dx = (double)(pts[2 * i] - mx) - ax, dy = (double)(pts[2 * i + 1] - my) - ay;
        sxx += dx * dx; sxy += dx * dy; syy += dy * dy;
    }
    double f = 1.0 / (n - 1);
    sxx *= f; sxy *= f; syy *= f;
    double w[2], V[4];
    eig2_lapack(sxx, sxy, sxy, syy, w, V);
    int k = w[1] > w[0] ? 1 : 0; /* argmax: first maximum */
    double ux = V[0 + k], uy = V[2 + k];
    int imin = 0, imax = 0;
    double pmin = 0, pmax = 0;
    for (int i = 0; i < n; i++) {
        /* np.dot(centered(float32), axis(float64)) -> float64 */
        double pr = (double)(pts[2 * i] - mx) * ux + (double)(pts[2 * i + 1] - my) * uy;
        if (i == 0 || pr < pmin) { pmin = pr; imin = i; }
        if (i == 0 || pr > pmax) { pmax = pr; imax = i; }
    }
    p1[0] = pts[2 * imin]; p1[1] = pts[2 * imin + 1];
    p2[0] = pts[2 * imax]; p2[1] = pts[2 * imax + 1];
    return 1;
}

static int cv_round(double v) { return (int)lrint(v); }

/* create_rotated_line_kernel(size, angle): horizontal centre line rotated with
 * cv2.getRotationMatrix2D + cv2.warpAffine(INTER_NEAREST) (util_cylinder.py:57-76) */
ORC_API void orc_rotated_line_kernel(int size, double angle, uint8_t *ker)
{
    int c = size / 2;
    double a = angle * 3.1415926535897932384626433832795 / 180.0;
    double alpha = cos(a), beta = sin(a);
    double M[6] = {alpha, beta, (1 - alpha) * c - beta * c, -beta, alpha, beta * c + (1 - alpha) * c};
    /* warpAffine inverts the matrix (no WARP_INVERSE_MAP) */
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1. / D : 0;
    double A11 = M[4] * D, A22 = M[0] * D;
    double iM[6];
    iM[0] = A11; iM[1] = M[1] * (-D); iM[3] = M[3] * (-D); iM[4] = A22;
    double b1 = -iM[0] * M[2] - iM[1] * M[5];
    double b2 = -iM[3] * M[2] - iM[4] * M[5];
    iM[2] = b1; iM[5] = b2;
    const int AB_BITS = 10, AB_SCALE = 1 << AB_BITS, round_delta = AB_SCALE / 2;
    memset(ker, 0, (size_t)size * size);
    for (int y = 0; y < size; y++) {
        int X0 = cv_round((iM[1] * y + iM[2]) * AB_SCALE) + round_delta;
        int Y0 = cv_round((iM[4] * y + iM[5]) * AB_SCALE) + round_delta;
        for (int x = 0; x < size; x++) {
            int adelta = cv_round(iM[0] * x * AB_SCALE), bdelta = cv_round(iM[3] * x * AB_SCALE);
            int X = (X0 + adelta) >> AB_BITS, Y = (Y0 + bdelta) >> AB_BITS;
            /* source = the un-rotated kernel: row c is all ones */
            if (X >= 0 && X < size && Y == c) ker[y * size + x] = 1;
        }
    }
}

static int flt_cmp(const void *a, const void *b)
{
    float x = *(const float *)a, y = *(const float *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

/* expands_line_roi(mask, 1, mask_contour, patch 15, kernel_size) (util_cylinder.py:214-237, :137-212) */
void orc_expand_line_roi_ex(const uint8_t *mask_roi, const uint8_t *mask_contour, int h, int w, int kernel_size, int minp,
                            int maxp, uint8_t *out, int *dbg);
ORC_API void orc_expand_line_roi(const uint8_t *mask_roi, const uint8_t *mask_contour, int h, int w, int kernel_size,
                                 uint8_t *out, int *dbg /* optional: [n_contours, n_valid] */)
{
    orc_expand_line_roi_ex(mask_roi, mask_contour, h, w, kernel_size, 5, 200, out, dbg);
}

/* minp / maxp: vertex-count window of the fragments (5..200 in util_cylinder.py:137, 8..700 in util_plane.py:140) */
ORC_API void orc_expand_line_roi_ex(const uint8_t *mask_roi, const uint8_t *mask_contour, int h, int w, int kernel_size,
                                    int minp, int maxp, uint8_t *out, int *dbg /* optional: [n_contours, n_valid] */)
{
    const int patch = 15, half = patch / 2;
    size_t N = (size_t)h * w;
    uint8_t *base = (uint8_t *)malloc(N), *exp = (uint8_t *)malloc(N);
    orc_close_rect(mask_roi, h, w, 3, 3, base);
    for (size_t i = 0; i < N; i++) base[i] = base[i] ? 255 : 0;
    memcpy(exp, base, N);
    orc_contours *cs = orc_find_contours(base, h, w, 0, 2);
    int nc = orc_contours_count(cs);
    float *ang = (float *)malloc((size_t)(nc + 1) * sizeof(float)), *len = (float *)malloc((size_t)(nc + 1) * sizeof(float));
    float *ep = (float *)malloc((size_t)(nc + 1) * 4 * sizeof(float));
    char *valid = (char *)calloc((size_t)nc + 1, 1);
    float *alist = (float *)malloc((size_t)(nc + 1) * sizeof(float));
    int nv = 0;
    float glen = 0;
    for (int i = 0; i < nc; i++) {
        int np = orc_contour_size(cs, i);
        if (np < minp || np > maxp) continue;
        const orc_pt *p = orc_contour_points(cs, i);
        float *fp = (float *)malloc((size_t)np * 2 * sizeof(float));
        for (int k = 0; k < np; k++) { fp[2 * k] = (float)p[k].x; fp[2 * k + 1] = (float)p[k].y; }
        float p1[2], p2[2];
        int ok = orc_pca_endpoints(fp, np, p1, p2);
        free(fp);
        if (!ok) continue;
        float dx = p2[0] - p1[0], dy = p2[1] - p1[1];
        float length = (float)hypot((double)dx, (double)dy);   /* np.hypot on float32 */
        if (length < 1e-8) continue;
        /* angle = -np.degrees(np.arctan2(dy, dx)) in float32 */
        float at = (float)atan2((double)dy, (double)dx);
        float deg = at * (float)(180.0 / 3.14159265358979323846);
        ang[i] = -deg; len[i] = length;
        ep[4 * i] = p1[0]; ep[4 * i + 1] = p1[1]; ep[4 * i + 2] = p2[0]; ep[4 * i + 3] = p2[1];
        valid[i] = 1;
        alist[nv++] = ang[i];
        if (length > glen) glen = length;
    }
    if (dbg) { dbg[0] = nc; dbg[1] = nv; }
    if (nv > 0) {
        /* np.median of float32 values -> float32 (mean of the two middle ones for even counts) */
        qsort(alist, nv, sizeof(float), flt_cmp);
        float gang = (nv & 1) ? alist[nv / 2] : (float)(((double)alist[nv / 2 - 1] + (double)alist[nv / 2]) / 2.0);
        uint8_t *ker = (uint8_t *)malloc((size_t)kernel_size * kernel_size);
        int *koff = (int *)malloc((size_t)kernel_size * kernel_size * 2 * sizeof(int));
        uint8_t *dil = (uint8_t *)calloc(N, 1);
        int a = kernel_size / 2;
        for (int i = 0; i < nc; i++) {
            if (!valid[i]) continue;
            if ((double)len[i] > 0.8 * (double)glen) continue;
            float ak = fabsf(ang[i] - gang) > 5.0f ? gang : ang[i];
            orc_rotated_line_kernel(kernel_size, (double)ak, ker);
            int nko = 0;   /* the kernel is a thin line: walk its non-zero taps only */
            for (int kk = 0; kk < kernel_size * kernel_size; kk++)
                if (ker[kk]) { koff[2 * nko] = kk / kernel_size; koff[2 * nko + 1] = kk % kernel_size; nko++; }
            for (int e = 0; e < 2; e++) {
                int cx = (int)lrint((double)ep[4 * i + 2 * e]), cy = (int)lrint((double)ep[4 * i + 2 * e + 1]);
                int x1 = cx - half > 0 ? cx - half : 0, x2 = cx + half + 1 < w ? cx + half + 1 : w;
                int y1 = cy - half > 0 ? cy - half : 0, y2 = cy + half + 1 < h ? cy + half + 1 : h;
                /* dilate(endpoint patch, kernel) restricted to its support, then erode 3x3, OR */
                int bx1 = x1 - a - 1, bx2 = x2 + a + 1, by1 = y1 - a - 1, by2 = y2 + a + 1;
                if (bx1 < 0) bx1 = 0;
                if (by1 < 0) by1 = 0;
                if (bx2 > w) bx2 = w;
                if (by2 > h) by2 = h;
                for (int y = by1; y < by2; y++) memset(dil + (size_t)y * w + bx1, 0, (size_t)(bx2 - bx1));
                for (int y = y1; y < y2; y++)
                    for (int x = x1; x < x2; x++) {
                        if (!base[(size_t)y * w + x]) continue;
                        for (int q = 0; q < nko; q++) {
                            int yy = y - (koff[2 * q] - a), xx = x - (koff[2 * q + 1] - a);
                            if (yy >= 0 && yy < h && xx >= 0 && xx < w) dil[(size_t)yy * w + xx] = 255;
                        }
                    }
                for (int y = by1; y < by2; y++)
                    for (int x = bx1; x < bx2; x++) {
                        int all = 1;
                        for (int dy2 = -1; dy2 <= 1 && all; dy2++)
                            for (int dx2 = -1; dx2 <= 1; dx2++) {
                                int yy = y + dy2, xx = x + dx2;
                                if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
                                if (!dil[(size_t)yy * w + xx]) { all = 0; break; }
                            }
                        if (all) exp[(size_t)y * w + x] = 255;
                    }
            }
        }
        free(ker); free(koff); free(dil);
    }
    for (size_t i = 0; i < N; i++) out[i] = (exp[i] | base[i]) & mask_contour[i];
    orc_contours_free(cs);
    free(base); free(exp); free(ang); free(len); free(ep); free(valid); free(alist);
}

/* ------------------------------------------------------------------ a-7 connectedComponents (8-conn) */
/* labels: 0 background, components numbered 1.. in raster order of their first pixel
 * (label VALUES never reach the output, only membership: group_points_by_label) */
ORC_API int orc_connected_components(const uint8_t *mask, int h, int w, int32_t *labels)
{
    size_t N = (size_t)h * w;
    int32_t *parent = (int32_t *)malloc(N * sizeof(int32_t));
    for (size_t i = 0; i < N; i++) parent[i] = mask[i] ? (int32_t)i : -1;
#define FIND(x_) ({ int32_t r_ = (x_); while (parent[r_] != r_) r_ = parent[r_]; r_; })
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            size_t i = (size_t)y * w + x;
            if (!mask[i]) continue;
            static const int ox[4] = {-1, -1, 0, 1}, oy[4] = {0, -1, -1, -1};
            for (int k = 0; k < 4; k++) {
                int xx = x + ox[k], yy = y + oy[k];
                if (xx < 0 || xx >= w || yy < 0) continue;
                size_t j = (size_t)yy * w + xx;
                if (!mask[j]) continue;
                int32_t ra = FIND((int32_t)i), rb = FIND((int32_t)j);
                if (ra < rb) parent[rb] = ra; else if (rb < ra) parent[ra] = rb;
            }
        }
    int nl = 0;
    for (size_t i = 0; i < N; i++) {
        if (!mask[i]) { labels[i] = 0; continue; }
        int32_t r = FIND((int32_t)i);
        if ((size_t)r == i) labels[i] = ++nl;  /* roots are raster-first pixels: already labelled below */
    }
    for (size_t i = 0; i < N; i++)
        if (mask[i]) { int32_t r = FIND((int32_t)i); labels[i] = labels[r]; }
#undef FIND
    free(parent);
    return nl + 1;
}
