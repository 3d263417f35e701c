/*
 * ORACLE (test infrastructure) -- row f-2: the planar-target variant
 *   python_grid_detection_plane.py:74-119 (detect_grid) over utils/util_plane.py.  25 of the 34 functions it shares with
 *   util_cylinder.py are textually identical (load_and_preprocess_image, extract_joints,
 *   find_cylinder_centroids_and_center, group_points_by_label, find_and_assign_intersections_P, make_json ...); this
 *   file restates the ones that differ:
 *     get_convex_hull                      util_plane.py:2590-2689   (region finder; [ext] cv2 pieces, parity unpinned)
 *     mask_roi_around_center               :2733-2792  -> orc_mask_roi_around_center_ex(planar = 1)
 *     expand_line_roi (8..700 vertices)    :140-215    -> orc_expand_line_roi_ex
 *     color_and_expand_lines               :2799-2845  (kernel 201, degree 1, no remove_label / remove_minus_labels)
 *     fit_and_draw_polynomial              :411-634    (degree 1 + merging of "short" columns)   cv2-free: PINNED by
 *     clean_and_relabel                    :1204-1253  (drop empty, keep order)                  tests/golden/plane_lines.json
 *     indexing_data                        :1255-1472  (half = int(r / 4.5), ids as (row, col))
 */
#include "orc_common.h"

#include "orc_lines.h"

typedef struct orc_contours orc_contours;
orc_contours *orc_find_contours(const uint8_t *src, int h, int w, int mode, int method);
void orc_contours_free(orc_contours *cs);
int orc_contours_count(const orc_contours *cs);
int orc_contour_size(const orc_contours *cs, int i);
const orc_pt *orc_contour_points(const orc_contours *cs, int i);
double orc_contour_area(const orc_pt *p, int n);
void orc_bounding_rect(const orc_pt *p, int n, int *r);
int orc_convex_hull(const orc_pt *pin, int n, orc_pt *hull);
void orc_fill_poly(uint8_t *img, int h, int w, const orc_pt *v, int n, uint8_t c);
void orc_dilate_se(const uint8_t *src, int h, int w, const uint8_t *se, int ks, uint8_t *dst);
void orc_preprocess(const uint8_t *gray, int h, int w, uint8_t *blurred, uint8_t *mask, double *b_out);
int orc_extract_joints(const uint8_t *binary, int h, int w, uint8_t *hmask, uint8_t *vmask, int *cent, int cap);
int orc_mask_roi_around_center_ex(const uint8_t *hmask, const uint8_t *vmask, const uint8_t *mask_contour, const uint8_t *gray,
                                  int h, int w, uint8_t *roi_h, uint8_t *roi_v, int *r0, int *spot, int planar);
void orc_expand_line_roi_ex(const uint8_t *mask_roi, const uint8_t *mask_contour, int h, int w, int kernel_size, int minp,
                            int maxp, uint8_t *out, int *dbg);
int orc_connected_components(const uint8_t *mask, int h, int w, int32_t *labels);
void orc_blur7(const uint8_t *src, int h, int w, uint8_t *dst);
void orc_group_points(const int *cent, int n, const int32_t *labels, int lh, int lw, int x_off, int y_off, orc_lineset *out);

/* cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (ks, ks)) */
ORC_API void orc_ellipse_se(int ks, uint8_t *se)
{
    int r = ks / 2, c = ks / 2;
    double inv_r2 = r ? 1. / ((double)r * r) : 0;
    for (int i = 0; i < ks; i++) {
        int j1 = 0, j2 = 0, dy = i - r;
        if (abs(dy) <= r) {
            int dx = (int)lrint(c * sqrt((r * r - dy * dy) * inv_r2));
            j1 = c - dx > 0 ? c - dx : 0;
            j2 = c + dx + 1 < ks ? c + dx + 1 : ks;
        }
        for (int j = 0; j < ks; j++) se[i * ks + j] = (j >= j1 && j < j2) ? 1 : 0;
    }
}

/* the filled hull of the largest external contour of `bin` (first maximum of contourArea, strictly positive) -> out;
 * returns 0 and the contour's bounding rect, or 1 when there is none */
static int largest_hull_mask(const uint8_t *bin, int h, int w, int need_positive, uint8_t *out, int *rect)
{
    orc_contours *cs = orc_find_contours(bin, h, w, 0, 2);
    int nc = orc_contours_count(cs), best = -1;
    double ba = need_positive ? 0 : -1;
    for (int i = 0; i < nc; i++) {
        double a = orc_contour_area(orc_contour_points(cs, i), orc_contour_size(cs, i));
        if (a > ba) { ba = a; best = i; }
    }
    memset(out, 0, (size_t)h * w);
    if (best < 0) { orc_contours_free(cs); return 1; }
    const orc_pt *p = orc_contour_points(cs, best);
    int np = orc_contour_size(cs, best);
    orc_pt *hull = (orc_pt *)malloc((size_t)(np + 1) * sizeof(orc_pt));
    int nh = orc_convex_hull(p, np, hull);
    orc_fill_poly(out, h, w, hull, nh, 255);   /* polylines + fillPoly / drawContours(FILLED): same pixels */
    orc_bounding_rect(p, np, rect);
    free(hull);
    orc_contours_free(cs);
    return 0;
}

/* get_convex_hull(img, threshold=127, expansion_pixels): mask_contour = filled hull of the dilated filled hull;
 * rect = boundingRect(expanded_hull).  Status 1 = the reference raises ValueError (no contour). */
ORC_API int orc_get_convex_hull(const uint8_t *gray, int h, int w, int thr, int expansion, uint8_t *mask_contour, int *rect)
{
    size_t N = (size_t)h * w;
    uint8_t *bin = (uint8_t *)malloc(N), *m1 = (uint8_t *)malloc(N), *dil = (uint8_t *)malloc(N);
    for (size_t i = 0; i < N; i++) bin[i] = gray[i] > thr ? 255 : 0;
    int r1[4];
    int st = largest_hull_mask(bin, h, w, 1, m1, r1);   /* `if area > max_area` from 0: a zero-area contour never wins */
    if (st == 0) {
        int ks = expansion * 2 + 1;
        uint8_t *se = (uint8_t *)malloc((size_t)ks * ks);
        orc_ellipse_se(ks, se);
        orc_dilate_se(m1, h, w, se, ks, dil);
        free(se);
        st = largest_hull_mask(dil, h, w, 0, mask_contour, rect);   /* max(contours, key=contourArea) */
    } else memset(mask_contour, 0, N);
    free(bin); free(m1); free(dil);
    return st;
}

/* np.polyfit(x, y, 1): Householder QR on the column-scaled n x 2 Vandermonde */
static void polyfit1(const double *x, const double *y, int n, double *coef)
{
    double A[ORC_MAXLP][2], b[ORC_MAXLP], scale[2];
    for (int i = 0; i < n; i++) { A[i][0] = x[i]; A[i][1] = 1.0; b[i] = y[i]; }
    for (int c = 0; c < 2; c++) {
        double s = 0;
        for (int i = 0; i < n; i++) s += A[i][c] * A[i][c];
        scale[c] = sqrt(s);
        for (int i = 0; i < n; i++) A[i][c] /= scale[c];
    }
    static double v[ORC_MAXLP];
    for (int c = 0; c < 2; c++) {
        double nrm = 0;
        for (int i = c; i < n; i++) nrm += A[i][c] * A[i][c];
        nrm = sqrt(nrm);
        double alpha = A[c][c] > 0 ? -nrm : nrm;
        for (int i = c; i < n; i++) v[i] = A[i][c];
        v[c] -= alpha;
        double vn = 0;
        for (int i = c; i < n; i++) vn += v[i] * v[i];
        if (vn == 0) continue;
        for (int k = c; k < 2; k++) {
            double d = 0;
            for (int i = c; i < n; i++) d += v[i] * A[i][k];
            d = 2 * d / vn;
            for (int i = c; i < n; i++) A[i][k] -= d * v[i];
        }
        double d = 0;
        for (int i = c; i < n; i++) d += v[i] * b[i];
        d = 2 * d / vn;
        for (int i = c; i < n; i++) b[i] -= d * v[i];
    }
    double z[2];
    z[1] = b[1] / A[1][1];
    z[0] = (b[0] - A[0][1] * z[1]) / A[0][0];
    coef[0] = z[0] / scale[0];
    coef[1] = z[1] / scale[1];
}

/* sort the n points by coordinate kc (stable), fit the other coordinate as a degree-1 polynomial of it */
static void fit_sorted(const double (*pts)[2], int n, int kc, double *coef, double *lo, double *hi)
{
    static double t[ORC_MAXLP], u[ORC_MAXLP];
    static int ord[ORC_MAXLP];
    for (int i = 0; i < n; i++) ord[i] = i;
    for (int a = 1; a < n; a++) {
        int o = ord[a], b = a - 1;
        while (b >= 0 && pts[ord[b]][kc] > pts[o][kc]) { ord[b + 1] = ord[b]; b--; }
        ord[b + 1] = o;
    }
    for (int i = 0; i < n; i++) { t[i] = pts[ord[i]][kc]; u[i] = pts[ord[i]][1 - kc]; }
    polyfit1(t, u, n, coef);
    *lo = t[0]; *hi = t[n - 1];
}

/* fit_and_draw_polynomial(degree=1) of util_plane.py: equations are [c1, c0, lo, hi, |lo - hi|] (eq[5] unused).
 * Columns first: fit, find the "short" ones (|domain| <= 0.9 x the longest), merge runs of consecutive short columns
 * while their domains add up to at most the longest one, refit the merged columns, renumber; then the rows. */
ORC_API void orc_fit_lines_plane(orc_lineset *rows, orc_lineset *cols)
{
    int nc = cols->nlines;
    /* 1. first fit (domain +-10); columns with < 2 points keep the dummy [0]*5 */
    for (int g = 0; g < nc; g++) {
        memset(cols->eq[g], 0, sizeof(cols->eq[g]));
        if (cols->npts[g] < 2) continue;
        double c[2], lo, hi;
        fit_sorted((const double (*)[2])cols->pts[g], cols->npts[g], 1, c, &lo, &hi);
        lo -= 10; hi += 10;
        cols->eq[g][0] = c[0]; cols->eq[g][1] = c[1]; cols->eq[g][2] = lo; cols->eq[g][3] = hi; cols->eq[g][4] = fabs(lo - hi);
    }
    /* 2./3. */
    double thr = 0;
    for (int g = 0; g < nc; g++) if (fabs(cols->eq[g][4]) > thr) thr = fabs(cols->eq[g][4]);
    /* 4. merge groups over the columns in order */
    int gid[ORC_MAXL];   /* group number of a short column, -1 for a normal one */
    int ngroups = 0;
    {
        int open = 0;
        double cum = 0;
        for (int g = 0; g < nc; g++) {
            double d = fabs(cols->eq[g][4]);
            if (d <= 0.9 * thr) {
                if (open && cum + d <= thr) { gid[g] = ngroups - 1; cum += d; }
                else if (!open && cum + d <= thr) { gid[g] = ngroups++; open = 1; cum += d; }   /* first of a new run (cum = 0) */
                else { gid[g] = ngroups++; open = 1; cum = d; }
            } else {
                gid[g] = -1;
                open = 0; cum = 0;
            }
        }
    }
    /* 5./5a. rebuild the column list: a merged column sits where its first member was */
    orc_lineset *out = (orc_lineset *)calloc(1, sizeof(orc_lineset));
    static double mp[ORC_MAXLP][2];
    for (int g = 0; g < nc; g++) {
        if (gid[g] < 0) {
            int o = out->nlines++;
            out->npts[o] = cols->npts[g];
            memcpy(out->pts[o], cols->pts[g], sizeof(out->pts[o]));
            memcpy(out->eq[o], cols->eq[g], sizeof(out->eq[o]));
            out->label[o] = cols->label[g]; out->has_eq[o] = 1;
            continue;
        }
        if (g > 0 && gid[g - 1] == gid[g]) continue;   /* not the first member of its group */
        int m = 0;
        for (int q = g; q < nc && gid[q] == gid[g]; q++)
            for (int k = 0; k < cols->npts[q]; k++) {
                if (m < ORC_MAXLP) { mp[m][0] = cols->pts[q][k][0]; mp[m][1] = cols->pts[q][k][1]; m++; }
                else orc_capacity_overflow = 1;
            }
        if (m < 2) continue;   /* the members were deleted, nothing takes their place */
        int o = out->nlines++;
        double c[2], lo, hi;
        fit_sorted((const double (*)[2])mp, m, 1, c, &lo, &hi);
        out->npts[o] = m < ORC_MAXLP ? m : ORC_MAXLP;
        for (int k = 0; k < out->npts[o]; k++) { out->pts[o][k][0] = mp[k][0]; out->pts[o][k][1] = mp[k][1]; }
        out->eq[o][0] = c[0]; out->eq[o][1] = c[1]; out->eq[o][2] = lo; out->eq[o][3] = hi; out->eq[o][4] = fabs(lo - hi);
        out->label[o] = cols->label[g]; out->has_eq[o] = 1;
    }
    /* 6. final domains: +-50 around the points' own range */
    for (int g = 0; g < out->nlines; g++) {
        if (out->npts[g] < 2) continue;
        double lo = out->pts[g][0][1], hi = lo;
        for (int k = 1; k < out->npts[g]; k++) { double y = out->pts[g][k][1]; if (y < lo) lo = y; if (y > hi) hi = y; }
        lo -= 50; hi += 50;
        out->eq[g][2] = lo; out->eq[g][3] = hi; out->eq[g][4] = fabs(lo - hi);
    }
    memcpy(cols, out, sizeof(*cols));
    free(out);
    /* rows: plain degree-1 fit, domain +-50 */
    for (int g = 0; g < rows->nlines; g++) {
        memset(rows->eq[g], 0, sizeof(rows->eq[g]));
        if (rows->npts[g] < 2) continue;
        double c[2], lo, hi;
        fit_sorted((const double (*)[2])rows->pts[g], rows->npts[g], 0, c, &lo, &hi);
        lo -= 50; hi += 50;
        rows->eq[g][0] = c[0]; rows->eq[g][1] = c[1]; rows->eq[g][2] = lo; rows->eq[g][3] = hi; rows->eq[g][4] = fabs(lo - hi);
    }
}

/* poly_intersection_solver(row_eq, col_eq, 1): the same Newton iteration as the degree-2 restatement (orc_lines.c) */
ORC_API int orc_line_intersection(const double *a, const double *b, double *xs, double *ys)
{
    double x_min = a[2], x_max = a[3], y_min = b[2], y_max = b[3];
    double x = 0.5 * (x_min + x_max);
    double y = a[0] * x + a[1];
    int ok = 0;
    for (int it = 0; it < 50; it++) {
        double f1 = y - (a[0] * x + a[1]), f2 = x - (b[0] * y + b[1]);
        double da = a[0], db = b[0];
        double det = da * db - 1.0;
        if (det == 0 || !isfinite(det)) break;
        double dx = (-f1 * (-db) - 1.0 * (-f2)) / det;
        double dy = ((-da) * (-f2) - 1.0 * (-f1)) / det;
        x += dx; y += dy;
        if (!isfinite(x) || !isfinite(y)) break;
        double nd = sqrt(dx * dx + dy * dy), nx = sqrt(x * x + y * y);
        if (nd <= 1.49012e-8 * nx || nd == 0) { ok = 1; break; }
    }
    if (!ok) return 0;
    {
        double f1 = y - (a[0] * x + a[1]), f2 = x - (b[0] * y + b[1]);
        double da = a[0], db = b[0];
        double det = da * db - 1.0;
        if (det != 0 && isfinite(det)) {
            x += (-f1 * (-db) - 1.0 * (-f2)) / det;
            y += ((-da) * (-f2) - 1.0 * (-f1)) / det;
        }
    }
    if ((x_min - 1e-3 <= x && x <= x_max + 1e-3) && (y_min - 1e-3 <= y && y <= y_max + 1e-3)) { *xs = x; *ys = y; return 1; }
    return 0;
}

ORC_API void orc_intersections_plane(orc_lineset *rows, orc_lineset *cols, const int *rect)
{
    for (int r = 0; r < rows->nlines; r++) rows->npts[r] = 0;
    for (int c = 0; c < cols->nlines; c++) cols->npts[c] = 0;
    for (int r = 0; r < rows->nlines; r++)
        for (int c = 0; c < cols->nlines; c++) {
            double x, y;
            if (!orc_line_intersection(rows->eq[r], cols->eq[c], &x, &y)) continue;
            if (!((rect[0] <= x && x <= rect[0] + rect[2]) && (rect[1] <= y && y <= rect[1] + rect[3]))) continue;
            if (rows->npts[r] < ORC_MAXLP) { rows->pts[r][rows->npts[r]][0] = x; rows->pts[r][rows->npts[r]][1] = y; rows->npts[r]++; }
            if (cols->npts[c] < ORC_MAXLP) { cols->pts[c][cols->npts[c]][0] = x; cols->pts[c][cols->npts[c]][1] = y; cols->npts[c]++; }
        }
}

/* clean_and_relabel of util_plane.py on the point lists: empty lines go, the order stays */
ORC_API void orc_clean_plane(orc_lineset *ls)
{
    int o = 0;
    for (int g = 0; g < ls->nlines; g++) {
        if (ls->npts[g] == 0) continue;
        if (o != g) {
            ls->npts[o] = ls->npts[g]; ls->label[o] = ls->label[g]; ls->has_eq[o] = ls->has_eq[g];
            memcpy(ls->pts[o], ls->pts[g], sizeof(ls->pts[o]));
            memcpy(ls->eq[o], ls->eq[g], sizeof(ls->eq[o]));
        }
        o++;
    }
    ls->nlines = o;
}

/* indexing_data + make_json of the planar script: centre = first maximum of the (7x7-blurred) window mean with
 * half = int(r / 4.5); ids are (row - centre row, col - centre col); every column is kept; the table is sorted by id.
 * returns the number of table rows, -3 (no rows / no columns) or -4 (empty). */
ORC_API int orc_index_points_plane(const orc_lineset *rows, const orc_lineset *cols, const uint8_t *gauss7, int h, int w, int radius,
                                   double *center, double *xy, int *id, int cap)
{
    if (rows->nlines == 0) return -3;
    int half = (int)(radius / 4.5);
    double best = 0, cx = 0, cy = 0;
    int have = 0;
    for (int r = 0; r < rows->nlines; r++)
        for (int k = 0; k < rows->npts[r]; k++) {
            double x = rows->pts[r][k][0], y = rows->pts[r][k][1];
            int xs = (int)(x - half), xe = (int)(x + half), ys = (int)(y - half), ye = (int)(y + half);
            if (xs < 0) xs = 0;
            if (xe > w) xe = w;
            if (ys < 0) ys = 0;
            if (ye > h) ye = h;
            double m;
            long cnt = (long)(xe > xs ? xe - xs : 0) * (ye > ys ? ye - ys : 0);
            if (cnt <= 0) m = NAN;
            else {
                unsigned long s = 0;
                for (int yy = ys; yy < ye; yy++)
                    for (int xx = xs; xx < xe; xx++) s += gauss7[(size_t)yy * w + xx];
                m = (double)s / (double)cnt;
            }
            if (!have || m > best) { best = m; cx = x; cy = y; have = 1; }
        }
    if (!have) return -3;
    center[0] = cx; center[1] = cy;
    int crow = -1, ccol = -1;
    double md = INFINITY;
    for (int r = 0; r < rows->nlines; r++)
        for (int k = 0; k < rows->npts[r]; k++) {
            double ddx = cx - rows->pts[r][k][0], ddy = cy - rows->pts[r][k][1], d = sqrt(ddx * ddx + ddy * ddy);
            if (d < md) { md = d; crow = r; }
        }
    md = INFINITY;
    for (int c = 0; c < cols->nlines; c++)
        for (int k = 0; k < cols->npts[c]; k++) {
            double ddx = cx - cols->pts[c][k][0], ddy = cy - cols->pts[c][k][1], d = sqrt(ddx * ddx + ddy * ddy);
            if (d < md) { md = d; ccol = c; }
        }
    if (ccol < 0) return -3;
    int n = 0, total = 0;
    for (int c = 0; c < cols->nlines; c++) {
        int ci = c - ccol;
        for (int k = 0; k < cols->npts[c]; k++) {
            double px = cols->pts[c][k][0], py = cols->pts[c][k][1];
            int nr = -1;
            double mdd = INFINITY;
            for (int r = 0; r < rows->nlines; r++)
                for (int q = 0; q < rows->npts[r]; q++) {
                    double ddx = px - rows->pts[r][q][0], ddy = py - rows->pts[r][q][1], d = sqrt(ddx * ddx + ddy * ddy);
                    if (d < mdd) { mdd = d; nr = r; }
                }
            int ri = nr >= 0 ? nr - crow : 0;
            total++;
            if (n < cap) { xy[2 * n] = px; xy[2 * n + 1] = py; id[2 * n] = ri; id[2 * n + 1] = ci; n++; }
        }
    }
    if (total == 0) return -4;
    for (int a = 1; a < n; a++) {   /* make_json: stable sort by the id tuple */
        double kx = xy[2 * a], ky = xy[2 * a + 1];
        int k0 = id[2 * a], k1 = id[2 * a + 1];
        int b = a - 1;
        while (b >= 0 && (id[2 * b] > k0 || (id[2 * b] == k0 && id[2 * b + 1] > k1))) {
            xy[2 * b + 2] = xy[2 * b]; xy[2 * b + 3] = xy[2 * b + 1];
            id[2 * b + 2] = id[2 * b]; id[2 * b + 3] = id[2 * b + 1];
            b--;
        }
        xy[2 * b + 2] = kx; xy[2 * b + 3] = ky; id[2 * b + 2] = k0; id[2 * b + 3] = k1;
    }
    return total > cap ? cap : n;
}

typedef struct {
    uint8_t *binary, *hmask, *vmask, *mask_contour, *roi_h, *roi_v, *exp_h, *exp_v;
    int *joints;
    int cap_joints;
    int n_joints;
    int n_cyl_joints;
    int rect[4];
    int r0;
    int spot[4];
    int n_rows, n_cols;
    int n_keypoints;
} orc_detect_debug;

/* detect_grid of python_grid_detection_plane.py.  status: 0 ok, 1 no region (get_convex_hull raises), 2 no spot,
 * 3 no rows / cols, 4 empty */
ORC_API int orc_detect_grid_plane(const uint8_t *gray, int h, int w, double *center, double *xy, int *id, int cap, int *n_out,
                                  orc_detect_debug *dbg)
{
    size_t N = (size_t)h * w;
    uint8_t *blurred = (uint8_t *)malloc(N), *binary = (uint8_t *)malloc(N);
    uint8_t *hmask = (uint8_t *)malloc(N), *vmask = (uint8_t *)malloc(N), *mc = (uint8_t *)malloc(N);
    uint8_t *roi_h = (uint8_t *)malloc(N), *roi_v = (uint8_t *)malloc(N);
    uint8_t *exp_h = (uint8_t *)malloc(N), *exp_v = (uint8_t *)malloc(N), *g7 = (uint8_t *)malloc(N);
    int capj = 1 << 16;
    int *cent = (int *)malloc((size_t)capj * 2 * sizeof(int)), *cyl = (int *)malloc((size_t)capj * 2 * sizeof(int));
    orc_lineset *rows = (orc_lineset *)calloc(1, sizeof(orc_lineset)), *cols = (orc_lineset *)calloc(1, sizeof(orc_lineset));
    int32_t *lab_h = NULL, *lab_v = NULL;
    uint8_t *crop = NULL;
    *n_out = 0;
    orc_capacity_overflow = 0;
    orc_preprocess(gray, h, w, blurred, binary, NULL);
    int nj = orc_extract_joints(binary, h, w, hmask, vmask, cent, capj);
    if (nj > capj) nj = capj;
    int rect[4] = {0, 0, 0, 0}, r0 = 0, spot[4] = {0, 0, 0, 0};
    int st = orc_get_convex_hull(gray, h, w, 127, 5, mc, rect);
    int ncyl = 0;
    if (st == 0) {
        for (int i = 0; i < nj; i++) {
            int cx = cent[2 * i], cy = cent[2 * i + 1];
            if (rect[0] <= cx && cx < rect[0] + rect[2] && rect[1] <= cy && cy < rect[1] + rect[3]) {
                cyl[2 * ncyl] = cx; cyl[2 * ncyl + 1] = cy; ncyl++;
            }
        }
        if (ncyl > CPE_MAXJ) { orc_capacity_overflow = 1; ncyl = CPE_MAXJ; }   /* include/cpe.h: capacity of the joint table */
        st = orc_mask_roi_around_center_ex(hmask, vmask, mc, gray, h, w, roi_h, roi_v, &r0, spot, 1);
    }
    if (st == 0) {
        orc_expand_line_roi_ex(roi_h, mc, h, w, 201, 8, 700, exp_h, NULL);
        orc_expand_line_roi_ex(roi_v, mc, h, w, 201, 8, 700, exp_v, NULL);
        int x0 = rect[0], y0 = rect[1], cw = rect[2], ch = rect[3];
        if (x0 + cw > w) cw = w - x0;
        if (y0 + ch > h) ch = h - y0;
        crop = (uint8_t *)malloc((size_t)cw * ch);
        lab_h = (int32_t *)malloc((size_t)cw * ch * sizeof(int32_t));
        lab_v = (int32_t *)malloc((size_t)cw * ch * sizeof(int32_t));
        for (int y = 0; y < ch; y++) memcpy(crop + (size_t)y * cw, exp_h + (size_t)(y0 + y) * w + x0, (size_t)cw);
        orc_connected_components(crop, ch, cw, lab_h);
        for (int y = 0; y < ch; y++) memcpy(crop + (size_t)y * cw, exp_v + (size_t)(y0 + y) * w + x0, (size_t)cw);
        orc_connected_components(crop, ch, cw, lab_v);
        orc_group_points(cyl, ncyl, lab_h, ch, cw, x0, y0, rows);
        orc_group_points(cyl, ncyl, lab_v, ch, cw, x0, y0, cols);
        orc_fit_lines_plane(rows, cols);
        orc_intersections_plane(rows, cols, rect);
        orc_clean_plane(rows);
        orc_clean_plane(cols);
        orc_blur7(gray, h, w, g7);
        int n = orc_index_points_plane(rows, cols, g7, h, w, r0, center, xy, id, cap);
        if (n < 0) st = -n;
        else if (n > CPE_MAXP) orc_capacity_overflow = 1;
        else *n_out = n;
    }
    if (dbg) {
        if (dbg->binary) memcpy(dbg->binary, binary, N);
        if (dbg->hmask) memcpy(dbg->hmask, hmask, N);
        if (dbg->vmask) memcpy(dbg->vmask, vmask, N);
        if (dbg->mask_contour) memcpy(dbg->mask_contour, mc, N);
        if (dbg->roi_h && st != 1 && st != 2) memcpy(dbg->roi_h, roi_h, N);
        if (dbg->roi_v && st != 1 && st != 2) memcpy(dbg->roi_v, roi_v, N);
        if (dbg->exp_h && st != 1 && st != 2) memcpy(dbg->exp_h, exp_h, N);
        if (dbg->exp_v && st != 1 && st != 2) memcpy(dbg->exp_v, exp_v, N);
        if (dbg->joints) memcpy(dbg->joints, cent, (size_t)(nj < dbg->cap_joints ? nj : dbg->cap_joints) * 2 * sizeof(int));
        dbg->n_joints = nj;
        dbg->n_cyl_joints = ncyl;
        memcpy(dbg->rect, rect, sizeof(rect));
        dbg->r0 = r0;
        memcpy(dbg->spot, spot, sizeof(spot));
        dbg->n_rows = (st == 0 || st >= 3) ? rows->nlines : 0;
        dbg->n_cols = (st == 0 || st >= 3) ? cols->nlines : 0;
        dbg->n_keypoints = 0;
    }
    free(blurred); free(binary); free(hmask); free(vmask); free(mc); free(roi_h); free(roi_v); free(exp_h); free(exp_v);
    free(g7); free(cent); free(cyl); free(rows); free(cols); free(lab_h); free(lab_v); free(crop);
    if (orc_capacity_overflow) { st = ORC_ST_OVERFLOW; *n_out = 0; }
    return st;
}
