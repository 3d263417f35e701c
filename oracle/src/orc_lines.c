/*
 * ORACLE (test infrastructure) -- stages a-8 .. a-14: grid topology, per-line quadratic fits,
 * pairwise intersections, centre selection and (col,row) indexing
 *   group_points_by_label / sort_rows            util_cylinder.py:376-394
 *   create_dummy_rows_cols                       :401-430
 *   fit_and_draw_polynomial (degree 2)           :473-550   (np.polyfit: scaled Vandermonde LSQ)
 *   remove_label                                 :1211-1269
 *   find_and_assign_intersections_P              :1106-1151 (poly_intersection_solver :1074-1104)
 *   clean_and_relabel                            :1154-1206
 *   indexing_data                                :1350-1571
 *   remove_minus_labels / make_json (ordering)   :1657-1727
 * PINNED by tests/golden/{topology,intersections,make_json}.json, produced by the real functions.
 *
 * [ext] scipy.optimize.root(method='hybr') is restated as a damped-free analytic Newton iteration
 * on the same 2x2 system from the same start (agrees with MINPACK to ~3e-9 px);
 * np.polyfit's SVD least squares is restated as Householder QR on the column-scaled Vandermonde.
 */
#include "orc_common.h"

#include "orc_lines.h"

int orc_capacity_overflow = 0;

ORC_API size_t orc_lineset_size(void) { return sizeof(orc_lineset); }

/* group_points_by_label + sort_rows: groups in order of first appearance, then STABLE sort by min y
 * (for rows AND cols, util_cylinder.py:388) */
ORC_API void orc_group_points(const int *cent, int n, const int32_t *labels, int lh, int lw, int x_off, int y_off,
                              orc_lineset *out)
{
    memset(out, 0, sizeof(*out));
    for (int i = 0; i < n; i++) {
        int rx = cent[2 * i] - x_off, ry = cent[2 * i + 1] - y_off;
        if (!(0 <= ry && ry < lh && 0 <= rx && rx < lw)) continue;
        int lab = labels[(size_t)ry * lw + rx];
        if (lab <= 0) continue;
        int g = -1;
        for (int k = 0; k < out->nlines; k++)
            if (out->label[k] == lab) { g = k; break; }
        if (g < 0) {
            if (out->nlines == ORC_MAXL) { orc_capacity_overflow = 1; continue; }
            g = out->nlines++;
            out->label[g] = lab;
            out->npts[g] = 0;
        }
        if (out->npts[g] < ORC_MAXLP) {
            out->pts[g][out->npts[g]][0] = cent[2 * i];
            out->pts[g][out->npts[g]][1] = cent[2 * i + 1];
            out->npts[g]++;
        } else orc_capacity_overflow = 1;
    }
    /* stable insertion sort by min y */
    int nl = out->nlines;
    double key[ORC_MAXL];
    int ord[ORC_MAXL];
    for (int g = 0; g < nl; g++) {
        double m = out->pts[g][0][1];
        for (int k = 1; k < out->npts[g]; k++)
            if (out->pts[g][k][1] < m) m = out->pts[g][k][1];
        key[g] = m; ord[g] = g;
    }
    for (int a = 1; a < nl; a++) {
        int o = ord[a]; int b = a - 1;
        while (b >= 0 && key[ord[b]] > key[o]) { ord[b + 1] = ord[b]; b--; }
        ord[b + 1] = o;
    }
    orc_lineset *tmp = (orc_lineset *)malloc(sizeof(*tmp));
    memcpy(tmp, out, sizeof(*tmp));
    for (int g = 0; g < nl; g++) {
        int s = ord[g];
        out->npts[g] = tmp->npts[s];
        out->label[g] = tmp->label[s];
        memcpy(out->pts[g], tmp->pts[s], sizeof(out->pts[g]));
        memset(out->eq[g], 0, sizeof(out->eq[g]));   /* create_dummy_rows_cols: [0]*(degree+4) */
        out->has_eq[g] = 1;
    }
    free(tmp);
}

/* np.polyfit(x, y, 2): lhs = vander(x,3) scaled per column by its 2-norm, least squares, unscale.
 * Householder QR on the n x 3 scaled matrix. */
static void polyfit2(const double *x, const double *y, int n, double *coef)
{
    double A[ORC_MAXLP][3], b[ORC_MAXLP], scale[3];
    for (int i = 0; i < n; i++) { A[i][0] = x[i] * x[i]; A[i][1] = x[i]; A[i][2] = 1.0; b[i] = y[i]; }
    for (int c = 0; c < 3; c++) {
        double s = 0;
        for (int i = 0; i < n; i++) s += A[i][c] * A[i][c];
        scale[c] = sqrt(s);
        for (int i = 0; i < n; i++) A[i][c] /= scale[c];
    }
    for (int c = 0; c < 3; c++) {
        double nrm = 0;
        for (int i = c; i < n; i++) nrm += A[i][c] * A[i][c];
        nrm = sqrt(nrm);
        double alpha = A[c][c] > 0 ? -nrm : nrm;
        double v[ORC_MAXLP];
        for (int i = c; i < n; i++) v[i] = A[i][c];
        v[c] -= alpha;
        double vn = 0;
        for (int i = c; i < n; i++) vn += v[i] * v[i];
        if (vn == 0) continue;
        for (int k = c; k < 3; k++) {
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
    double z[3];
    for (int r = 2; r >= 0; r--) {
        double s = b[r];
        for (int k = r + 1; k < 3; k++) s -= A[r][k] * z[k];
        z[r] = s / A[r][r];
    }
    for (int c = 0; c < 3; c++) coef[c] = z[c] / scale[c];
}

/* fit_and_draw_polynomial(degree=2) for one side; is_row: y = f(x) (sort by x) else x = f(y) (sort by y) */
ORC_API void orc_fit_lines(orc_lineset *ls, int is_row)
{
    for (int g = 0; g < ls->nlines; g++) {
        int n = ls->npts[g];
        if (n < 3) continue;
        double t[ORC_MAXLP], u[ORC_MAXLP];
        int ord[ORC_MAXLP];
        for (int i = 0; i < n; i++) ord[i] = i;
        int kc = is_row ? 0 : 1;
        for (int a = 1; a < n; a++) { /* stable sort by the independent coordinate */
            int o = ord[a]; int b = a - 1;
            while (b >= 0 && ls->pts[g][ord[b]][kc] > ls->pts[g][o][kc]) { ord[b + 1] = ord[b]; b--; }
            ord[b + 1] = o;
        }
        for (int i = 0; i < n; i++) { t[i] = ls->pts[g][ord[i]][kc]; u[i] = ls->pts[g][ord[i]][1 - kc]; }
        double c[3];
        polyfit2(t, u, n, c);
        double lo = t[0] - 50, hi = t[n - 1] + 50;
        ls->eq[g][0] = c[0]; ls->eq[g][1] = c[1]; ls->eq[g][2] = c[2];
        ls->eq[g][3] = lo; ls->eq[g][4] = hi; ls->eq[g][5] = fabs(hi - lo);
    }
}

static void drop_line(orc_lineset *ls, int g)
{
    for (int k = g; k + 1 < ls->nlines; k++) {
        ls->npts[k] = ls->npts[k + 1];
        ls->label[k] = ls->label[k + 1];
        ls->has_eq[k] = ls->has_eq[k + 1];
        memcpy(ls->pts[k], ls->pts[k + 1], sizeof(ls->pts[k]));
        memcpy(ls->eq[k], ls->eq[k + 1], sizeof(ls->eq[k]));
    }
    ls->nlines--;
}

/* remove_label: first row and last col (in the min-y order) are deleted */
ORC_API void orc_remove_label(orc_lineset *rows, orc_lineset *cols)
{
    if (rows->nlines > 0) drop_line(rows, 0);
    if (cols->nlines > 0) drop_line(cols, cols->nlines - 1);
}

static double polyval2(const double *c, double x) { return (c[0] * x + c[1]) * x + c[2]; }

/* poly_intersection_solver(row_eq, col_eq, 2): returns 1 and (x,y) when the reference returns a tuple */
ORC_API int orc_poly_intersection(const double *row_eq, const double *col_eq, double *xs, double *ys)
{
    const double *a = row_eq, *b = col_eq;
    double x_min = a[3], x_max = a[4], y_min = b[3], y_max = b[4];
    double x = 0.5 * (x_min + x_max);
    double y = polyval2(a, x);
    int ok = 0;
    for (int it = 0; it < 50; it++) {
        double f1 = y - polyval2(a, x), f2 = x - polyval2(b, y);
        /* J = [[-a'(x), 1], [1, -b'(y)]] */
        double da = 2 * a[0] * x + a[1], db = 2 * b[0] * y + b[1];
        double det = da * db - 1.0;
        if (det == 0 || !isfinite(det)) break;
        /* solve J d = -f */
        double dx = (-f1 * (-db) - 1.0 * (-f2)) / det;
        double dy = ((-da) * (-f2) - 1.0 * (-f1)) / det;
        x += dx; y += dy;
        if (!isfinite(x) || !isfinite(y)) break;
        double nd = sqrt(dx * dx + dy * dy), nx = sqrt(x * x + y * y);
        if (nd <= 1.49012e-8 * nx || nd == 0) { ok = 1; break; }
    }
    if (!ok) return 0;
    /* one polishing step keeps |f| at round-off like MINPACK's last iterate */
    {
        double f1 = y - polyval2(a, x), f2 = x - polyval2(b, y);
        double da = 2 * a[0] * x + a[1], db = 2 * b[0] * y + b[1];
        double det = da * db - 1.0;
        if (det != 0 && isfinite(det)) {
            x += (-f1 * (-db) - 1.0 * (-f2)) / det;
            y += ((-da) * (-f2) - 1.0 * (-f1)) / det;
        }
    }
    if ((x_min - 1e-3 <= x && x <= x_max + 1e-3) && (y_min - 1e-3 <= y && y <= y_max + 1e-3)) {
        *xs = x; *ys = y;
        return 1;
    }
    return 0;
}

/* find_and_assign_intersections_P: rows x cols in order; rect = boundingRect(max_contour) (closed test) */
ORC_API void orc_intersections(orc_lineset *rows, orc_lineset *cols, const int *rect)
{
    for (int r = 0; r < rows->nlines; r++) rows->npts[r] = 0;
    for (int c = 0; c < cols->nlines; c++) cols->npts[c] = 0;
    for (int r = 0; r < rows->nlines; r++)
        for (int c = 0; c < cols->nlines; c++) {
            double x, y;
            if (!orc_poly_intersection(rows->eq[r], cols->eq[c], &x, &y)) continue;
            if (!((rect[0] <= x && x <= rect[0] + rect[2]) && (rect[1] <= y && y <= rect[1] + rect[3]))) continue;
            if (rows->npts[r] < ORC_MAXLP) {
                rows->pts[r][rows->npts[r]][0] = x; rows->pts[r][rows->npts[r]][1] = y; rows->npts[r]++;
            }
            if (cols->npts[c] < ORC_MAXLP) {
                cols->pts[c][cols->npts[c]][0] = x; cols->pts[c][cols->npts[c]][1] = y; cols->npts[c]++;
            }
        }
}

/* clean_and_relabel one side: drop empty lines, stable sort by mean of coordinate `axis` */
static void clean_side(orc_lineset *ls, int axis)
{
    for (int g = 0; g < ls->nlines;) {
        if (ls->npts[g] == 0) drop_line(ls, g);
        else g++;
    }
    int nl = ls->nlines;
    double key[ORC_MAXL];
    int ord[ORC_MAXL];
    for (int g = 0; g < nl; g++) {
        double s = 0;
        for (int k = 0; k < ls->npts[g]; k++) s += ls->pts[g][k][axis];
        key[g] = s / ls->npts[g];
        ord[g] = g;
    }
    for (int a = 1; a < nl; a++) {
        int o = ord[a]; int b = a - 1;
        while (b >= 0 && key[ord[b]] > key[o]) { ord[b + 1] = ord[b]; b--; }
        ord[b + 1] = o;
    }
    orc_lineset *tmp = (orc_lineset *)malloc(sizeof(*tmp));
    memcpy(tmp, ls, sizeof(*tmp));
    for (int g = 0; g < nl; g++) {
        int s = ord[g];
        ls->npts[g] = tmp->npts[s];
        ls->label[g] = tmp->label[s];
        ls->has_eq[g] = tmp->has_eq[s];
        memcpy(ls->pts[g], tmp->pts[s], sizeof(ls->pts[g]));
        memcpy(ls->eq[g], tmp->eq[s], sizeof(ls->eq[g]));
    }
    free(tmp);
}

ORC_API void orc_clean_and_relabel(orc_lineset *rows, orc_lineset *cols)
{
    clean_side(rows, 1);
    clean_side(cols, 0);
}

/* indexing_data + remove_minus_labels + make_json ordering.
 *   gauss7: cv2.GaussianBlur(img,(7,7),0) of the grey frame (h x w)
 * outputs: center[2]; table rows (x, y, col, row) for col >= 0 sorted by (col,row), stable.
 * returns number of table rows, or -3 (no valid rows/cols, :1430-1460) / -4 (empty, :1703-1704). */
ORC_API int orc_index_points(const orc_lineset *rows, const orc_lineset *cols, const uint8_t *gauss7, int h, int w,
                             int r0, double *center, double *xy, int *id, int cap)
{
    if (rows->nlines == 0) return -3;
    int half = (int)(r0 / 5.0);
    if (half < 3) half = 3;
    if (half > 10) half = half + 5;
    /* centre = FIRST maximum of the window mean over rows in order, points in order */
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
            if (cnt <= 0) m = NAN; /* np.mean of an empty slice */
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
    /* centre row / col = line owning the nearest point (first minimum, math.hypot) */
    int crow = -1, ccol = -1;
    double md = INFINITY;
    for (int r = 0; r < rows->nlines; r++)
        for (int k = 0; k < rows->npts[r]; k++) {
            double ddx = cx - rows->pts[r][k][0], ddy = cy - rows->pts[r][k][1]; double d = sqrt(ddx * ddx + ddy * ddy);
            if (d < md) { md = d; crow = r; }
        }
    md = INFINITY;
    for (int c = 0; c < cols->nlines; c++)
        for (int k = 0; k < cols->npts[c]; k++) {
            double ddx = cx - cols->pts[c][k][0], ddy = cy - cols->pts[c][k][1]; double d = sqrt(ddx * ddx + ddy * ddy);
            if (d < md) { md = d; ccol = c; }
        }
    if (ccol < 0) return -3;
    /* cols_dict: every col point gets id = (col - centre col, nearest row - centre row) */
    int n = 0;
    int total = 0;
    for (int c = 0; c < cols->nlines; c++) {
        int ci = c - ccol;
        for (int k = 0; k < cols->npts[c]; k++) {
            double px = cols->pts[c][k][0], py = cols->pts[c][k][1];
            int nr = -1;
            double mdd = INFINITY;
            for (int r = 0; r < rows->nlines; r++)
                for (int q = 0; q < rows->npts[r]; q++) {
                    double ddx = px - rows->pts[r][q][0], ddy = py - rows->pts[r][q][1]; double d = sqrt(ddx * ddx + ddy * ddy);
                    if (d < mdd) { mdd = d; nr = r; }
                }
            int ri = nr >= 0 ? nr - crow : 0;
            if (ci < 0) continue; /* remove_minus_labels */
            total++;
            if (n < cap) { xy[2 * n] = px; xy[2 * n + 1] = py; id[2 * n] = ci; id[2 * n + 1] = ri; n++; }
        }
    }
    if (total == 0) return -4;
    /* make_json: stable sort by (col,row) */
    for (int a = 1; a < n; a++) {
        double kx = xy[2 * a], ky = xy[2 * a + 1];
        int kc = id[2 * a], kr = id[2 * a + 1];
        int b = a - 1;
        while (b >= 0 && (id[2 * b] > kc || (id[2 * b] == kc && id[2 * b + 1] > kr))) {
            xy[2 * b + 2] = xy[2 * b]; xy[2 * b + 3] = xy[2 * b + 1];
            id[2 * b + 2] = id[2 * b]; id[2 * b + 3] = id[2 * b + 1];
            b--;
        }
        xy[2 * b + 2] = kx; xy[2 * b + 3] = ky; id[2 * b + 2] = kc; id[2 * b + 3] = kr;
    }
    return total > cap ? cap : n;
}

/* ------------------------------------------------------------------------------------------------------
 * Row f-4: grey-level centre-of-gravity refinement of the fitted lines
 *   modify_grayscale_Cline / process_row / process_col / compute_center_of_gravity_x,y
 *   (util_cylinder.py:706-971; its call in color_and_expand_lines is commented out, :2040).
 * cv2-free for 2-D input: PINNED by tests/golden/subpixel.npz produced by the real functions.
 * numpy details restated: np.arange(start, stop, step) values are start + i*((start+step)-start);
 * np.sum of <= 8 float32 / float64 values (sequential below 8, the 8-way tree at 8); refined points are
 * stored as float32; np.polyfit runs in float64 on those (streaming Givens QR on the column-scaled rows here).
 * Returns 0, or 7 when the reference raises (a sample lies more than window/2 + 1 px above / left of the image:
 * negative slice stop at :722,:769 -> shape mismatch at :741,:786).
 */
static float sum_f32_np(const float *a, int n)
{
    if (n < 8) {
        float r = 0.f;
        for (int i = 0; i < n; i++) r += a[i];
        return r;
    }
    float r[8];
    for (int k = 0; k < 8; k++) r[k] = a[k];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
        for (int k = 0; k < 8; k++) r[k] += a[i + k];
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}
static double sum_f64_np(const double *a, int n)
{
    if (n < 8) {
        double r = 0.;
        for (int i = 0; i < n; i++) r += a[i];
        return r;
    }
    double r[8];
    for (int k = 0; k < 8; k++) r[k] = a[k];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
        for (int k = 0; k < 8; k++) r[k] += a[i + k];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}

/* centre of gravity along one axis; along_y: column ix fixed, window over rows.  returns 1 if the reference raises */
static int cog_refine(const uint8_t *gray, int h, int w, int half, int along_y, double fixed, double moving, double *out)
{
    const int L = along_y ? h : w, Lf = along_y ? w : h;
    int ifx = (int)lrint(fixed); /* int(round(.)): half to even */
    int lo = (int)floor(moving) - half, hi = (int)ceil(moving) + half + 1;
    if (lo < 0) lo = 0;
    if (hi > L) hi = L;
    *out = moving;
    if (ifx < 0 || ifx >= Lf) return 0;
    if (hi < 0) { /* python slice a[lo:hi] with negative hi wraps; np.arange(lo, hi) is empty */
        int wrapped = L + hi;
        int len_roi = wrapped > lo ? wrapped - lo : 0;
        if (len_roi != 0) return 1; /* ValueError: operands could not be broadcast */
        return 0;                   /* both empty: s == 0 -> unrefined */
    }
    int n = hi > lo ? hi - lo : 0;
    if (n == 0) return 0;
    float G[16];
    double prod[16];
    if (n > 16) n = 16;
    for (int k = 0; k < n; k++) {
        int idx = lo + k;
        uint8_t v = along_y ? gray[(size_t)idx * w + ifx] : gray[(size_t)ifx * w + idx];
        G[k] = (float)((double)v * (1.0 / 255));
    }
    float s = sum_f32_np(G, n);
    if (s == 0) return 0;
    for (int k = 0; k < n; k++) prod[k] = (double)(lo + k) * (double)G[k];
    double cog = sum_f64_np(prod, n) / (double)s;
    double delta = cog - moving;
    if (fabs(delta) > 0.5) delta = delta > 0 ? 0.5 : -0.5;
    double nv = moving + delta;
    if (nv < 0) nv = 0;
    if (nv > L - 1) nv = L - 1;
    *out = nv;
    return 0;
}

/* np.polyfit(x, y, 2) in streaming form: column norms, then Givens rotations of the scaled rows into R | c */
static void polyfit2_stream(const float *xs, const float *ys, int n, double *coef)
{
    double s0 = 0, s1 = 0, s2 = 0;
    for (int i = 0; i < n; i++) {
        double x = (double)xs[i], x2 = x * x;
        s0 += x2 * x2; s1 += x * x; s2 += 1.0;
    }
    double sc[3] = {sqrt(s0), sqrt(s1), sqrt(s2)};
    double R[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, c[3] = {0, 0, 0};
    for (int i = 0; i < n; i++) {
        double x = (double)xs[i];
        double row[3] = {(x * x) / sc[0], x / sc[1], 1.0 / sc[2]}, rhs = (double)ys[i];
        for (int k = 0; k < 3; k++) {
            if (row[k] == 0) continue;
            double a = R[k][k], b = row[k];
            double r = sqrt(a * a + b * b);
            double cg = a / r, sg = b / r;
            for (int j = k; j < 3; j++) {
                double t = cg * R[k][j] + sg * row[j];
                row[j] = -sg * R[k][j] + cg * row[j];
                R[k][j] = t;
            }
            double t = cg * c[k] + sg * rhs;
            rhs = -sg * c[k] + cg * rhs;
            c[k] = t;
        }
    }
    double z[3];
    for (int r = 2; r >= 0; r--) {
        double s = c[r];
        for (int k = r + 1; k < 3; k++) s -= R[r][k] * z[k];
        z[r] = s / R[r][r];
    }
    for (int k = 0; k < 3; k++) coef[k] = z[k] / sc[k];
}

/* modify_grayscale_Cline(gray2d, rows, cols, draw_points=False, degree=2, sample_step, window_size) */
ORC_API int orc_subpixel_refine(const uint8_t *gray, int h, int w, orc_lineset *rows, orc_lineset *cols, int window,
                                double step)
{
    const int half = window / 2;
    for (int side = 0; side < 2; side++) {
        orc_lineset *ls = side == 0 ? rows : cols;
        for (int g = 0; g < ls->nlines; g++) {
            double *eq = ls->eq[g];
            double lo = eq[3], hi = eq[4];
            if (hi < lo) continue;
            double stop = hi + 0.0001;
            double cnt = ceil((stop - lo) / step);
            int K = cnt > 0 ? (int)cnt : 0;
            if (K == 0) continue;
            float *xs = (float *)malloc((size_t)K * sizeof(float)), *ys = (float *)malloc((size_t)K * sizeof(float));
            double delta = (lo + step) - lo;
            for (int i = 0; i < K; i++) {
                double t = i == 0 ? lo : (i == 1 ? lo + step : lo + i * delta);
                double u = ((0.0 * t + eq[0]) * t + eq[1]) * t + eq[2];
                double ref;
                int raised = cog_refine(gray, h, w, half, side == 0 ? 1 : 0, t, u, &ref);   /* fixed = sample abscissa, moving = polynomial value */
                if (raised) { free(xs); free(ys); return 7; }
                /* rows: (x = t, y = refined); cols: (x = refined, y = t); the polyfit abscissa is t in both */
                xs[i] = (float)t;
                ys[i] = (float)ref;
            }
            if (K >= 3) {
                double c[3];
                polyfit2_stream(xs, ys, K, c);
                float mn = xs[0], mx = xs[0];
                for (int i = 1; i < K; i++) { if (xs[i] < mn) mn = xs[i]; if (xs[i] > mx) mx = xs[i]; }
                eq[0] = c[0]; eq[1] = c[1]; eq[2] = c[2];
                eq[3] = (double)mn; eq[4] = (double)mx; eq[5] = fabs((double)mx - (double)mn);
            }
            free(xs); free(ys);
        }
    }
    return 0;
}
