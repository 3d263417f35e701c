// Stages a-7 .. a-14 on the GPU, one workgroup per image:
//   label lookup of the joints (label_and_color_masks / group_points_by_label, util_cylinder.py:24-33, 376-394),
//   create_dummy_rows_cols (:401-430), fit_and_draw_polynomial(2) (:473-550), remove_label (:1211-1269),
//   find_and_assign_intersections_P / poly_intersection_solver (:1074-1151), clean_and_relabel (:1154-1206),
//   indexing_data (:1350-1571), remove_minus_labels + make_json ordering (:1657-1727).
// The arithmetic (operation order of the polynomial fit, the Newton iteration, the means) is the one
// written in oracle/src/orc_lines.c, which is pinned against the real reference functions.
// Lines live in per-image workspace slots; every re-ordering is an index list (no data movement).
#include "cpe_dev.h"

namespace cpe {

constexpr int ENT_CAP = 2 * CPE_MAXP;   // entries of the (col, row) table that are ranked; more than CPE_MAXP is an overflow anyway

struct LinesWS {
    // joints per label group: ONE pool per direction, the groups one after the other in order of first appearance
    // (goff / gn), so a group may hold any share of the frame's joints; second half: the merged column lists of the
    // planar script
    double pool[2][2 * MAXJ][2];
    double ipts[2][MAXL][MAXL][2];   // intersections per line: at most one per line of the other direction
    double eq[2][MAXL][6];
    double ixy[MAXL][MAXL][2];
    double key[2][MAXL];
    double ent_xy[ENT_CAP][2];
    int ent_id[ENT_CAP][2];
    int gn[2][MAXL], goff[2][MAXL], in[2][MAXL], glabel[2][MAXL];
    unsigned char ival[MAXL][MAXL];
    double fit_scr[2][7 * MAXJ];     // scratch of the fits, 7 doubles per point of a line: sorted abscissae / ordinates, QR columns
    int fit_ord[2][MAXJ];
    int jlab[2][MAXJ];               // per joint and mask: label (union-find root) or -1, then its group and its rank in it
    int jgrp[2][MAXJ], jrank[2][MAXJ];
    int fin_ord[2][MAXL], fin_n[2];  // lines that survive clean_and_relabel, in their final order (row1.., col1..)
};

size_t lines_ws_bytes() { return align_up(sizeof(LinesWS), 256); }

namespace {

__device__ __forceinline__ double polyval2(const double *c, double x) { return (c[0] * x + c[1]) * x + c[2]; }

// np.polyfit(x, y, 2) as restated in the oracle: column-scaled Vandermonde + Householder QR
// scr: 5 n doubles of scratch (the per-line slots of LinesWS: a line may hold MAXLP points, too many for the stack)
__device__ void polyfit2(const double *x, const double *y, int n, double *coef, double *scr)
{
    double (*A)[3] = reinterpret_cast<double (*)[3]>(scr);
    double *b = scr + 3 * n, *v = scr + 4 * n;
    double scale[3];
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

// np.polyfit(x, y, 1) (planar script, util_plane.py:411-634): column-scaled n x 2 Vandermonde + Householder QR
__device__ void polyfit1(const double *x, const double *y, int n, double *coef, double *scr)
{
    double (*A)[2] = reinterpret_cast<double (*)[2]>(scr);
    double *b = scr + 2 * n, *v = scr + 3 * n;
    double scale[2];
    for (int i = 0; i < n; i++) { A[i][0] = x[i]; A[i][1] = 1.0; b[i] = y[i]; }
    for (int c = 0; c < 2; c++) {
        double s = 0;
        for (int i = 0; i < n; i++) s += A[i][c] * A[i][c];
        scale[c] = sqrt(s);
        for (int i = 0; i < n; i++) A[i][c] /= scale[c];
    }
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
            double dd = 0;
            for (int i = c; i < n; i++) dd += v[i] * A[i][k];
            dd = 2 * dd / vn;
            for (int i = c; i < n; i++) A[i][k] -= dd * v[i];
        }
        double dd = 0;
        for (int i = c; i < n; i++) dd += v[i] * b[i];
        dd = 2 * dd / vn;
        for (int i = c; i < n; i++) b[i] -= dd * v[i];
    }
    double z1 = b[1] / A[1][1];
    double z0 = (b[0] - A[0][1] * z1) / A[0][0];
    coef[0] = z0 / scale[0];
    coef[1] = z1 / scale[1];
}

// sort n points of a line by coordinate kc (stable), fit the other coordinate: coef (c1, c0), range of the abscissa
__device__ void fit_line_sorted(const double (*pts)[2], int n, int kc, double *coef, double &lo, double &hi, double *scr, int *ord)
{
    double *tt = scr, *uu = scr + n;
    for (int i = 0; i < n; i++) ord[i] = i;
    for (int a = 1; a < n; a++) {
        int o = ord[a];
        int b = a - 1;
        while (b >= 0 && pts[ord[b]][kc] > pts[o][kc]) { ord[b + 1] = ord[b]; b--; }
        ord[b + 1] = o;
    }
    for (int i = 0; i < n; i++) { tt[i] = pts[ord[i]][kc]; uu[i] = pts[ord[i]][1 - kc]; }
    polyfit1(tt, uu, n, coef, scr + 2 * n);
    lo = tt[0]; hi = tt[n - 1];
}

// poly_intersection_solver(row_eq, col_eq, degree 1): equations [c1, c0, lo, hi, ...]
__device__ bool line_intersection(const double *a, const double *b, double &xs, double &ys)
{
    double x_min = a[2], x_max = a[3], y_min = b[2], y_max = b[3];
    double x = 0.5 * (x_min + x_max);
    double y = a[0] * x + a[1];
    bool ok = false;
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
        if (nd <= 1.49012e-8 * nx || nd == 0) { ok = true; break; }
    }
    if (!ok) return false;
    {
        double f1 = y - (a[0] * x + a[1]), f2 = x - (b[0] * y + b[1]);
        double da = a[0], db = b[0];
        double det = da * db - 1.0;
        if (det != 0 && isfinite(det)) {
            x += (-f1 * (-db) - 1.0 * (-f2)) / det;
            y += ((-da) * (-f2) - 1.0 * (-f1)) / det;
        }
    }
    if ((x_min - 1e-3 <= x && x <= x_max + 1e-3) && (y_min - 1e-3 <= y && y <= y_max + 1e-3)) {
        xs = x; ys = y;
        return true;
    }
    return false;
}

// poly_intersection_solver restated (analytic Newton from the reference's start point)
__device__ bool poly_intersection(const double *a, const double *b, double &xs, double &ys)
{
    double x_min = a[3], x_max = a[4], y_min = b[3], y_max = b[4];
    double x = 0.5 * (x_min + x_max);
    double y = polyval2(a, x);
    bool ok = false;
    for (int it = 0; it < 50; it++) {
        double f1 = y - polyval2(a, x), f2 = x - polyval2(b, y);
        double da = 2 * a[0] * x + a[1], db = 2 * b[0] * y + b[1];
        double det = da * db - 1.0;
        if (det == 0 || !isfinite(det)) break;
        double dx = (-f1 * (-db) - 1.0 * (-f2)) / det;
        double dy = ((-da) * (-f2) - 1.0 * (-f1)) / det;
        x += dx; y += dy;
        if (!isfinite(x) || !isfinite(y)) break;
        double nd = sqrt(dx * dx + dy * dy), nx = sqrt(x * x + y * y);
        if (nd <= 1.49012e-8 * nx || nd == 0) { ok = true; break; }
    }
    if (!ok) return false;
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
        xs = x; ys = y;
        return true;
    }
    return false;
}

// ---- row f-4: grey-level centre-of-gravity refinement of the fitted lines (util_cylinder.py:706-971) --------------
// numpy's summation of <= 8 values: sequential below 8, the 8-way tree at 8
__device__ float sum_f32_np(const float *a, int n)
{
    if (n < 8) {
        float r = 0.f;
        for (int i = 0; i < n; i++) r += a[i];
        return r;
    }
    float res = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    for (int i = 8; i < n; i++) res += a[i];
    return res;
}
__device__ double sum_f64_np(const double *a, int n)
{
    if (n < 8) {
        double r = 0.;
        for (int i = 0; i < n; i++) r += a[i];
        return r;
    }
    double res = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    for (int i = 8; i < n; i++) res += a[i];
    return res;
}

// compute_center_of_gravity_x / _y for one sample; returns true where the reference raises
__device__ bool cog_refine(const uint8_t *gray, int h, int w, int half, bool along_y, double fixed, double moving, double &out)
{
    const int L = along_y ? h : w, Lf = along_y ? w : h;
    int ifx = (int)rint(fixed);
    int lo = (int)floor(moving) - half, hi = (int)ceil(moving) + half + 1;
    if (lo < 0) lo = 0;
    if (hi > L) hi = L;
    out = moving;
    if (ifx < 0 || ifx >= Lf) return false;
    if (hi < 0) {
        int wrapped = L + hi;
        int len_roi = wrapped > lo ? wrapped - lo : 0;
        return len_roi != 0;
    }
    int n = hi > lo ? hi - lo : 0;
    if (n == 0) return false;
    float G[16];
    double prod[16];
    if (n > 16) n = 16;
    for (int k = 0; k < n; k++) {
        int idx = lo + k;
        uint8_t v = along_y ? gray[(size_t)idx * w + ifx] : gray[(size_t)ifx * w + idx];
        G[k] = (float)((double)v * (1.0 / 255));
    }
    float s = sum_f32_np(G, n);
    if (s == 0) return false;
    for (int k = 0; k < n; k++) prod[k] = (double)(lo + k) * (double)G[k];
    double cog = sum_f64_np(prod, n) / (double)s;
    double delta = cog - moving;
    if (fabs(delta) > 0.5) delta = delta > 0 ? 0.5 : -0.5;
    double nv = moving + delta;
    if (nv < 0) nv = 0;
    if (nv > L - 1) nv = L - 1;
    out = nv;
    return false;
}

// np.polyfit(x, y, 2) in streaming form (column norms, then Givens rotations of the scaled rows)
__device__ void polyfit2_stream(const float *xs, const float *ys, int n, double *coef)
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
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (row[k] == 0) continue;
            double a = R[k][k], b = row[k];
            double r = sqrt(a * a + b * b);
            double cg = a / r, sg = b / r;
#pragma unroll
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

// stable insertion sort of an index list by key (ascending)
__device__ void sort_by_key(int *ord, int n, const double *key)
{
    for (int a = 1; a < n; a++) {
        int o = ord[a];
        int b = a - 1;
        while (b >= 0 && key[ord[b]] > key[o]) { ord[b + 1] = ord[b]; b--; }
        ord[b + 1] = o;
    }
}

__global__ __launch_bounds__(256) void k_lines(const int *__restrict__ lab_h, const int *__restrict__ lab_v,
                                               const uint8_t *__restrict__ exp_h, const uint8_t *__restrict__ exp_v,
                                               const uint8_t *__restrict__ g7, int h, int w,
                                               const int *__restrict__ joints, FrameState *__restrict__ st,
                                               LinesWS *__restrict__ wsall, double *__restrict__ o_xy,
                                               int *__restrict__ o_id, int *__restrict__ o_n, double *__restrict__ o_center,
                                               const uint8_t *__restrict__ gray, int subpixel, int sp_window, double sp_step,
                                               float *__restrict__ sp_scratch, int sp_cap, int planar)
{
    const int f = blockIdx.x, t = threadIdx.x;
    FrameState &S = st[f];
    __shared__ int s_ng[2], s_ord[2][MAXL], s_n[2], s_ovf;
    __shared__ int s_pref[MAXL + 1];
    __shared__ double s_rv[256];
    __shared__ int s_ri[256];
    __shared__ int s_crow, s_ccol, s_total;
    LinesWS &W = wsall[f];
    if (t == 0) { o_n[f] = 0; o_center[2 * f] = 0; o_center[2 * f + 1] = 0; W.fin_n[0] = 0; W.fin_n[1] = 0; }
    if (S.status != CPE_ST_OK) return;
    const size_t N = (size_t)h * w;
    // union-find planes of the two expanded masks (unions done, not flattened): only the joints' labels are resolved
    const int *L[2] = {lab_h + f * N, lab_v + f * N};
    const uint8_t *E[2] = {exp_h + f * N, exp_v + f * N};
    const int *J = joints + (size_t)f * MAXJ * 2;
    const int nj = min(S.n_joints, MAXJ);
    if (t == 0) s_ovf = 0;
    __syncthreads();

    // group_points_by_label, groups in order of first appearance.
    // a) every thread resolves labels (union-find roots) of its share of the joints for both masks;
    // b) wavefront 0 / 1 walk the joints of side 0 / 1 in order with one lane per group (four groups per lane): a ballot
    //    finds the joint's group; the joint's rank in the group is the group's count at that moment;
    // c) group offsets = prefix sums of the counts, every joint goes to pool[offset of its group + rank] (all threads).
    for (int i = t; i < nj; i += 256) {
        const int jx = J[2 * i], jy = J[2 * i + 1];
        const bool inb = !(jx < 0 || jx >= w || jy < 0 || jy >= h);
        for (int sd = 0; sd < 2; sd++) {
            int lab = -1;
            if (inb && E[sd][(size_t)jy * w + jx]) lab = uf_find(L[sd], jy * w + jx);   // else: background label / skipped
            W.jlab[sd][i] = lab;
        }
    }
    __syncthreads();
    static_assert(MAXL == 256, "the grouping below gives every lane of a wavefront four groups");
    if (t < 128) {
        const int sd = t >> 6, lane = t & 63;
        int ng = 0, my_lab[4] = {-2, -2, -2, -2}, my_n[4] = {0, 0, 0, 0};   // lane k owns groups k, k + 64, k + 128, k + 192
        bool ovf = false;
        for (int i = 0; i < nj; i++) {
            const int lab = W.jlab[sd][i];
            int g = -1, rank = 0;
            if (lab >= 0) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const unsigned long long mb = __ballot(lane + 64 * q < ng && my_lab[q] == lab);
                    if (g < 0 && mb) g = 64 * q + __ffsll((long long)mb) - 1;
                }
                if (g < 0) {
                    if (ng == MAXL) ovf = true;
                    else {
                        g = ng++;
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            if (lane == (g & 63) && (g >> 6) == q) { my_lab[q] = lab; my_n[q] = 0; }
                    }
                }
                if (g >= 0) {
                    int cnt = 0;
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        if ((g >> 6) == q) cnt = my_n[q];
                    rank = __shfl(cnt, g & 63);
                    if (rank >= MAXLP) { ovf = true; g = -1; }
                    else {
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            if (lane == (g & 63) && (g >> 6) == q) my_n[q]++;
                    }
                }
            }
            if (lane == 0) { W.jgrp[sd][i] = g; W.jrank[sd][i] = rank; }
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (lane + 64 * q < ng) { W.glabel[sd][lane + 64 * q] = my_lab[q]; W.gn[sd][lane + 64 * q] = my_n[q]; }
        if (__ballot(ovf)) s_ovf = 1;
        if (lane == 0) s_ng[sd] = ng;
    }
    __syncthreads();
    if (t == 0 || t == 64) {
        const int sd = t >> 6;
        int acc = 0;
        for (int g = 0; g < s_ng[sd]; g++) { W.goff[sd][g] = acc; acc += W.gn[sd][g]; }
    }
    __syncthreads();
    for (int i = t; i < nj; i += 256)
        for (int sd = 0; sd < 2; sd++) {
            const int g = W.jgrp[sd][i];
            if (g < 0) continue;
            double *p = W.pool[sd][W.goff[sd][g] + W.jrank[sd][i]];
            p[0] = (double)J[2 * i];
            p[1] = (double)J[2 * i + 1];
        }
    __syncthreads();
    auto gp = [&](int sd, int slot) -> double (*)[2] { return W.pool[sd] + W.goff[sd][slot]; };
    // sort_rows: stable by min y (rows AND cols), then create_dummy_rows_cols + fit (degree 2)
    for (int idx = t; idx < 2 * MAXL; idx += 256) {
        const int sd = idx / MAXL, g = idx % MAXL;
        if (g < s_ng[sd]) {
            const int n = W.gn[sd][g];
            double (*P)[2] = gp(sd, g);
            double *scr = W.fit_scr[sd] + 7 * (size_t)W.goff[sd][g];
            int *ord = W.fit_ord[sd] + W.goff[sd][g];
            double m = P[0][1];
            for (int k = 1; k < n; k++) m = fmin(m, P[k][1]);
            W.key[sd][g] = m;
            s_ord[sd][g] = g;
            for (int k = 0; k < 6; k++) W.eq[sd][g][k] = 0;
            if (planar) {
                // degree 1; rows get their final +-50 domain, columns a provisional +-10 one (merged below)
                if (n >= 2) {
                    double c[2], lo, hi;
                    fit_line_sorted((const double (*)[2])P, n, sd == 0 ? 0 : 1, c, lo, hi, scr, ord);
                    const double mg = sd == 0 ? 50.0 : 10.0;
                    lo -= mg; hi += mg;
                    W.eq[sd][g][0] = c[0]; W.eq[sd][g][1] = c[1]; W.eq[sd][g][2] = lo; W.eq[sd][g][3] = hi; W.eq[sd][g][4] = fabs(lo - hi);
                }
            } else if (n >= 3) {
                double *tt = scr, *uu = tt + n;
                const int kc = sd == 0 ? 0 : 1;  // rows: y = f(x) sorted by x; cols: x = f(y) sorted by y
                for (int i = 0; i < n; i++) ord[i] = i;
                for (int a = 1; a < n; a++) {
                    int o = ord[a];
                    int b = a - 1;
                    while (b >= 0 && P[ord[b]][kc] > P[o][kc]) { ord[b + 1] = ord[b]; b--; }
                    ord[b + 1] = o;
                }
                for (int i = 0; i < n; i++) { tt[i] = P[ord[i]][kc]; uu[i] = P[ord[i]][1 - kc]; }
                double c[3];
                polyfit2(tt, uu, n, c, tt + 2 * n);
                double lo = tt[0] - 50, hi = tt[n - 1] + 50;
                W.eq[sd][g][0] = c[0]; W.eq[sd][g][1] = c[1]; W.eq[sd][g][2] = c[2];
                W.eq[sd][g][3] = lo; W.eq[sd][g][4] = hi; W.eq[sd][g][5] = fabs(hi - lo);
            }
        }
    }
    __syncthreads();
    if (t == 0 || t == 64) {
        const int sd = t >> 6;
        sort_by_key(s_ord[sd], s_ng[sd], W.key[sd]);
        int n = s_ng[sd];
        if (!planar) {
            // remove_label: first row, last col
            if (sd == 0) {
                if (n > 0) { for (int k = 0; k + 1 < n; k++) s_ord[0][k] = s_ord[0][k + 1]; n--; }
            } else {
                if (n > 0) n--;
            }
        } else if (sd == 1) {
            // fit_and_draw_polynomial of util_plane.py, steps 2-6: runs of consecutive "short" columns (domain <= 0.9 x
            // the longest) are merged while their domains add up to at most the longest, refitted, and take the place
            // of their first member; then every column gets its final +-50 domain
            double thr = 0;
            for (int k = 0; k < n; k++) thr = fmax(thr, fabs(W.eq[1][s_ord[1][k]][4]));
            int m = 0;          // columns kept so far (s_ord[1][0..m))
            int k = 0;
            int macc = MAXJ;    // merged lists go to the second half of the pool, one after the other
            while (k < n) {
                const int slot = s_ord[1][k];
                const double dk = fabs(W.eq[1][slot][4]);
                if (!(dk <= 0.9 * thr)) { s_ord[1][m++] = slot; k++; continue; }
                // a run starts here: take members while they are short and the sum stays within thr
                double cum = dk;
                int cnt = W.gn[1][slot];
                int k2 = k + 1;
                double (*D)[2] = W.pool[1] + macc;
                {
                    double (*P0)[2] = gp(1, slot);
                    for (int q = 0; q < cnt; q++) { D[q][0] = P0[q][0]; D[q][1] = P0[q][1]; }
                }
                while (k2 < n) {
                    const int s2 = s_ord[1][k2];
                    const double d2 = fabs(W.eq[1][s2][4]);
                    if (!(d2 <= 0.9 * thr) || cum + d2 > thr) break;
                    cum += d2;
                    double (*P2)[2] = gp(1, s2);
                    for (int q = 0; q < W.gn[1][s2]; q++) {
                        if (cnt < MAXLP) { D[cnt][0] = P2[q][0]; D[cnt][1] = P2[q][1]; cnt++; }
                        else s_ovf = 1;
                    }
                    k2++;
                }
                W.gn[1][slot] = cnt;
                W.goff[1][slot] = macc;
                macc += cnt;
                if (cnt >= 2) {
                    double c[2], lo, hi;
                    // (the fits of the single groups are done: their scratch is free)
                    fit_line_sorted((const double (*)[2])D, cnt, 1, c, lo, hi, W.fit_scr[1], W.fit_ord[1]);
                    W.eq[1][slot][0] = c[0]; W.eq[1][slot][1] = c[1]; W.eq[1][slot][2] = lo; W.eq[1][slot][3] = hi; W.eq[1][slot][4] = fabs(lo - hi);
                    s_ord[1][m++] = slot;
                }   // else: the members are deleted and nothing takes their place
                k = k2;
            }
            n = m;
            for (int q = 0; q < n; q++) {
                const int slot = s_ord[1][q];
                const int np = W.gn[1][slot];
                if (np < 2) continue;
                double (*P)[2] = gp(1, slot);
                double lo = P[0][1], hi = lo;
                for (int i = 1; i < np; i++) { lo = fmin(lo, P[i][1]); hi = fmax(hi, P[i][1]); }
                lo -= 50; hi += 50;
                W.eq[1][slot][2] = lo; W.eq[1][slot][3] = hi; W.eq[1][slot][4] = fabs(lo - hi);
            }
        }
        s_n[sd] = n;
    }
    __syncthreads();
    // a truncated joint list changes everything downstream: flag it now, the early returns below must not hide it
    if (t == 0 && s_ovf) set_overflow(S, OVF_LINES);
    const int nr = s_n[0], nc = s_n[1];
    if (subpixel) {
        // modify_grayscale_Cline(gray, rows, cols, degree 2, sample_step, window): sample every fitted line, pull each
        // sample to the grey-level centre of gravity across the line, re-fit.  Samples of line l live in
        // sp_scratch[(f * 2*MAXL + l) * 2 * sp_cap ...] as float32 (abscissa | refined ordinate), as the reference stores them.
        __shared__ int s_raise, s_K[2 * MAXL];
        if (t == 0) s_raise = 0;
        const int half = sp_window / 2;
        const uint8_t *gimg = gray + f * N;
        float *sbase = sp_scratch + (size_t)f * 2 * MAXL * 2 * sp_cap;
        for (int l = t; l < 2 * MAXL; l += 256) {
            const int sd = l / MAXL, pos = l % MAXL;
            int K = 0;
            if (pos < s_n[sd]) {
                const double *eq = W.eq[sd][s_ord[sd][pos]];
                if (!(eq[4] < eq[3])) {
                    double cntd = ceil(((eq[4] + 0.0001) - eq[3]) / sp_step);
                    K = cntd > 0 ? (cntd > (double)sp_cap ? sp_cap + 1 : (int)cntd) : 0;
                }
            }
            s_K[l] = K;
        }
        __syncthreads();
        for (int l = 0; l < 2 * MAXL; l++) {
            const int sd = l / MAXL, pos = l % MAXL;
            const int K = s_K[l];
            if (K == 0) continue;
            if (K > sp_cap) { if (t == 0) s_ovf = 1; continue; }
            const double *eq = W.eq[sd][s_ord[sd][pos]];
            const double lo = eq[3];
            const double delta = (lo + sp_step) - lo;
            float *xs = sbase + (size_t)l * 2 * sp_cap, *ys = xs + sp_cap;
            for (int i = t; i < K; i += 256) {
                double tt = i == 0 ? lo : (i == 1 ? lo + sp_step : lo + i * delta);
                double u = ((0.0 * tt + eq[0]) * tt + eq[1]) * tt + eq[2];
                double ref;
                if (cog_refine(gimg, h, w, half, sd == 0, tt, u, ref)) s_raise = 1;
                xs[i] = (float)tt;
                ys[i] = (float)ref;
            }
        }
        __syncthreads();
        if (t == 0 && s_ovf) set_overflow(S, OVF_LINES);
        if (s_raise) {
            if (t == 0) S.status = CPE_ST_SUBPIXEL_RAISED;
            return;
        }
        for (int l = t; l < 2 * MAXL; l += 256) {
            const int K = s_K[l];
            if (K >= 3 && K <= sp_cap) {
                const int sd = l / MAXL, pos = l % MAXL;
                double *eq = W.eq[sd][s_ord[sd][pos]];
                const float *xs = sbase + (size_t)l * 2 * sp_cap, *ys = xs + sp_cap;
                double c[3];
                polyfit2_stream(xs, ys, K, c);
                float mn = xs[0], mx = xs[0];
                for (int i = 1; i < K; i++) { mn = fminf(mn, xs[i]); mx = fmaxf(mx, xs[i]); }
                eq[0] = c[0]; eq[1] = c[1]; eq[2] = c[2];
                eq[3] = (double)mn; eq[4] = (double)mx; eq[5] = fabs((double)mx - (double)mn);
            }
        }
        __syncthreads();
    }
    // find_and_assign_intersections_P: all (row, col) pairs
    const int *rect = S.rect;
    for (int p = t; p < nr * nc; p += 256) {
        int r = p / nc, c = p - r * nc;
        double x, y;
        bool ok = planar ? line_intersection(W.eq[0][s_ord[0][r]], W.eq[1][s_ord[1][c]], x, y)
                         : poly_intersection(W.eq[0][s_ord[0][r]], W.eq[1][s_ord[1][c]], x, y);
        if (ok) ok = (rect[0] <= x && x <= rect[0] + rect[2]) && (rect[1] <= y && y <= rect[1] + rect[3]);
        W.ival[r][c] = ok ? 1 : 0;
        W.ixy[r][c][0] = x;
        W.ixy[r][c][1] = y;
    }
    __syncthreads();
    // per-line lists in loop order, and the clean_and_relabel keys (mean y for rows, mean x for cols)
    for (int l = t; l < 2 * MAXL; l += 256) {
        const int sd = l / MAXL, pos = l % MAXL;
        if (pos < s_n[sd]) {
            const int slot = s_ord[sd][pos];
            int k = 0;
            double sum = 0;
            const int other = sd == 0 ? nc : nr;
            for (int q = 0; q < other; q++) {
                int r = sd == 0 ? pos : q, c = sd == 0 ? q : pos;
                if (!W.ival[r][c]) continue;
                W.ipts[sd][slot][k][0] = W.ixy[r][c][0];   // k < other <= MAXL
                W.ipts[sd][slot][k][1] = W.ixy[r][c][1];
                sum += W.ixy[r][c][sd == 0 ? 1 : 0];
                k++;
            }
            W.in[sd][slot] = k;
            W.key[sd][slot] = k > 0 ? sum / k : 0.0;
        }
    }
    __syncthreads();
    if (t == 0 || t == 64) {
        const int sd = t >> 6;
        int n = 0;
        for (int k = 0; k < s_n[sd]; k++)
            if (W.in[sd][s_ord[sd][k]] > 0) s_ord[sd][n++] = s_ord[sd][k];
        if (!planar) sort_by_key(s_ord[sd], n, W.key[sd]);   // util_plane.py's clean_and_relabel keeps the order
        s_n[sd] = n;
        for (int k = 0; k < n; k++) W.fin_ord[sd][k] = s_ord[sd][k];
        W.fin_n[sd] = n;
    }
    __syncthreads();
    const int NR = s_n[0], NC = s_n[1];
    if (t == 0) { S.n_rows = NR; S.n_cols = NC; }
    if (NR == 0) {
        if (t == 0) S.status = CPE_ST_NO_LINES;
        return;
    }
    // ---- indexing_data: centre = first maximum of the blurred-window mean over the row points
    if (t == 0) {
        int acc = 0;
        for (int r = 0; r < NR; r++) { s_pref[r] = acc; acc += W.in[0][s_ord[0][r]]; }
        s_pref[NR] = acc;
    }
    __syncthreads();
    const int PR = s_pref[NR];
    int half = (int)(S.r0 / 5.0);
    if (half < 3) half = 3;
    if (half > 10) half = half + 5;
    if (planar) half = (int)(S.r0 / 4.5);   // util_plane.py:1280: no clamps (0 => every window is empty => mean = NaN)
    const uint8_t *G = g7 + f * N;
    double bv = -1e300;
    int bq = INT_MAX;
    __shared__ int s_first_nan;
    if (t == 0) s_first_nan = 0;
    __syncthreads();
    for (int q = t; q < PR; q += 256) {
        int r = 0;
        while (s_pref[r + 1] <= q) r++;
        const double *p = W.ipts[0][s_ord[0][r]][q - s_pref[r]];
        double x = p[0], y = p[1];
        int xs = (int)(x - half), xe = (int)(x + half), ys = (int)(y - half), ye = (int)(y + half);
        xs = max(xs, 0); xe = min(xe, w); ys = max(ys, 0); ye = min(ye, h);
        long cnt = (long)(xe > xs ? xe - xs : 0) * (ye > ys ? ye - ys : 0);
        double m = -1.0;   // stands for NaN (np.mean of an empty slice): never larger than anything
        if (cnt <= 0 && q == 0) s_first_nan = 1;
        if (cnt > 0) {
            unsigned long sum = 0;
            for (int yy = ys; yy < ye; yy++)
                for (int xx = xs; xx < xe; xx++) sum += G[(size_t)yy * w + xx];
            m = (double)sum / (double)cnt;
        }
        if (m > bv) { bv = m; bq = q; }   // q increases per thread: keeps the first maximum
    }
    s_rv[t] = bv; s_ri[t] = bq;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (t < off) {
            double ov = s_rv[t + off]; int oi = s_ri[t + off];
            if (ov > s_rv[t] || (ov == s_rv[t] && oi < s_ri[t])) { s_rv[t] = ov; s_ri[t] = oi; }
        }
        __syncthreads();
    }
    double cx, cy;
    {
        // max() over (mean, point) keeps the first item unless a later mean compares larger: a NaN first mean stays
        int q = s_first_nan ? 0 : s_ri[0], r = 0;
        while (s_pref[r + 1] <= q) r++;
        const double *p = W.ipts[0][s_ord[0][r]][q - s_pref[r]];
        cx = p[0]; cy = p[1];
    }
    __syncthreads();
    // centre row: first minimum distance over row points (in order); centre col: same over col points
    for (int side = 0; side < 2; side++) {
        double bd = 1e300;
        int bi = INT_MAX;
        const int nl = side == 0 ? NR : NC;
        // flattened order index = (line position << 8) | point index  (MAXL <= 256 points per line)
        for (int ln = 0; ln < nl; ln++) {
            const int slot = s_ord[side][ln];
            const int np = W.in[side][slot];
            for (int k = t; k < np; k += 256) {
                double ddx = cx - W.ipts[side][slot][k][0], ddy = cy - W.ipts[side][slot][k][1];
                double d = sqrt(ddx * ddx + ddy * ddy);
                int qi = (ln << 8) | k;
                if (d < bd || (d == bd && qi < bi)) { bd = d; bi = qi; }
            }
        }
        s_rv[t] = bd; s_ri[t] = bi;
        __syncthreads();
        for (int off = 128; off >= 1; off >>= 1) {
            if (t < off) {
                double ov = s_rv[t + off]; int oi = s_ri[t + off];
                if (ov < s_rv[t] || (ov == s_rv[t] && oi < s_ri[t])) { s_rv[t] = ov; s_ri[t] = oi; }
            }
            __syncthreads();
        }
        if (t == 0) {
            int v = s_ri[0] == INT_MAX ? -1 : (s_ri[0] >> 8);
            if (side == 0) s_crow = v; else s_ccol = v;
        }
        __syncthreads();
    }
    const int crow = s_crow, ccol = s_ccol;
    if (ccol < 0) {
        if (t == 0) S.status = CPE_ST_NO_LINES;
        return;
    }
    // cols_dict: every col point -> id (col - centre col, nearest row - centre row); keep col >= 0
    if (t == 0) {
        int acc = 0;
        for (int c = 0; c < NC; c++) { s_pref[c] = acc; acc += (planar || c >= ccol) ? W.in[1][s_ord[1][c]] : 0; }   // remove_minus_labels: cylinder script only
        s_pref[NC] = acc;
        s_total = acc;
    }
    __syncthreads();
    const int total = s_total;
    if (total == 0) {
        if (t == 0) S.status = CPE_ST_EMPTY;
        return;
    }
    if (total > CPE_MAXP) {     // more grid points than a table holds (include/cpe.h): CPE_ST_OVERFLOW, as in the oracle
        if (t == 0) set_overflow(S, OVF_LINES);
        return;
    }
    for (int e = t; e < total; e += 256) {
        int c = 0;
        while (s_pref[c + 1] <= e) c++;
        const double *p = W.ipts[1][s_ord[1][c]][e - s_pref[c]];
        double px = p[0], py = p[1];
        int nrw = -1;
        double mdd = INFINITY;
        for (int r = 0; r < NR; r++) {
            const int slot = s_ord[0][r];
            const int np = W.in[0][slot];
            for (int q = 0; q < np; q++) {
                double ddx = px - W.ipts[0][slot][q][0], ddy = py - W.ipts[0][slot][q][1];
                double d = sqrt(ddx * ddx + ddy * ddy);
                if (d < mdd) { mdd = d; nrw = r; }
            }
        }
        W.ent_xy[e][0] = px; W.ent_xy[e][1] = py;
        const int ci = c - ccol, ri = nrw >= 0 ? nrw - crow : 0;
        W.ent_id[e][0] = planar ? ri : ci;   // the planar script's ids are (row, col)
        W.ent_id[e][1] = planar ? ci : ri;
    }
    __syncthreads();
    // make_json: stable sort by (col,row)
    const int nout = min(total, CPE_MAXP);
    for (int e = t; e < total; e += 256) {
        int kc = W.ent_id[e][0], kr = W.ent_id[e][1], rank = 0;
        for (int j = 0; j < total; j++) {
            int jc = W.ent_id[j][0], jr = W.ent_id[j][1];
            bool less = jc < kc || (jc == kc && (jr < kr || (jr == kr && j < e)));
            rank += less ? 1 : 0;
        }
        if (rank < CPE_MAXP) {
            size_t o = (size_t)f * CPE_MAXP + rank;
            o_xy[2 * o] = W.ent_xy[e][0]; o_xy[2 * o + 1] = W.ent_xy[e][1];
            o_id[2 * o] = kc; o_id[2 * o + 1] = kr;
        }
    }
    if (t == 0) {
        o_n[f] = nout;
        o_center[2 * f] = cx; o_center[2 * f + 1] = cy;
        if (total > CPE_MAXP || s_ovf) set_overflow(S, OVF_LINES);
    }
}

// rows_updated / cols_updated of one frame (what color_and_expand_lines returns beside the JSON, util_cylinder.py:2044-2060):
// per surviving line, in final order, its equation and its intersection list in loop order
__global__ __launch_bounds__(256) void k_line_tables(const LinesWS *__restrict__ wsall, int f, double *__restrict__ eq,
                                                     int *__restrict__ npts, double *__restrict__ pts, int *__restrict__ n_lines)
{
    const LinesWS &W = wsall[f];
    const int t = threadIdx.x;
    if (t < 2) n_lines[t] = W.fin_n[t];
    for (int i = t; i < 2 * MAXL; i += 256) {
        const int sd = i / MAXL, pos = i % MAXL;
        const bool live = pos < W.fin_n[sd];
        const int slot = live ? W.fin_ord[sd][pos] : 0;
        npts[i] = live ? W.in[sd][slot] : 0;
        for (int k = 0; k < 6; k++) eq[i * 6 + k] = live ? W.eq[sd][slot][k] : 0.0;
    }
    for (int i = t; i < 2 * MAXL * MAXL; i += 256) {
        const int sd = i / (MAXL * MAXL), pos = (i / MAXL) % MAXL, k = i % MAXL;
        double x = 0, y = 0;
        if (pos < W.fin_n[sd]) {
            const int slot = W.fin_ord[sd][pos];
            if (k < W.in[sd][slot]) { x = W.ipts[sd][slot][k][0]; y = W.ipts[sd][slot][k][1]; }
        }
        pts[2 * i] = x; pts[2 * i + 1] = y;
    }
}

}  // namespace

int lines_export(const void *lines_ws, int f, double *eq, int *npts, double *pts, int *n_lines, hipStream_t s)
{
    CPE_LAUNCH_BEGIN();
    CPE_KLAUNCH(k_line_tables, dim3(1), dim3(256), 0, s, (const LinesWS *)lines_ws, f, eq, npts, pts, n_lines);
    CPE_CHECK_LAUNCH("k_line_tables");
    return CPE_OK;
}

int lines_stage(const int *lab_h, const int *lab_v, const uint8_t *exp_h, const uint8_t *exp_v, const uint8_t *g7, int n, int h, int w, const int *joints,
                FrameState *st, void *lines_ws, double *o_xy, int *o_id, int *o_n, double *o_center, const uint8_t *gray,
                int subpixel, int sp_window, double sp_step, float *sp_scratch, int sp_cap, hipStream_t s, int planar)
{
    CPE_LAUNCH_BEGIN();
    CPE_KLAUNCH(k_lines, dim3(n), dim3(256), 0, s, lab_h, lab_v, exp_h, exp_v, g7, h, w, joints, st, (LinesWS *)lines_ws, o_xy,
                       o_id, o_n, o_center, gray, subpixel, sp_window, sp_step, sp_scratch, sp_cap, planar);
    CPE_CHECK_LAUNCH("k_lines");
    return CPE_OK;
}

}  // namespace cpe
