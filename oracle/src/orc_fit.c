/*
 * ORACLE (test infrastructure) -- the geometric half of the hot path:
 *   utils/fitSingleCylinder.m:5-25, chooseIdx.m:19-104, findGridCorrespondences.m,
 *   triangulateWithThreshold.m:16-43, fitCylinderWPts3.m, getDistPts3ToLine.m, estCurvatures.m,
 *   fitplane.m, applyCylParamsPrior.m, cylParams2T.m, projPts3.m
 *
 * [ext] (MATLAB toolbox code absent from /root/reference; restated from its published behaviour,
 * SURVEY.md appendix B -- PARITY UNPINNED, no MATLAB/Octave in this image):
 *   triangulate  : per point 4x4 DLT, right singular vector of the smallest singular value
 *                  (one-sided Jacobi SVD here), error = mean over the two views of the pixel distance
 *   pca / eig    : cyclic Jacobi on the 3x3 covariance
 *   knnsearch    : brute force, ties by index, self included
 *   A\b          : 5x5 normal equations, Gaussian elimination with partial pivoting
 *   fminsearch   : Lagarias et al. Nelder-Mead exactly as MATLAB's fminsearch.m orders it
 *
 * Documented deviation (SURVEY.md section 7 hard-part 6): estCurvatures takes eig's FIRST principal
 * direction, which depends on the LAPACK-determined sign of fitplane's normal V(:,1) (fitplane.m:13-14).
 * That sign is not reproducible without MATLAB; here the normal is oriented away from the camera
 * (z >= 0, the same assumption fitCylinderWPts3.m:13-19 makes for rdir) and eig's first (ascending)
 * principal direction is then taken exactly as the reference does.
 *
 * Every reduction over the points of a frame uses the fixed 64-lane tree `sum64` (lane l adds
 * elements l, l+64, ... in order, then xor-butterfly 32..1), which is what one CDNA wavefront
 * executes -- so the HIP kernels reproduce these numbers bit for bit.
 */
#include "orc_common.h"
#include <float.h>
#include <stdio.h>

/* ------------------------------------------------------------------ reductions */
static double sum64(const double *v, int n)
{
    double p[64];
    for (int l = 0; l < 64; l++) {
        double a = 0.0;
        for (int k = l; k < n; k += 64) a = a + v[k];
        p[l] = a;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        double q[64];
        for (int l = 0; l < 64; l++) q[l] = p[l] + p[l ^ off];
        memcpy(p, q, sizeof(p));
    }
    return p[0];
}

/* ------------------------------------------------------------------ triangulate [ext] */
static void mat34_from_K_Rt(const double *K, const double *T, double *P) /* P = K * T(1:3,:) */
{
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) {
            double s = 0.0;
            for (int k = 0; k < 3; k++) s = s + K[r * 3 + k] * T[k * 4 + c];
            P[r * 4 + c] = s;
        }
}

/* one-sided Jacobi SVD of a 4x4: returns the right singular vector of the smallest singular value */
static void svd4_null(const double *Ain, double *x)
{
    double U[16], V[16];
    memcpy(U, Ain, sizeof(U));
    for (int i = 0; i < 16; i++) V[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; sweep++) {
        int rotated = 0;
        for (int p = 0; p < 3; p++)
            for (int q = p + 1; q < 4; q++) {
                double al = 0.0, be = 0.0, ga = 0.0;
                for (int k = 0; k < 4; k++) {
                    al = al + U[k * 4 + p] * U[k * 4 + p];
                    be = be + U[k * 4 + q] * U[k * 4 + q];
                    ga = ga + U[k * 4 + p] * U[k * 4 + q];
                }
                if (fabs(ga) <= 1e-15 * sqrt(al * be)) continue;
                rotated = 1;
                double zeta = (be - al) / (2.0 * ga);
                double t = 1.0 / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                if (zeta < 0) t = -t;
                double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int k = 0; k < 4; k++) {
                    double up = U[k * 4 + p], uq = U[k * 4 + q];
                    U[k * 4 + p] = c * up - s * uq;
                    U[k * 4 + q] = s * up + c * uq;
                    double vp = V[k * 4 + p], vq = V[k * 4 + q];
                    V[k * 4 + p] = c * vp - s * vq;
                    V[k * 4 + q] = s * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
    int jm = 0;
    double best = 0;
    for (int j = 0; j < 4; j++) {
        double nn = 0.0;
        for (int k = 0; k < 4; k++) nn = nn + U[k * 4 + j] * U[k * 4 + j];
        if (j == 0 || nn < best) { best = nn; jm = j; }
    }
    for (int k = 0; k < 4; k++) x[k] = V[k * 4 + jm];
}

static void project(const double *P, const double *X, double *u, double *v)
{
    double a = ((P[0] * X[0] + P[1] * X[1]) + P[2] * X[2]) + P[3];
    double b = ((P[4] * X[0] + P[5] * X[1]) + P[6] * X[2]) + P[7];
    double c = ((P[8] * X[0] + P[9] * X[1]) + P[10] * X[2]) + P[11];
    *u = a / c;
    *v = b / c;
}

/* triangulate(p1, p2, stereoParams): fitSingleCylinder.m:15, chooseIdx.m:57 */
ORC_API void orc_triangulate(const double *p1, const double *p2, int n, const double *K1, const double *K2,
                             const double *T21, double *X, double *err)
{
    static const double I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    double P1[12], P2[12];
    mat34_from_K_Rt(K1, I4, P1);
    mat34_from_K_Rt(K2, T21, P2);
    for (int i = 0; i < n; i++) {
        double A[16], x[4];
        for (int c = 0; c < 4; c++) {
            A[0 * 4 + c] = p1[2 * i] * P1[8 + c] - P1[c];
            A[1 * 4 + c] = p1[2 * i + 1] * P1[8 + c] - P1[4 + c];
            A[2 * 4 + c] = p2[2 * i] * P2[8 + c] - P2[c];
            A[3 * 4 + c] = p2[2 * i + 1] * P2[8 + c] - P2[4 + c];
        }
        svd4_null(A, x);
        double Xi[3] = {x[0] / x[3], x[1] / x[3], x[2] / x[3]};
        X[3 * i] = Xi[0]; X[3 * i + 1] = Xi[1]; X[3 * i + 2] = Xi[2];
        double u, v, e1, e2, dx, dy;
        project(P1, Xi, &u, &v);
        dx = p1[2 * i] - u; dy = p1[2 * i + 1] - v;
        e1 = sqrt(dx * dx + dy * dy);
        project(P2, Xi, &u, &v);
        dx = p2[2 * i] - u; dy = p2[2 * i + 1] - v;
        e2 = sqrt(dx * dx + dy * dy);
        err[i] = (e1 + e2) / 2.0;
    }
}

/* ------------------------------------------------------------------ index selection */
/* findGridCorrespondences.m: equi-join in gp1 order (first match in gp2) */
ORC_API int orc_find_correspondences(const double *gp1, int n1, const double *gp2, int n2, double *c1,
                                     double *c2, int *idx)
{
    int m = 0;
    for (int i = 0; i < n1; i++) {
        for (int j = 0; j < n2; j++)
            if (gp2[4 * j + 2] == gp1[4 * i + 2] && gp2[4 * j + 3] == gp1[4 * i + 3]) {
                c1[2 * m] = gp1[4 * i]; c1[2 * m + 1] = gp1[4 * i + 1];
                c2[2 * m] = gp2[4 * j]; c2[2 * m + 1] = gp2[4 * j + 1];
                if (idx) { idx[2 * m] = (int)gp1[4 * i + 2]; idx[2 * m + 1] = (int)gp1[4 * i + 3]; }
                m++;
                break;
            }
    }
    return m;
}

static int find_row(const double *gp, int n, int c, int r)
{
    for (int i = 0; i < n; i++)
        if ((int)gp[4 * i + 2] == c && (int)gp[4 * i + 3] == r) return i;
    return -1;
}

/* containers.Map char keys 'c_r' sort as strings (chooseIdx.m:69,89) */
static int key_cmp(const void *a, const void *b)
{
    const int *ka = (const int *)a, *kb = (const int *)b;
    char sa[32], sb[32];
    snprintf(sa, sizeof sa, "%d_%d", ka[0], ka[1]);
    snprintf(sb, sizeof sb, "%d_%d", kb[0], kb[1]);
    return strcmp(sa, sb);
}

static int int_cmp(const void *a, const void *b) { return *(const int *)a - *(const int *)b; }

static int uniq_sorted(const double *gp, int n, int col, int *out)
{
    int m = 0;
    for (int i = 0; i < n; i++) out[m++] = (int)gp[4 * i + col];
    qsort(out, m, sizeof(int), int_cmp);
    int u = 0;
    for (int i = 0; i < m; i++)
        if (u == 0 || out[i] != out[u - 1]) out[u++] = out[i];
    return u;
}

/* chooseIdx(gp1, gp2, ., stereoParams, patchSize, error_th): returns count; *fallback set when the
 * plain join was used (chooseIdx.m:101-104). idx (optional) receives the (col,row) of each output row. */
ORC_API int orc_choose_idx(const double *gp1, int n1, const double *gp2, int n2, const double *K1,
                           const double *K2, const double *T21, int patch, double th, double *c1,
                           double *c2, int *idx, int *fallback)
{
    int *ux = (int *)malloc((size_t)(n1 + 1) * sizeof(int)), *uy = (int *)malloc((size_t)(n1 + 1) * sizeof(int));
    int nx = uniq_sorted(gp1, n1, 2, ux), ny = uniq_sorted(gp1, n1, 3, uy);
    /* map: key (c,r) -> (row in gp1, row in gp2, best error) */
    int cap = n1 + 1, nk = 0;
    int *keys = (int *)malloc((size_t)cap * 4 * sizeof(int)); /* c, r, loc1, loc2 */
    double *kerr = (double *)malloc((size_t)cap * sizeof(double));
    int np = patch * patch;
    double *q1 = (double *)malloc((size_t)np * 2 * sizeof(double)), *q2 = (double *)malloc((size_t)np * 2 * sizeof(double));
    double *X = (double *)malloc((size_t)np * 3 * sizeof(double)), *er = (double *)malloc((size_t)np * sizeof(double));
    int *l1 = (int *)malloc((size_t)np * sizeof(int)), *l2 = (int *)malloc((size_t)np * sizeof(int));
    int *cc = (int *)malloc((size_t)np * 2 * sizeof(int));
    for (int ix = 0; ix + patch <= nx; ix++)
        for (int iy = 0; iy + patch <= ny; iy++) {
            int ok = 1, m = 0;
            for (int a = 0; a < patch && ok; a++)
                for (int b = 0; b < patch; b++) {
                    int c = ux[ix + a], r = uy[iy + b];
                    int i1 = find_row(gp1, n1, c, r), i2 = find_row(gp2, n2, c, r);
                    if (i1 < 0 || i2 < 0) { ok = 0; break; }
                    cc[2 * m] = c; cc[2 * m + 1] = r; l1[m] = i1; l2[m] = i2;
                    q1[2 * m] = gp1[4 * i1]; q1[2 * m + 1] = gp1[4 * i1 + 1];
                    q2[2 * m] = gp2[4 * i2]; q2[2 * m + 1] = gp2[4 * i2 + 1];
                    m++;
                }
            if (!ok) continue;
            orc_triangulate(q1, q2, np, K1, K2, T21, X, er);
            double s = 0.0;
            for (int k = 0; k < np; k++) s = s + er[k];
            if (!(s / np < th)) continue;
            for (int k = 0; k < np; k++) {
                int f = -1;
                for (int j = 0; j < nk; j++)
                    if (keys[4 * j] == cc[2 * k] && keys[4 * j + 1] == cc[2 * k + 1]) { f = j; break; }
                if (f < 0) {
                    if (nk == cap) {
                        cap *= 2;
                        keys = (int *)realloc(keys, (size_t)cap * 4 * sizeof(int));
                        kerr = (double *)realloc(kerr, (size_t)cap * sizeof(double));
                    }
                    keys[4 * nk] = cc[2 * k]; keys[4 * nk + 1] = cc[2 * k + 1];
                    keys[4 * nk + 2] = l1[k]; keys[4 * nk + 3] = l2[k];
                    kerr[nk] = er[k];
                    nk++;
                } else if (er[k] < kerr[f]) {
                    kerr[f] = er[k]; keys[4 * f + 2] = l1[k]; keys[4 * f + 3] = l2[k];
                }
            }
        }
    int m;
    if (nk == 0) {
        *fallback = 1;
        m = orc_find_correspondences(gp1, n1, gp2, n2, c1, c2, idx);
    } else {
        *fallback = 0;
        qsort(keys, nk, 4 * sizeof(int), key_cmp);
        for (int j = 0; j < nk; j++) {
            int i1 = keys[4 * j + 2], i2 = keys[4 * j + 3];
            c1[2 * j] = gp1[4 * i1]; c1[2 * j + 1] = gp1[4 * i1 + 1];
            c2[2 * j] = gp2[4 * i2]; c2[2 * j + 1] = gp2[4 * i2 + 1];
            if (idx) { idx[2 * j] = keys[4 * j]; idx[2 * j + 1] = keys[4 * j + 1]; }
        }
        m = nk;
    }
    free(ux); free(uy); free(keys); free(kerr); free(q1); free(q2); free(X); free(er); free(l1); free(l2); free(cc);
    return m;
}

/* triangulateWithThreshold.m:16-43 */
ORC_API int orc_triangulate_with_threshold(const double *gp1, int n1, const double *gp2, int n2,
                                           const double *K1, const double *K2, const double *T21, double th,
                                           double *c1, double *c2, int *idx, int *fallback)
{
    int m = orc_find_correspondences(gp1, n1, gp2, n2, c1, c2, idx);
    *fallback = 0;
    if (m == 0) return 0;
    double *X = (double *)malloc((size_t)m * 3 * sizeof(double)), *er = (double *)malloc((size_t)m * sizeof(double));
    orc_triangulate(c1, c2, m, K1, K2, T21, X, er);
    int k = 0;
    for (int i = 0; i < m; i++)
        if (er[i] < th) {
            c1[2 * k] = c1[2 * i]; c1[2 * k + 1] = c1[2 * i + 1];
            c2[2 * k] = c2[2 * i]; c2[2 * k + 1] = c2[2 * i + 1];
            if (idx) { idx[2 * k] = idx[2 * i]; idx[2 * k + 1] = idx[2 * i + 1]; }
            k++;
        }
    free(X); free(er);
    if (k == 0) {
        *fallback = 1;
        return orc_find_correspondences(gp1, n1, gp2, n2, c1, c2, idx);
    }
    return k;
}

/* ------------------------------------------------------------------ small dense algebra */
/* cyclic Jacobi on a symmetric 3x3; eigenvalues ascending in w, eigenvectors in columns of V */
static void eig3(const double *Ain, double *w, double *V)
{
    double A[9];
    memcpy(A, Ain, sizeof(A));
    for (int i = 0; i < 9; i++) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; sweep++) {
        int rotated = 0;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                double apq = A[p * 3 + q];
                if (fabs(apq) <= 1e-17 * (fabs(A[p * 3 + p]) + fabs(A[q * 3 + q]))) continue;
                rotated = 1;
                double theta = (A[q * 3 + q] - A[p * 3 + p]) / (2.0 * apq);
                double t = 1.0 / (fabs(theta) + sqrt(theta * theta + 1.0));
                if (theta < 0) t = -t;
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; k++) { /* A <- A J */
                    double akp = A[k * 3 + p], akq = A[k * 3 + q];
                    A[k * 3 + p] = c * akp - s * akq;
                    A[k * 3 + q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; k++) { /* A <- J^T A */
                    double apk = A[p * 3 + k], aqk = A[q * 3 + k];
                    A[p * 3 + k] = c * apk - s * aqk;
                    A[q * 3 + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; k++) {
                    double vkp = V[k * 3 + p], vkq = V[k * 3 + q];
                    V[k * 3 + p] = c * vkp - s * vkq;
                    V[k * 3 + q] = s * vkp + c * vkq;
                }
            }
        if (!rotated) break;
    }
    int o[3] = {0, 1, 2};
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2 - i; j++)
            if (A[o[j + 1] * 4] < A[o[j] * 4]) { int t = o[j]; o[j] = o[j + 1]; o[j + 1] = t; }
    double Vs[9];
    for (int j = 0; j < 3; j++) {
        w[j] = A[o[j] * 4];
        for (int k = 0; k < 3; k++) Vs[k * 3 + j] = V[k * 3 + o[j]];
    }
    memcpy(V, Vs, sizeof(Vs));
}

/* getDistPts3ToLine.m: distance of P (n x 3, row i = point i) to the line p1 -> p2 */
static void dist_to_line(const double *P, int n, const double *p1, const double *p2, double *d)
{
    double v[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
    double nv2 = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2];
    for (int i = 0; i < n; i++) {
        const double *x = P + 3 * i;
        double al = (((x[0] - p1[0]) * v[0] + (x[1] - p1[1]) * v[1]) + (x[2] - p1[2]) * v[2]) / nv2;
        double e0 = x[0] - (p1[0] + v[0] * al), e1 = x[1] - (p1[1] + v[1] * al), e2 = x[2] - (p1[2] + v[2] * al);
        d[i] = sqrt((e0 * e0 + e1 * e1) + e2 * e2);
    }
}

ORC_API void orc_dist_pts3_to_line(const double *P, int n, const double *p1, const double *p2, double *d)
{
    dist_to_line(P, n, p1, p2, d);
}

/* dist() of fitCylinderWPts3.m:44-49: sum((d - R)^2) */
static double cyl_objective(const double *x, const double *P, int n, double R, double *tmp)
{
    double p2[3] = {x[0] + x[3], x[1] + x[4], x[2] + x[5]};
    dist_to_line(P, n, x, p2, tmp);
    for (int i = 0; i < n; i++) {
        double v = tmp[i] - R;
        tmp[i] = v * v;
    }
    return sum64(tmp, n);
}

ORC_API double orc_cyl_objective(const double *x, const double *P, int n, double R)
{
    double *tmp = (double *)malloc((size_t)(n + 1) * sizeof(double));
    double f = cyl_objective(x, P, n, R, tmp);
    free(tmp);
    return f;
}

static double eps_of(double x) /* MATLAB eps(x) */
{
    x = fabs(x);
    if (x < DBL_MIN) return 4.9406564584124654e-324;
    int e;
    frexp(x, &e); /* x = m * 2^e, m in [0.5,1) */
    return ldexp(1.0, e - 53);
}

/* fminsearch clone (MATLAB fminsearch.m). n = 6. */
typedef double (*nm_objective)(const double *x, void *ctx);
typedef struct { const double *P; int n; double R; double *tmp; } cyl_ctx;
static double cyl_obj_cb(const double *x, void *c_)
{
    cyl_ctx *c = (cyl_ctx *)c_;
    return cyl_objective(x, c->P, c->n, c->R, c->tmp);
}

static void nelder_mead6_fn(const double *x0, nm_objective fobj, void *ctx, double tolx, double tolf, int maxiter,
                            int maxfun, double *xout, double *fout, int *iters, int *evals);

static void nelder_mead6(const double *x0, const double *P, int n, double R, double tolx, double tolf,
                         int maxiter, int maxfun, double *xout, double *fout, int *iters, int *evals)
{
    cyl_ctx c = {P, n, R, (double *)malloc((size_t)(n + 1) * sizeof(double))};
    nelder_mead6_fn(x0, cyl_obj_cb, &c, tolx, tolf, maxiter, maxfun, xout, fout, iters, evals);
    free(c.tmp);
}

static void nelder_mead6_fn(const double *x0, nm_objective fobj, void *ctx, double tolx, double tolf, int maxiter,
                            int maxfun, double *xout, double *fout, int *iters, int *evals)
{
    enum { N = 6 };
    double v[N + 1][N], fv[N + 1];
    memcpy(v[0], x0, sizeof(double) * N);
    fv[0] = fobj(v[0], ctx);
    for (int j = 0; j < N; j++) {
        memcpy(v[j + 1], x0, sizeof(double) * N);
        if (v[j + 1][j] != 0) v[j + 1][j] = (1 + 0.05) * v[j + 1][j];
        else v[j + 1][j] = 0.00025;
        fv[j + 1] = fobj(v[j + 1], ctx);
    }
    int func_evals = N + 1, itercount = 1;
#define SORT_SIMPLEX()                                                             \
    for (int a_ = 1; a_ <= N; a_++) { /* stable insertion sort ascending */        \
        double fk = fv[a_], vk[N];                                                 \
        memcpy(vk, v[a_], sizeof vk);                                              \
        int b_ = a_ - 1;                                                           \
        while (b_ >= 0 && fv[b_] > fk) {                                           \
            fv[b_ + 1] = fv[b_];                                                   \
            memcpy(v[b_ + 1], v[b_], sizeof vk);                                   \
            b_--;                                                                  \
        }                                                                          \
        fv[b_ + 1] = fk;                                                           \
        memcpy(v[b_ + 1], vk, sizeof vk);                                          \
    }
    SORT_SIMPLEX();
    while (func_evals < maxfun && itercount < maxiter) {
        double df = 0, dx = 0, vmax = v[0][0];
        for (int j = 1; j <= N; j++) {
            double a = fabs(fv[0] - fv[j]);
            if (a > df) df = a;
            for (int k = 0; k < N; k++) {
                double b = fabs(v[j][k] - v[0][k]);
                if (b > dx) dx = b;
            }
        }
        for (int k = 1; k < N; k++)
            if (v[0][k] > vmax) vmax = v[0][k];
        double tf = 10 * eps_of(fv[0]), tx = 10 * eps_of(vmax);
        if (df <= (tolf > tf ? tolf : tf) && dx <= (tolx > tx ? tolx : tx)) break;

        double xbar[N], xr[N], xe[N], xc[N];
        for (int k = 0; k < N; k++) {
            double s = v[0][k];
            for (int j = 1; j < N; j++) s = s + v[j][k];
            xbar[k] = s / N;
        }
        for (int k = 0; k < N; k++) xr[k] = 2.0 * xbar[k] - 1.0 * v[N][k];
        double fxr = fobj(xr, ctx);
        func_evals++;
        int shrink = 0;
        if (fxr < fv[0]) {
            for (int k = 0; k < N; k++) xe[k] = 3.0 * xbar[k] - 2.0 * v[N][k];
            double fxe = fobj(xe, ctx);
            func_evals++;
            if (fxe < fxr) { memcpy(v[N], xe, sizeof xe); fv[N] = fxe; }
            else { memcpy(v[N], xr, sizeof xr); fv[N] = fxr; }
        } else if (fxr < fv[N - 1]) {
            memcpy(v[N], xr, sizeof xr); fv[N] = fxr;
        } else if (fxr < fv[N]) {
            for (int k = 0; k < N; k++) xc[k] = 1.5 * xbar[k] - 0.5 * v[N][k];
            double fxc = fobj(xc, ctx);
            func_evals++;
            if (fxc <= fxr) { memcpy(v[N], xc, sizeof xc); fv[N] = fxc; }
            else shrink = 1;
        } else {
            for (int k = 0; k < N; k++) xc[k] = 0.5 * xbar[k] + 0.5 * v[N][k];
            double fxcc = fobj(xc, ctx);
            func_evals++;
            if (fxcc < fv[N]) { memcpy(v[N], xc, sizeof xc); fv[N] = fxcc; }
            else shrink = 1;
        }
        if (shrink) {
            for (int j = 1; j <= N; j++) {
                for (int k = 0; k < N; k++) v[j][k] = v[0][k] + 0.5 * (v[j][k] - v[0][k]);
                fv[j] = fobj(v[j], ctx);
            }
            func_evals += N;
        }
        SORT_SIMPLEX();
        itercount++;
    }
#undef SORT_SIMPLEX
    memcpy(xout, v[0], sizeof(double) * N);
    *fout = fv[0];
    *iters = itercount;
    *evals = func_evals;
}

/* Levenberg-Marquardt on the same objective -- restatement of the build's own fast mode (csrc/fit.hip),
 * NOT of anything in the reference (fitCylinderWPts3.m:38 runs fminsearch).  Same operation order as the kernel. */
static void lm6(const double *x0, double f0, const double *P, int n, double R, double tolx, double tolf, int maxiter,
                double *xout, double *fout, int *iters, int *evals)
{
    double x[6], *tmp = (double *)malloc((size_t)(n + 1) * sizeof(double));
    double *cols = (double *)malloc((size_t)(n + 1) * 27 * sizeof(double));
    memcpy(x, x0, sizeof x);
    double fx = f0, lambda = 1e-3;
    int itercount = 0, func_evals = 1;
    for (; itercount < maxiter && itercount < 200;) {
        double A[21], g[6];
        {
            double p2[3] = {x[0] + x[3], x[1] + x[4], x[2] + x[5]};
            double vv[3] = {p2[0] - x[0], p2[1] - x[1], p2[2] - x[2]};
            double nv2 = (vv[0] * vv[0] + vv[1] * vv[1]) + vv[2] * vv[2];
            for (int k = 0; k < n; k++) {
                const double *pt = P + 3 * k;
                double *c = cols + (size_t)k * 27;
                for (int q = 0; q < 27; q++) c[q] = 0.0;
                double al = (((pt[0] - x[0]) * vv[0] + (pt[1] - x[1]) * vv[1]) + (pt[2] - x[2]) * vv[2]) / nv2;
                double e[3] = {pt[0] - (x[0] + vv[0] * al), pt[1] - (x[1] + vv[1] * al), pt[2] - (x[2] + vv[2] * al)};
                double dd = sqrt((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]);
                if (dd > 0) {
                    double r = dd - R, c1 = -1.0 / dd, c2 = -(al / dd);
                    double j[6] = {c1 * e[0], c1 * e[1], c1 * e[2], c2 * e[0], c2 * e[1], c2 * e[2]};
                    int q = 0;
                    for (int a = 0; a < 6; a++) {
                        for (int b = a; b < 6; b++) { c[q] = j[a] * j[b]; q++; }
                        c[21 + a] = j[a] * r;
                    }
                }
            }
            /* the kernel accumulates  acc = acc + term  per lane (skipping points with d == 0, whose terms are 0 here:
             * adding +0.0 leaves every partial sum unchanged), then the 64-lane tree */
            for (int q = 0; q < 27; q++) {
                for (int k = 0; k < n; k++) tmp[k] = cols[(size_t)k * 27 + q];
                double sres = sum64(tmp, n);
                if (q < 21) A[q] = sres; else g[q - 21] = sres;
            }
        }
        itercount++;
        int accepted = 0;
        double dmax = 0, fprev = fx;
        for (int tr = 0; tr < 12 && !accepted; tr++) {
            double M[36], rhs[6], dl[6];
            {
                int q = 0;
                double trA = 0;
                for (int a = 0; a < 6; a++)
                    for (int b = a; b < 6; b++) { M[a * 6 + b] = A[q]; M[b * 6 + a] = A[q]; if (a == b) trA = trA + A[q]; q++; }
                for (int a = 0; a < 6; a++) { M[a * 6 + a] = M[a * 6 + a] + lambda * M[a * 6 + a] + 1e-12 * trA; rhs[a] = -g[a]; }
            }
            int singular = 0;
            for (int c = 0; c < 6; c++) {
                int pv = c;
                for (int r = c + 1; r < 6; r++)
                    if (fabs(M[r * 6 + c]) > fabs(M[pv * 6 + c])) pv = r;
                if (M[pv * 6 + c] == 0) { singular = 1; break; }
                if (pv != c) {
                    for (int k = 0; k < 6; k++) { double t_ = M[c * 6 + k]; M[c * 6 + k] = M[pv * 6 + k]; M[pv * 6 + k] = t_; }
                    double t_ = rhs[c]; rhs[c] = rhs[pv]; rhs[pv] = t_;
                }
                for (int r = c + 1; r < 6; r++) {
                    double fct = M[r * 6 + c] / M[c * 6 + c];
                    for (int k = c; k < 6; k++) M[r * 6 + k] = M[r * 6 + k] - fct * M[c * 6 + k];
                    rhs[r] = rhs[r] - fct * rhs[c];
                }
            }
            if (singular) { lambda = lambda * 10; continue; }
            for (int r = 5; r >= 0; r--) {
                double sacc = rhs[r];
                for (int k = r + 1; k < 6; k++) sacc = sacc - M[r * 6 + k] * dl[k];
                dl[r] = sacc / M[r * 6 + r];
            }
            double xn[6];
            for (int k = 0; k < 6; k++) xn[k] = x[k] + dl[k];
            double fn = cyl_objective(xn, P, n, R, tmp);
            func_evals++;
            if (fn < fx) {
                dmax = 0;
                for (int k = 0; k < 6; k++) { dmax = fmax(dmax, fabs(dl[k])); x[k] = xn[k]; }
                fx = fn;
                lambda = fmax(lambda / 10, 1e-12);
                accepted = 1;
            } else {
                lambda = lambda * 10;
            }
        }
        if (!accepted) break;
        if ((fprev - fx) <= tolf * 1e-3 * (1.0 + fx) && dmax <= tolx) break;
    }
    memcpy(xout, x, sizeof x);
    *fout = fx;
    *iters = itercount;
    *evals = func_evals;
    free(tmp); free(cols);
}

/* solve M x = b (5x5), partial pivoting; M, b destroyed */
static void solve5(double *M, double *b, double *x)
{
    enum { N = 5 };
    for (int c = 0; c < N; c++) {
        int pv = c;
        for (int r = c + 1; r < N; r++)
            if (fabs(M[r * N + c]) > fabs(M[pv * N + c])) pv = r;
        if (pv != c) {
            for (int k = 0; k < N; k++) { double t = M[c * N + k]; M[c * N + k] = M[pv * N + k]; M[pv * N + k] = t; }
            double t = b[c]; b[c] = b[pv]; b[pv] = t;
        }
        for (int r = c + 1; r < N; r++) {
            double f = M[r * N + c] / M[c * N + c];
            for (int k = c; k < N; k++) M[r * N + k] = M[r * N + k] - f * M[c * N + k];
            b[r] = b[r] - f * b[c];
        }
    }
    for (int r = N - 1; r >= 0; r--) {
        double s = b[r];
        for (int k = r + 1; k < N; k++) s = s - M[r * N + k] * x[k];
        x[r] = s / M[r * N + r];
    }
}

/* estCurvatures.m for ONE point i (only K(:,1,i) is consumed, fitCylinderWPts3.m:29):
 * returns K(:,1,i), with the plane normal's sign fixed to z >= 0 (documented deviation, see header) */
static void est_curv_dir(const double *P, int n, int i, double *dir)
{
    int K = n < 20 ? n : 20;
    int nb[20];
    double *d2 = (double *)malloc((size_t)n * sizeof(double));
    char *used = (char *)calloc((size_t)n, 1);
    for (int j = 0; j < n; j++) {
        double a = P[3 * j] - P[3 * i], b = P[3 * j + 1] - P[3 * i + 1], c = P[3 * j + 2] - P[3 * i + 2];
        d2[j] = (a * a + b * b) + c * c;
    }
    for (int k = 0; k < K; k++) { /* knnsearch: ascending distance, ties by index */
        int bj = -1;
        for (int j = 0; j < n; j++)
            if (!used[j] && (bj < 0 || d2[j] < d2[bj])) bj = j;
        used[bj] = 1;
        nb[k] = bj;
    }
    free(d2); free(used);
    /* fitplane.m: cov -> eigenvector of the smallest eigenvalue */
    double mu[3] = {0, 0, 0};
    for (int k = 0; k < K; k++)
        for (int c = 0; c < 3; c++) mu[c] = mu[c] + P[3 * nb[k] + c];
    for (int c = 0; c < 3; c++) mu[c] = mu[c] / K;
    double Cv[9] = {0};
    for (int k = 0; k < K; k++) {
        double e[3] = {P[3 * nb[k]] - mu[0], P[3 * nb[k] + 1] - mu[1], P[3 * nb[k] + 2] - mu[2]};
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) Cv[a * 3 + b] = Cv[a * 3 + b] + e[a] * e[b];
    }
    for (int a = 0; a < 9; a++) Cv[a] = Cv[a] / (K - 1);
    double w[3], V[9];
    eig3(Cv, w, V);
    double z[3] = {V[0], V[3], V[6]};
    if (z[2] < 0) { z[0] = -z[0]; z[1] = -z[1]; z[2] = -z[2]; } /* sign fixed like rdir (see header) */
    /* createLocCoordSys (estCurvatures.m:20-29) -- not normalised, as in the reference */
    double x[3] = {1, 0, 0};
    if (fabs(z[0]) > 0.9) { x[0] = 0; x[1] = 1; }
    double y[3] = {z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0]};
    double xx[3] = {y[1] * z[2] - y[2] * z[1], y[2] * z[0] - y[0] * z[2], y[0] * z[1] - y[1] * z[0]};
    /* fitquadsurf (estCurvatures.m:31-38) */
    double M[25] = {0}, rhs[5] = {0}, co[5];
    for (int k = 0; k < K; k++) {
        double e[3] = {P[3 * nb[k]] - mu[0], P[3 * nb[k] + 1] - mu[1], P[3 * nb[k] + 2] - mu[2]};
        double lx = (e[0] * xx[0] + e[1] * xx[1]) + e[2] * xx[2];
        double ly = (e[0] * y[0] + e[1] * y[1]) + e[2] * y[2];
        double lz = (e[0] * z[0] + e[1] * z[1]) + e[2] * z[2];
        double row[5] = {lx * lx, lx * ly, ly * ly, lx, ly};
        for (int a = 0; a < 5; a++) {
            for (int b = 0; b < 5; b++) M[a * 5 + b] = M[a * 5 + b] + row[a] * row[b];
            rhs[a] = rhs[a] + row[a] * lz;
        }
    }
    solve5(M, rhs, co);
    /* eig([2a b; b 2c]) */
    double a = co[0] * 2, b = co[1], c = co[2] * 2;
    double hd = (a - c) / 2.0, mid = (a + c) / 2.0, rad = sqrt(hd * hd + b * b);
    double l1 = mid - rad, l2 = mid + rad;
    double lam = l1; /* eig ascending: V(:,1) (estCurvatures.m:13-14) */
    (void)l2;
    /* eigenvector of lam: (b, lam - a) or (lam - c, b) -- take the better conditioned */
    double v0, v1;
    if (fabs(lam - a) >= fabs(lam - c)) { v0 = b; v1 = lam - a; }
    else { v0 = lam - c; v1 = b; }
    double nn = sqrt(v0 * v0 + v1 * v1);
    if (nn == 0) { v0 = 1; v1 = 0; nn = 1; }
    v0 = v0 / nn; v1 = v1 / nn;
    for (int k = 0; k < 3; k++) dir[k] = xx[k] * v0 + y[k] * v1;
}

/* fitCylinderWPts3(Pts3, cylRadius): P is n x 3 (row = point). returns 0 ok, 5 too few points */
static int fit_cylinder_mode(const double *P, int n, double R, double tolx, double tolf, int maxiter, int maxfun,
                             int mode, double *cyl0, double *cyl, double *fvals, int *iters, int *evals);

ORC_API int orc_fit_cylinder(const double *P, int n, double R, double tolx, double tolf, int maxiter,
                             int maxfun, double *cyl0, double *cyl, double *fvals, int *iters, int *evals)
{
    return fit_cylinder_mode(P, n, R, tolx, tolf, maxiter, maxfun, 0, cyl0, cyl, fvals, iters, evals);
}

/* mode 1 = Levenberg-Marquardt (the build's fast mode) */
ORC_API int orc_fit_cylinder_mode(const double *P, int n, double R, double tolx, double tolf, int maxiter,
                                  int maxfun, int mode, double *cyl0, double *cyl, double *fvals, int *iters, int *evals)
{
    return fit_cylinder_mode(P, n, R, tolx, tolf, maxiter, maxfun, mode, cyl0, cyl, fvals, iters, evals);
}

/* initial cylinder of fitCylinderWPts3.m:7-36 and the objective there */
static void fit_init(const double *P, int n, double R, double *cyl0, double *f0)
{
    double *tmp = (double *)malloc((size_t)(n + 1) * sizeof(double));
    double ctr[3];
    for (int c = 0; c < 3; c++) {
        for (int i = 0; i < n; i++) tmp[i] = P[3 * i + c];
        ctr[c] = sum64(tmp, n) / n;
    }
    double Cv[9];
    for (int a = 0; a < 3; a++)
        for (int b = a; b < 3; b++) {
            for (int i = 0; i < n; i++) tmp[i] = (P[3 * i + a] - ctr[a]) * (P[3 * i + b] - ctr[b]);
            Cv[a * 3 + b] = Cv[b * 3 + a] = sum64(tmp, n) / (n - 1);
        }
    double w[3], V[9];
    eig3(Cv, w, V);
    double rdir[3] = {V[0], V[3], V[6]}; /* pca coeff(:,3): least variance */
    if (rdir[2] < 0) { rdir[0] = -rdir[0]; rdir[1] = -rdir[1]; rdir[2] = -rdir[2]; }
    double p2[3] = {ctr[0] + rdir[0], ctr[1] + rdir[1], ctr[2] + rdir[2]};
    dist_to_line(P, n, ctr, p2, tmp);
    int im = 0;
    for (int i = 1; i < n; i++)
        if (tmp[i] < tmp[im]) im = i;
    double e0 = ctr[0] - P[3 * im], e1 = ctr[1] - P[3 * im + 1], e2 = ctr[2] - P[3 * im + 2];
    double d2s = sqrt((e0 * e0 + e1 * e1) + e2 * e2);
    double dir0[3];
    est_curv_dir(P, n, im, dir0);
    for (int c = 0; c < 3; c++) {
        cyl0[c] = ctr[c] + rdir[c] * (R - d2s);
        cyl0[3 + c] = dir0[c];
    }
    *f0 = cyl_objective(cyl0, P, n, R, tmp);
    free(tmp);
}

static int fit_cylinder_mode(const double *P, int n, double R, double tolx, double tolf, int maxiter, int maxfun,
                             int mode, double *cyl0, double *cyl, double *fvals, int *iters, int *evals)
{
    if (n < 3) return 5;
    fit_init(P, n, R, cyl0, &fvals[0]);
    if (mode == 1) lm6(cyl0, fvals[0], P, n, R, tolx, tolf, maxiter, cyl, &fvals[1], iters, evals);
    else nelder_mead6(cyl0, P, n, R, tolx, tolf, maxiter, maxfun, cyl, &fvals[1], iters, evals);
    return 0;
}

/* ---- BUILD-DEFINED (no counterpart in the reference; BASELINE config 5 / SURVEY 7.8): RANSAC around the fit ----------
 * H hypotheses per frame.  Hypothesis 0 uses all points; hypothesis h > 0 keeps point k with probability S/n, decided by
 * a counter-based hash of (seed, frame, h, k) -- no sequential generator state, so every lane / thread decides for its
 * own points.  Each hypothesis: `hyp_iters` LM iterations from the all-points initial cylinder on its subset, then the
 * inlier count | dist(point, axis) - R | < tau over ALL points.  The first hypothesis with the largest count wins;
 * the final fit (mode: 0 Nelder-Mead, 1 LM) runs on its inliers from its parameters.  Subsets / inlier sets of fewer
 * than 6 points are skipped / replaced by all points. */
static uint64_t ransac_hash(uint64_t seed, uint64_t frame, uint64_t h, uint64_t k)
{
    uint64_t z = (seed ^ (frame * 0xD1B54A32D192ED03ULL) ^ (h << 32) ^ k) + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

ORC_API int orc_fit_cylinder_ransac(const double *P, int n, double R, int H, int S, double tau, uint64_t seed, uint64_t frame,
                                    int hyp_iters, double tolx, double tolf, int maxiter, int maxfun, int mode, double *cyl0,
                                    double *cyl, double *fvals, int *iters, int *evals, int *n_inl, uint8_t *mask)
{
    if (n < 3) return 5;
    double *Q = (double *)malloc((size_t)(n + 1) * 3 * sizeof(double)), *d = (double *)malloc((size_t)(n + 1) * sizeof(double));
    fit_init(P, n, R, cyl0, &fvals[0]);
    const double q = (double)S / (double)n;
    int best_cnt = -1;
    double best_x[6] = {0, 0, 0, 0, 0, 0};
    for (int h = 0; h < H; h++) {
        int nq = 0;
        for (int k = 0; k < n; k++) {
            int take = 1;
            if (h > 0) take = ((double)(ransac_hash(seed, frame, (uint64_t)h, (uint64_t)k) >> 11) * 0x1.0p-53) < q;
            if (take) { Q[3 * nq] = P[3 * k]; Q[3 * nq + 1] = P[3 * k + 1]; Q[3 * nq + 2] = P[3 * k + 2]; nq++; }
        }
        if (nq < 6) continue;
        double xh[6], fh, fq0 = cyl_objective(cyl0, Q, nq, R, d);
        int it, ev;
        lm6(cyl0, fq0, Q, nq, R, tolx, tolf, hyp_iters, xh, &fh, &it, &ev);
        double p2[3] = {xh[0] + xh[3], xh[1] + xh[4], xh[2] + xh[5]};
        dist_to_line(P, n, xh, p2, d);
        int c = 0;
        for (int k = 0; k < n; k++) c += fabs(d[k] - R) < tau;
        if (c > best_cnt) { best_cnt = c; memcpy(best_x, xh, sizeof best_x); }
    }
    if (best_cnt < 0) { best_cnt = 0; memcpy(best_x, cyl0, sizeof best_x); }   /* no hypothesis had 6 points */
    int nq = 0;
    {
        double p2[3] = {best_x[0] + best_x[3], best_x[1] + best_x[4], best_x[2] + best_x[5]};
        dist_to_line(P, n, best_x, p2, d);
        for (int k = 0; k < n; k++) {
            mask[k] = fabs(d[k] - R) < tau;
            if (mask[k]) { Q[3 * nq] = P[3 * k]; Q[3 * nq + 1] = P[3 * k + 1]; Q[3 * nq + 2] = P[3 * k + 2]; nq++; }
        }
    }
    *n_inl = nq;
    if (nq < 6) {   /* too few inliers to fit: all points, and say so in the mask */
        memcpy(Q, P, (size_t)n * 3 * sizeof(double));
        nq = n;
        for (int k = 0; k < n; k++) mask[k] = 1;
    }
    double fs = cyl_objective(best_x, Q, nq, R, d);
    if (mode == 1) lm6(best_x, fs, Q, nq, R, tolx, tolf, maxiter, cyl, &fvals[1], iters, evals);
    else nelder_mead6(best_x, Q, nq, R, tolx, tolf, maxiter, maxfun, cyl, &fvals[1], iters, evals);
    /* applyCylParamsPrior / cylParams2T are applied by the caller on the points of the final fit (Q) */
    free(Q); free(d);
    return 0;
}

/* applyCylParamsPrior.m */
ORC_API void orc_apply_prior(double *cyl, const double *P, int n)
{
    double o[3] = {cyl[0], cyl[1], cyl[2]}, d[3] = {cyl[3], cyl[4], cyl[5]};
    if (d[1] < 0) { d[0] = -d[0]; d[1] = -d[1]; d[2] = -d[2]; }
    double ymin = P[1];
    for (int i = 1; i < n; i++)
        if (P[3 * i + 1] < ymin) ymin = P[3 * i + 1];
    double t = 0;
    if (!(fabs(d[1]) < DBL_EPSILON)) t = (ymin - o[1]) / d[1];
    for (int c = 0; c < 3; c++) { cyl[c] = o[c] + t * d[c]; cyl[3 + c] = d[c]; }
}

/* cylParams2T.m -> row-major 4x4 */
ORC_API void orc_cyl2T(const double *cyl, double *T)
{
    double y[3] = {cyl[3], cyl[4], cyl[5]};
    double ny = sqrt((y[0] * y[0] + y[1] * y[1]) + y[2] * y[2]);
    for (int c = 0; c < 3; c++) y[c] = y[c] / ny;
    double z[3] = {0 * y[2] - 0 * y[1], 0 * y[0] - 1 * y[2], 1 * y[1] - 0 * y[0]}; /* cross([1 0 0], y) */
    double nz = sqrt((z[0] * z[0] + z[1] * z[1]) + z[2] * z[2]);
    for (int c = 0; c < 3; c++) z[c] = z[c] / nz;
    double x[3] = {y[1] * z[2] - y[2] * z[1], y[2] * z[0] - y[0] * z[2], y[0] * z[1] - y[1] * z[0]};
    double nx = sqrt((x[0] * x[0] + x[1] * x[1]) + x[2] * x[2]);
    for (int c = 0; c < 3; c++) x[c] = x[c] / nx;
    for (int r = 0; r < 3; r++) {
        T[r * 4 + 0] = x[r]; T[r * 4 + 1] = y[r]; T[r * 4 + 2] = z[r]; T[r * 4 + 3] = cyl[r];
    }
    T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1;
}

/* fitSingleCylinder.m:5-25.  selector: 0 = chooseIdx(3, th) [live], 1 = triangulateWithThreshold(th),
 * 2 = findGridCorrespondences.  outputs: pts3 (m x 3), cyl (2 x 6, after applyCylParamsPrior), T (4x4),
 * fvals[2], meanError.  returns status (0 ok, 5 too few points). *m_out = number of points. */
ORC_API int orc_fit_single_cylinder(const double *gp1, int n1, const double *gp2, int n2, const double *K1,
                                    const double *K2, const double *T21, double R, int selector, double th,
                                    double *pts3, int *m_out, double *cyl, double *T, double *fvals,
                                    double *mean_err, int *iters, int *evals, int *fallback)
{
    int cap = (n1 > n2 ? n1 : n2) + 1;
    double *c1 = (double *)malloc((size_t)cap * 2 * sizeof(double)), *c2 = (double *)malloc((size_t)cap * 2 * sizeof(double));
    int m;
    *fallback = 0;
    if (selector == 0) m = orc_choose_idx(gp1, n1, gp2, n2, K1, K2, T21, 3, th, c1, c2, NULL, fallback);
    else if (selector == 1) m = orc_triangulate_with_threshold(gp1, n1, gp2, n2, K1, K2, T21, th, c1, c2, NULL, fallback);
    else m = orc_find_correspondences(gp1, n1, gp2, n2, c1, c2, NULL);
    *m_out = m;
    int st = 5;
    if (m > 0) {
        double *er = (double *)malloc((size_t)m * sizeof(double));
        orc_triangulate(c1, c2, m, K1, K2, T21, pts3, er);
        *mean_err = sum64(er, m) / m;
        free(er);
        st = orc_fit_cylinder(pts3, m, R, 1e-5, 1e-5, 100000, 100000, cyl, cyl + 6, fvals, iters, evals);
        if (st == 0) {
            orc_apply_prior(cyl, pts3, m);
            orc_apply_prior(cyl + 6, pts3, m);
            orc_cyl2T(cyl + 6, T);
        }
    }
    free(c1); free(c2);
    return st;
}


/* ------------------------------------------------------------------------------------------------------------
 * Row f-1: multi-frame AGV-pose fit, utils/fitCylinderWPts3sAngs.m:1-94 (+ getTAGVcyl.m, vec2T.m,
 * T2vec.m).  [ext] rotvec2mat3d / rotmat2vec3d / mrdivide / fminsearch restated -- PARITY UNPINNED (no MATLAB).
 * Reproduced quirk: cylParams{i} is the 2x6 matrix [cylParams0; cylParams] and applyCylParamsPrior indexes it
 * LINEARLY (applyCylParamsPrior.m:6-7), so "origin" = [M(1,1) M(2,1) M(1,2)] and "direction" = [M(2,2) M(1,3) M(2,3)].
 */
ORC_API void orc_get_TAGVcyl(double pan, double tilt, double *T) /* getTAGVcyl.m, row-major 4x4 */
{
    double cp = cos(pan), sp = sin(pan), ct = cos(-tilt), st = sin(-tilt);
    double TAP[16] = {cp, -sp, 0, 0, sp, cp, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    double TPT0[16] = {1, 0, 0, -143.1, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    double L = sqrt((-143.1 * -143.1 + 0.0 * 0.0) + 0.0 * 0.0);
    double mtr = -tan(tilt) * L;
    double T01[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, mtr, 0, 0, 0, 1};
    double T12[16] = {ct, 0, st, 0, 0, 1, 0, 0, -st, 0, ct, 0, 0, 0, 0, 1};
    double T2C[16] = {0, -1, 0, 321.1, -1, 0, 0, 0, 0, 0, -1, 110, 0, 0, 0, 1};
    const double *chain[4] = {TPT0, T01, T12, T2C};
    double acc[16], nxt[16];
    memcpy(acc, TAP, sizeof acc);
    for (int m = 0; m < 4; m++) {
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) {
                double s_ = 0.0;
                for (int k = 0; k < 4; k++) s_ = s_ + acc[r * 4 + k] * chain[m][k * 4 + c];
                nxt[r * 4 + c] = s_;
            }
        memcpy(acc, nxt, sizeof acc);
    }
    memcpy(T, acc, sizeof acc);
}

static void rotvec2mat(const double *v, double *R) /* rotvec2mat3d (premultiply) */
{
    double th = sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    if (th < 1e-6) { for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0; return; }
    double u[3] = {v[0] / th, v[1] / th, v[2] / th};
    double c = cos(th), s_ = sin(th), t = 1 - c;
    double K[9] = {0, -u[2], u[1], u[2], 0, -u[0], -u[1], u[0], 0};
    for (int r = 0; r < 3; r++)
        for (int q = 0; q < 3; q++) R[r * 3 + q] = (c * (r == q ? 1.0 : 0.0) + t * (u[r] * u[q])) + s_ * K[r * 3 + q];
}
static void mat2rotvec(const double *R, double *v) /* rotmat2vec3d (without its SVD re-orthogonalisation) */
{
    double t = (R[0] + R[4]) + R[8];
    double ca = (t - 1) / 2;
    if (ca > 1) ca = 1;
    if (ca < -1) ca = -1;
    double th = acos(ca);
    double r[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
    if (sin(th) >= 1e-4) {
        double vth = 1 / (2 * sin(th));
        for (int k = 0; k < 3; k++) v[k] = th * (r[k] * vth);
    } else if (t - 1 > 0) {
        for (int k = 0; k < 3; k++) v[k] = (.5 - (t - 3) / 12) * r[k];
    } else {
        int a = 0;
        if (R[4] > R[a * 4]) a = 1;
        if (R[8] > R[a * 4]) a = 2;
        int b = (a + 1) % 3, c = (a + 2) % 3;
        double s_ = sqrt(R[a * 4] - R[b * 4] - R[c * 4] + 1);
        double w[3];
        w[a] = s_ / 2;
        w[b] = (R[b * 3 + a] + R[a * 3 + b]) / (2 * s_);
        w[c] = (R[c * 3 + a] + R[a * 3 + c]) / (2 * s_);
        double nw = sqrt((w[0] * w[0] + w[1] * w[1]) + w[2] * w[2]);
        for (int k = 0; k < 3; k++) v[k] = th * w[k] / nw;
    }
}
ORC_API void orc_vec2T(const double *x, double *T) /* vec2T.m, row-major */
{
    double R[9];
    rotvec2mat(x, R);
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) T[r * 4 + c] = R[r * 3 + c]; T[r * 4 + 3] = x[3 + r]; }
    T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1;
}
ORC_API void orc_T2vec(const double *T, double *x) /* T2vec.m */
{
    double R[9];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R[r * 3 + c] = T[r * 4 + c];
    mat2rotvec(R, x);
    for (int r = 0; r < 3; r++) x[3 + r] = T[r * 4 + 3];
}

typedef struct { const double *P; const int *cnt; int F, cap; const double *TAGV; double R; double *tmp; } multi_ctx;
/* dist() of fitCylinderWPts3sAngs.m:82-94 */
static double multi_obj(const double *x, void *c_)
{
    multi_ctx *c = (multi_ctx *)c_;
    double T[16];
    orc_vec2T(x, T);
    double v = 0;
    for (int i = 0; i < c->F; i++) {
        const double *A = c->TAGV + 16 * (size_t)i;
        int n = c->cnt[i];
        if (n <= 0) { v = v + 0.0; continue; }
        double org[3], dy[3];
        for (int r = 0; r < 3; r++) {
            dy[r] = ((T[r * 4] * A[1] + T[r * 4 + 1] * A[5]) + T[r * 4 + 2] * A[9]) + T[r * 4 + 3] * A[13];
            org[r] = ((T[r * 4] * A[3] + T[r * 4 + 1] * A[7]) + T[r * 4 + 2] * A[11]) + T[r * 4 + 3] * A[15];
        }
        double p2[3] = {org[0] + dy[0], org[1] + dy[1], org[2] + dy[2]};
        dist_to_line(c->P + (size_t)i * c->cap * 3, n, org, p2, c->tmp);
        for (int k = 0; k < n; k++) { double w = c->tmp[k] - c->R; c->tmp[k] = w * w; }
        v = v + sum64(c->tmp, n) / n;
    }
    return v;
}

ORC_API double orc_multi_objective(const double *x, const double *P, const int *cnt, int F, int cap, const double *TAGV, double R)
{
    multi_ctx c = {P, cnt, F, cap, TAGV, R, (double *)malloc((size_t)(cap + 1) * sizeof(double))};
    double v = multi_obj(x, &c);
    free(c.tmp);
    return v;
}

static void cross3v(const double *a, const double *b, double *o)
{
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}

/* initial AGV pose T0 of fitCylinderWPts3sAngs.m:40-69 from the (quirkily indexed) per-frame fits */
ORC_API void orc_multi_init(const double *cyl_raw /* F x 2 x 6 */, const double *P, const int *cnt, int cap,
                            const double *TAGV, double *x0)
{
    double cp[2][6];
    for (int i = 0; i < 2; i++) {
        const double *M = cyl_raw + 12 * (size_t)i; /* M(r,c) = M[r*6 + c]; linear index k (1-based): row (k-1)%2, col (k-1)/2 */
        double lin[6];
        for (int k = 0; k < 6; k++) lin[k] = M[(k % 2) * 6 + (k / 2)];
        memcpy(cp[i], lin, sizeof lin);
        orc_apply_prior(cp[i], P + (size_t)i * cap * 3, cnt[i]);
    }
    const double *A1 = TAGV, *A2 = TAGV + 16;
    double p1[3] = {A1[3], A1[7], A1[11]}, p2[3] = {A2[3], A2[7], A2[11]};
    double y1[3] = {A1[1], A1[5], A1[9]};
    double d12[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
    double nd[3]; cross3v(y1, d12, nd);
    double nn = sqrt((nd[0] * nd[0] + nd[1] * nd[1]) + nd[2] * nd[2]);
    for (int k = 0; k < 3; k++) nd[k] = nd[k] / nn;
    double ed12[3] = {cp[1][0] - cp[0][0], cp[1][1] - cp[0][1], cp[1][2] - cp[0][2]};
    double dir1[3] = {cp[0][3], cp[0][4], cp[0][5]};
    double en[3]; cross3v(dir1, ed12, en);
    double ne = sqrt((en[0] * en[0] + en[1] * en[1]) + en[2] * en[2]);
    for (int k = 0; k < 3; k++) en[k] = en[k] / ne;
    double c1[3], c2[3];
    cross3v(dir1, en, c1);
    cross3v(y1, nd, c2);
    /* R = A / B with A = [dir1 en c1], B = [y1 nd c2] (columns): solve R B = A  <=>  B' R' = A' by Gaussian elimination */
    double Bt[9], At[9];
    const double *Ac[3] = {dir1, en, c1}, *Bc[3] = {y1, nd, c2};
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { Bt[r * 3 + c] = Bc[r][c]; At[r * 3 + c] = Ac[r][c]; }
    for (int c = 0; c < 3; c++) {
        int pv = c;
        for (int r = c + 1; r < 3; r++) if (fabs(Bt[r * 3 + c]) > fabs(Bt[pv * 3 + c])) pv = r;
        if (pv != c) for (int k = 0; k < 3; k++) {
            double t_ = Bt[c * 3 + k]; Bt[c * 3 + k] = Bt[pv * 3 + k]; Bt[pv * 3 + k] = t_;
            t_ = At[c * 3 + k]; At[c * 3 + k] = At[pv * 3 + k]; At[pv * 3 + k] = t_;
        }
        for (int r = c + 1; r < 3; r++) {
            double f = Bt[r * 3 + c] / Bt[c * 3 + c];
            for (int k = c; k < 3; k++) Bt[r * 3 + k] = Bt[r * 3 + k] - f * Bt[c * 3 + k];
            for (int k = 0; k < 3; k++) At[r * 3 + k] = At[r * 3 + k] - f * At[c * 3 + k];
        }
    }
    double Rt[9];
    for (int k = 0; k < 3; k++)
        for (int r = 2; r >= 0; r--) {
            double s_ = At[r * 3 + k];
            for (int q = r + 1; q < 3; q++) s_ = s_ - Bt[r * 3 + q] * Rt[q * 3 + k];
            Rt[r * 3 + k] = s_ / Bt[r * 3 + r];
        }
    double T0[16];
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) T0[r * 4 + c] = Rt[c * 3 + r];
    }
    for (int r = 0; r < 3; r++) {
        double s_ = (T0[r * 4] * p1[0] + T0[r * 4 + 1] * p1[1]) + T0[r * 4 + 2] * p1[2];
        T0[r * 4 + 3] = cp[0][r] - s_;
    }
    T0[12] = 0; T0[13] = 0; T0[14] = 0; T0[15] = 1;
    orc_T2vec(T0, x0);
}

/* fitCylinderWPts3sAngs: P (F x cap x 3), cnt[F], TAGV (F x 16), cyl_raw (F x 2 x 6 from fitCylinderWPts3 per frame) */
ORC_API void orc_multi_fit(const double *P, const int *cnt, int F, int cap, const double *TAGV, const double *cyl_raw, double R,
                           double *x0, double *x, double *T, double *fvals, int *iters, int *evals)
{
    orc_multi_init(cyl_raw, P, cnt, cap, TAGV, x0);
    multi_ctx c = {P, cnt, F, cap, TAGV, R, (double *)malloc((size_t)(cap + 1) * sizeof(double))};
    fvals[0] = multi_obj(x0, &c);
    nelder_mead6_fn(x0, multi_obj, &c, 1e-5, 1e-5, 100000, 100000, x, &fvals[1], iters, evals);
    orc_vec2T(x, T);
    free(c.tmp);
}
