// Geometric half of the hot path for gfx950: one wavefront (64 lanes) per stereo frame.
//   reference: utils/fitSingleCylinder.m:5-25, chooseIdx.m:19-104, findGridCorrespondences.m,
//   triangulateWithThreshold.m:16-43, fitCylinderWPts3.m, getDistPts3ToLine.m, estCurvatures.m,
//   fitplane.m, applyCylParamsPrior.m, cylParams2T.m; [ext] MATLAB triangulate / pca / knnsearch /
//   fminsearch restated as in SURVEY.md appendix B.
//
// Nothing here is a dense contraction (per-point 4x4 SVDs, per-frame reductions), so this is plain
// f64 VALU work, no MFMA.  Every reduction over the points of a frame is the fixed tree "lane l
// adds points l, l+64, ... in order, then xor-butterfly 32..1" and the build uses
// -ffp-contract=off, so results are bit-identical to the CPU oracle's.
//
// k_select_triangulate : index join through dense (col,row)->slot tables, per-point DLT
//                        triangulation (one lane per point), the 3x3 patch scan of chooseIdx with
//                        per-point errors reused across patches, compaction in the reference's
//                        output order (containers.Map string-key order).
// k_fit_cylinder       : mean / PCA / nearest-to-axis point / 20-NN quadric at that one point /
//                        Nelder-Mead (MATLAB fminsearch order) / applyCylParamsPrior / cylParams2T.
#include "cpe_internal.h"
#include <float.h>

namespace {

constexpr int MAXP = CPE_MAXP;
// the per-point LDS arrays of the one-wavefront-per-frame kernels (dynamic: they exceed the static 64 KB limit)
extern __shared__ double fit_dyn[];
constexpr size_t SEL_LDS_BYTES = (size_t)MAXP * (3 * 8 + 8 + 8 + 4 + 4 + 1);
constexpr size_t FIT_LDS_BYTES = (size_t)MAXP * 4 * 8;
constexpr size_t RANSAC_LDS_BYTES = (size_t)MAXP * 7 * 8;
constexpr int TBL = CPE_FIT_TABLE_DIM;  // dense (col,row) table is TBL x TBL per image
constexpr int MAXU = TBL;

__device__ __forceinline__ double wave_sum(double p)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) p = p + __shfl_xor(p, off, 64);
    return p;
}
__device__ __forceinline__ int wave_min_i(int v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off, 64);
    return v;
}

// P = K * T(1:3,:)
__device__ void make_P(const double *K, const double *T, double *P)
{
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) {
            double s = 0.0;
            for (int k = 0; k < 3; k++) s = s + K[r * 3 + k] * T[k * 4 + c];
            P[r * 4 + c] = s;
        }
}

// one-sided Jacobi SVD of a 4x4, returns the right singular vector of the smallest singular value
__device__ void svd4_null(double *U, double *x)
{
    double V[16];
#pragma unroll
    for (int i = 0; i < 16; i++) V[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; sweep++) {
        int rotated = 0;
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int q = p + 1; q < 4; q++) {
                double al = 0.0, be = 0.0, ga = 0.0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    al = al + U[k * 4 + p] * U[k * 4 + p];
                    be = be + U[k * 4 + q] * U[k * 4 + q];
                    ga = ga + U[k * 4 + p] * U[k * 4 + q];
                }
                if (!(fabs(ga) <= 1e-15 * sqrt(al * be))) {
                    rotated = 1;
                    double zeta = (be - al) / (2.0 * ga);
                    double t = 1.0 / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    if (zeta < 0) t = -t;
                    double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        double up = U[k * 4 + p], uq = U[k * 4 + q];
                        U[k * 4 + p] = c * up - s * uq;
                        U[k * 4 + q] = s * up + c * uq;
                        double vp = V[k * 4 + p], vq = V[k * 4 + q];
                        V[k * 4 + p] = c * vp - s * vq;
                        V[k * 4 + q] = s * vp + c * vq;
                    }
                }
            }
        if (!rotated) break;
    }
    double best = 0;
    x[0] = x[1] = x[2] = x[3] = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        double nn = 0.0;
#pragma unroll
        for (int k = 0; k < 4; k++) nn = nn + U[k * 4 + j] * U[k * 4 + j];
        if (j == 0 || nn < best) {
            best = nn;
#pragma unroll
            for (int k = 0; k < 4; k++) x[k] = V[k * 4 + j];
        }
    }
}

__device__ __forceinline__ void project(const double *P, const double *X, double &u, double &v)
{
    double a = ((P[0] * X[0] + P[1] * X[1]) + P[2] * X[2]) + P[3];
    double b = ((P[4] * X[0] + P[5] * X[1]) + P[6] * X[2]) + P[7];
    double c = ((P[8] * X[0] + P[9] * X[1]) + P[10] * X[2]) + P[11];
    u = a / c;
    v = b / c;
}

__device__ void triangulate_one(const double *P1, const double *P2, double u1, double v1, double u2,
                                double v2, double *X, double &err)
{
    double A[16], x[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        A[0 * 4 + c] = u1 * P1[8 + c] - P1[c];
        A[1 * 4 + c] = v1 * P1[8 + c] - P1[4 + c];
        A[2 * 4 + c] = u2 * P2[8 + c] - P2[c];
        A[3 * 4 + c] = v2 * P2[8 + c] - P2[4 + c];
    }
    svd4_null(A, x);
    X[0] = x[0] / x[3];
    X[1] = x[1] / x[3];
    X[2] = x[2] / x[3];
    double u, v, dx, dy;
    project(P1, X, u, v);
    dx = u1 - u;
    dy = v1 - v;
    double e1 = sqrt(dx * dx + dy * dy);
    project(P2, X, u, v);
    dx = u2 - u;
    dy = v2 - v;
    double e2 = sqrt(dx * dx + dy * dy);
    err = (e1 + e2) / 2.0;
}

// order of the char keys sprintf('%d_%d', c, r) in a containers.Map (chooseIdx.m:69,89):
// strings compare by char code, '-' < digits < '_'; encoded base 13 (pad 0) into one integer.
__device__ unsigned long long string_key(int c, int r)
{
    int codes[12];
    int n = 0;
    int vals[2] = {c, r};
    for (int part = 0; part < 2; part++) {
        int v = vals[part];
        if (v < 0) { codes[n++] = 1; v = -v; }
        int dig[5], nd = 0;
        do { dig[nd++] = v % 10; v /= 10; } while (v > 0 && nd < 5);
        for (int k = nd - 1; k >= 0; k--) codes[n++] = 2 + dig[k];
        if (part == 0) codes[n++] = 12;
    }
    unsigned long long key = 0;
    for (int k = 0; k < 11; k++) key = key * 13ull + (unsigned long long)(k < n ? codes[k] : 0);
    return key;
}

__global__ __launch_bounds__(64) void k_select_triangulate(
    const double *__restrict__ xy1, const int *__restrict__ id1, const int *__restrict__ cnt1,
    const double *__restrict__ xy2, const int *__restrict__ id2, const int *__restrict__ cnt2,
    const double *__restrict__ K1, const double *__restrict__ K2, const double *__restrict__ T21,
    int selector, int patch, double th, int *__restrict__ tables /* n * 2 * TBL*TBL */,
    double *__restrict__ o_p1, double *__restrict__ o_p2, int *__restrict__ o_idx,
    double *__restrict__ o_X, double *__restrict__ o_err, int *__restrict__ o_m,
    double *__restrict__ o_mean_err, int *__restrict__ o_flags)
{
    __builtin_amdgcn_s_setprio(3);   // one wavefront per frame beside wide kernels (see k_fit_cylinder)
    // per-point arrays: dynamic LDS (SEL_LDS_BYTES; more than the 64 KB a kernel may declare statically)
    double *sX = fit_dyn;                                                                   // [MAXP * 3]
    double *sErr = sX + MAXP * 3;                                                           // [MAXP]
    unsigned long long *sKey = reinterpret_cast<unsigned long long *>(sErr + MAXP);         // [MAXP]
    int *sJ = reinterpret_cast<int *>(sKey + MAXP);                                         // [MAXP] matching slot in image 2 (or -1)
    int *sOrder = sJ + MAXP;                                                                // [MAXP] output position -> gp1 slot
    unsigned char *sSel = reinterpret_cast<unsigned char *>(sOrder + MAXP);                 // [MAXP]
    __shared__ int sUx[MAXU], sUy[MAXU];
    __shared__ unsigned char sFx[MAXU], sFy[MAXU];
    const int f = blockIdx.x, lane = threadIdx.x;
    const double *a1 = xy1 + (size_t)f * MAXP * 2, *a2 = xy2 + (size_t)f * MAXP * 2;
    const int *i1 = id1 + (size_t)f * MAXP * 2, *i2 = id2 + (size_t)f * MAXP * 2;
    int n1 = min(max(cnt1[f], 0), MAXP), n2 = min(max(cnt2[f], 0), MAXP);
    int *t1 = tables + (size_t)f * 2 * TBL * TBL, *t2 = t1 + TBL * TBL;
    int flags = 0;

    // index range over both tables
    int cmin = INT_MAX, cmax = INT_MIN, rmin = INT_MAX, rmax = INT_MIN;
    for (int i = lane; i < n1; i += 64) {
        int c = i1[2 * i], r = i1[2 * i + 1];
        cmin = min(cmin, c); cmax = max(cmax, c); rmin = min(rmin, r); rmax = max(rmax, r);
    }
    for (int i = lane; i < n2; i += 64) {
        int c = i2[2 * i], r = i2[2 * i + 1];
        cmin = min(cmin, c); cmax = max(cmax, c); rmin = min(rmin, r); rmax = max(rmax, r);
    }
    cmin = wave_min_i(cmin); cmax = wave_max_i(cmax); rmin = wave_min_i(rmin); rmax = wave_max_i(rmax);
    int m = 0;
    bool ok = n1 > 0 && n2 > 0;
    if (ok && ((long long)cmax - cmin >= TBL || (long long)rmax - rmin >= TBL || cmin < -9999 || cmax > 9999 ||
               rmin < -9999 || rmax > 9999)) {
        ok = false;
        flags |= CPE_FIT_FLAG_OVERFLOW;
    }
    if (ok) {
        const int tw = cmax - cmin + 1, thh = rmax - rmin + 1;
        for (int i = lane; i < tw * thh; i += 64) { t1[i] = INT_MAX; t2[i] = INT_MAX; }
        __syncthreads();
        for (int i = lane; i < n1; i += 64) atomicMin(&t1[(i1[2 * i + 1] - rmin) * tw + (i1[2 * i] - cmin)], i);
        for (int i = lane; i < n2; i += 64) atomicMin(&t2[(i2[2 * i + 1] - rmin) * tw + (i2[2 * i] - cmin)], i);
        __syncthreads();
#define TBL1(c, r) __hip_atomic_load(&t1[((r) - rmin) * tw + ((c) - cmin)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define TBL2(c, r) __hip_atomic_load(&t2[((r) - rmin) * tw + ((c) - cmin)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
        double P1[12], P2[12];
        {
            const double I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
            double k1[9], k2[9], tt[16];
            for (int k = 0; k < 9; k++) { k1[k] = K1[k]; k2[k] = K2[k]; }
            for (int k = 0; k < 16; k++) tt[k] = T21[k];
            make_P(k1, I4, P1);
            make_P(k2, tt, P2);
        }
        // join (first occurrence on both sides) + per-point triangulation
        int njoin = 0;
        for (int i = lane; i < n1; i += 64) {
            int c = i1[2 * i], r = i1[2 * i + 1];
            int j = TBL2(c, r);
            int jj = (j != INT_MAX) ? j : -1;
            sJ[i] = jj;
            sSel[i] = 0;
            if (jj >= 0) {
                double X[3], e;
                triangulate_one(P1, P2, a1[2 * i], a1[2 * i + 1], a2[2 * jj], a2[2 * jj + 1], X, e);
                sX[3 * i] = X[0]; sX[3 * i + 1] = X[1]; sX[3 * i + 2] = X[2];
                sErr[i] = e;
                njoin++;
            }
        }
        njoin = wave_sum_i(njoin);
        __syncthreads();

        bool string_order = false;
        if (selector == 0 && njoin > 0) {
            // unique sorted col / row values of image 1 (chooseIdx.m:22-23)
            int nx = 0, ny = 0;
            for (int i = lane; i < MAXU; i += 64) { sFx[i] = 0; sFy[i] = 0; }
            __syncthreads();
            for (int i = lane; i < n1; i += 64) { sFx[i1[2 * i] - cmin] = 1; sFy[i1[2 * i + 1] - rmin] = 1; }
            __syncthreads();
            if (lane == 0) {
                for (int c = 0; c < tw; c++)
                    if (sFx[c]) sUx[nx++] = c + cmin;
                for (int r = 0; r < thh; r++)
                    if (sFy[r]) sUy[ny++] = r + rmin;
            }
            nx = __shfl(nx, 0, 64);
            ny = __shfl(ny, 0, 64);
            __syncthreads();
            const int px = nx - patch + 1, py = ny - patch + 1;
            int nsel = 0;
            if (px > 0 && py > 0) {
                for (int pidx = lane; pidx < px * py; pidx += 64) {
                    int ix = pidx / py, iy = pidx - ix * py;
                    bool all = true;
                    double s = 0.0;
                    for (int a = 0; a < patch && all; a++)
                        for (int b = 0; b < patch; b++) {
                            int c = sUx[ix + a], r = sUy[iy + b];
                            int s1 = TBL1(c, r), s2 = TBL2(c, r);
                            if (s1 == INT_MAX || s2 == INT_MAX) { all = false; break; }
                            s = s + sErr[s1];
                        }
                    if (all && (s / (double)(patch * patch) < th)) {
                        for (int a = 0; a < patch; a++)
                            for (int b = 0; b < patch; b++) sSel[TBL1(sUx[ix + a], sUy[iy + b])] = 1;
                    }
                }
            }
            __syncthreads();
            for (int i = lane; i < n1; i += 64) nsel += sSel[i];
            nsel = wave_sum_i(nsel);
            if (nsel > 0) string_order = true;
            else flags |= CPE_FIT_FLAG_FALLBACK;
        } else if (selector == 1 && njoin > 0) {
            int nsel = 0;
            for (int i = lane; i < n1; i += 64) {
                unsigned char sel = (sJ[i] >= 0 && sErr[i] < th) ? 1 : 0;
                sSel[i] = sel;
                nsel += sel;
            }
            nsel = wave_sum_i(nsel);
            if (nsel == 0) flags |= CPE_FIT_FLAG_FALLBACK;
        } else {
            flags |= (selector == 2) ? 0 : CPE_FIT_FLAG_FALLBACK;
        }
        __syncthreads();
        if ((flags & CPE_FIT_FLAG_FALLBACK) || selector == 2)
            for (int i = lane; i < n1; i += 64) sSel[i] = sJ[i] >= 0 ? 1 : 0;
        __syncthreads();

        // output order
        if (string_order) {
            for (int i = lane; i < n1; i += 64)
                sKey[i] = sSel[i] ? string_key(i1[2 * i], i1[2 * i + 1]) : ~0ull;
            __syncthreads();
            for (int i = lane; i < n1; i += 64) {
                if (!sSel[i]) continue;
                unsigned long long ki = sKey[i];
                int rank = 0;
                for (int j = 0; j < n1; j++) rank += (sKey[j] < ki) ? 1 : 0;
                sOrder[rank] = i;
            }
            for (int i = lane; i < n1; i += 64) m += sSel[i];
            m = wave_sum_i(m);
        } else {
            int base = 0;
            for (int i0 = 0; i0 < n1; i0 += 64) {
                int i = i0 + lane;
                bool sel = i < n1 && sSel[i];
                unsigned long long bal = __ballot(sel);
                if (sel) sOrder[base + __popcll(bal & ((1ull << lane) - 1ull))] = i;
                base += __popcll(bal);
            }
            m = base;
        }
        __syncthreads();
        double es = 0.0;
        for (int k = lane; k < m; k += 64) {
            int i = sOrder[k], j = sJ[i];
            size_t o = (size_t)f * MAXP + k;
            o_p1[2 * o] = a1[2 * i]; o_p1[2 * o + 1] = a1[2 * i + 1];
            o_p2[2 * o] = a2[2 * j]; o_p2[2 * o + 1] = a2[2 * j + 1];
            o_idx[2 * o] = i1[2 * i]; o_idx[2 * o + 1] = i1[2 * i + 1];
            o_X[3 * o] = sX[3 * i]; o_X[3 * o + 1] = sX[3 * i + 1]; o_X[3 * o + 2] = sX[3 * i + 2];
            o_err[o] = sErr[i];
            es = es + sErr[i];
        }
        es = wave_sum(es);
        if (lane == 0) o_mean_err[f] = m > 0 ? es / (double)m : 0.0;
#undef TBL1
#undef TBL2
    } else if (lane == 0) {
        o_mean_err[f] = 0.0;
    }
    if (lane == 0) { o_m[f] = m; o_flags[f] = flags; }
}

// ------------------------------------------------------------------------------------------ fit
struct Pts {
    const double *p;  // LDS, 3 per point
    int n;
};

// getDistPts3ToLine.m for one point
__device__ __forceinline__ double dist_pt_line(const double *x, const double *p1, const double *v, double nv2)
{
    double al = (((x[0] - p1[0]) * v[0] + (x[1] - p1[1]) * v[1]) + (x[2] - p1[2]) * v[2]) / nv2;
    double e0 = x[0] - (p1[0] + v[0] * al), e1 = x[1] - (p1[1] + v[1] * al), e2 = x[2] - (p1[2] + v[2] * al);
    return sqrt((e0 * e0 + e1 * e1) + e2 * e2);
}

// dist() of fitCylinderWPts3.m:44-49
__device__ double cyl_objective(const double *x, const Pts &P, double R, int lane)
{
    double p2[3] = {x[0] + x[3], x[1] + x[4], x[2] + x[5]};
    double v[3] = {p2[0] - x[0], p2[1] - x[1], p2[2] - x[2]};
    double nv2 = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2];
    double acc = 0.0;
    for (int k = lane; k < P.n; k += 64) {
        double d = dist_pt_line(P.p + 3 * k, x, v, nv2);
        double w = d - R;
        acc = acc + w * w;
    }
    return wave_sum(acc);
}

__device__ void eig3(const double *Ain, double *w, double *V)
{
    double A[9];
#pragma unroll
    for (int i = 0; i < 9; i++) { A[i] = Ain[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 30; sweep++) {
        int rotated = 0;
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
            for (int q = p + 1; q < 3; q++) {
                double apq = A[p * 3 + q];
                if (!(fabs(apq) <= 1e-17 * (fabs(A[p * 3 + p]) + fabs(A[q * 3 + q])))) {
                    rotated = 1;
                    double theta = (A[q * 3 + q] - A[p * 3 + p]) / (2.0 * apq);
                    double t = 1.0 / (fabs(theta) + sqrt(theta * theta + 1.0));
                    if (theta < 0) t = -t;
                    double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        double akp = A[k * 3 + p], akq = A[k * 3 + q];
                        A[k * 3 + p] = c * akp - s * akq;
                        A[k * 3 + q] = s * akp + c * akq;
                    }
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        double apk = A[p * 3 + k], aqk = A[q * 3 + k];
                        A[p * 3 + k] = c * apk - s * aqk;
                        A[q * 3 + k] = s * apk + c * aqk;
                    }
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        double vkp = V[k * 3 + p], vkq = V[k * 3 + q];
                        V[k * 3 + p] = c * vkp - s * vkq;
                        V[k * 3 + q] = s * vkp + c * vkq;
                    }
                }
            }
        if (!rotated) break;
    }
    // ascending, stable bubble (as the oracle)
    double d[3] = {A[0], A[4], A[8]};
    double Vs[9];
#pragma unroll
    for (int i = 0; i < 9; i++) Vs[i] = V[i];
#define CSWAP(a, b)                                                                  \
    if (d[b] < d[a]) {                                                               \
        double t_ = d[a]; d[a] = d[b]; d[b] = t_;                                    \
        for (int k = 0; k < 3; k++) { double u_ = Vs[k * 3 + a]; Vs[k * 3 + a] = Vs[k * 3 + b]; Vs[k * 3 + b] = u_; } \
    }
    CSWAP(0, 1) CSWAP(1, 2) CSWAP(0, 1)
#undef CSWAP
#pragma unroll
    for (int i = 0; i < 3; i++) w[i] = d[i];
#pragma unroll
    for (int i = 0; i < 9; i++) V[i] = Vs[i];
}

__device__ void solve5(double *M, double *b, double *x)
{
    constexpr int N = 5;
    for (int c = 0; c < N; c++) {
        int pv = c;
        for (int r = c + 1; r < N; r++)
            if (fabs(M[r * N + c]) > fabs(M[pv * N + c])) pv = r;
        if (pv != c) {
            for (int k = 0; k < N; k++) { double t = M[c * N + k]; M[c * N + k] = M[pv * N + k]; M[pv * N + k] = t; }
            double t = b[c]; b[c] = b[pv]; b[pv] = t;
        }
        for (int r = c + 1; r < N; r++) {
            double fct = M[r * N + c] / M[c * N + c];
            for (int k = c; k < N; k++) M[r * N + k] = M[r * N + k] - fct * M[c * N + k];
            b[r] = b[r] - fct * b[c];
        }
    }
    for (int r = N - 1; r >= 0; r--) {
        double s = b[r];
        for (int k = r + 1; k < N; k++) s = s - M[r * N + k] * x[k];
        x[r] = s / M[r * N + r];
    }
}

__device__ __forceinline__ double eps_of(double x)  // MATLAB eps(x)
{
    x = fabs(x);
    if (x < DBL_MIN) return 4.9406564584124654e-324;
    int e;
    frexp(x, &e);
    return ldexp(1.0, e - 53);
}

// ---- the pieces of fitCylinderWPts3.m as device functions (one wavefront, points in LDS); k_fit_cylinder runs them on
// all points of a frame, k_fit_ransac (build-defined, BASELINE config 5) on subsets as well
// initial cylinder (fitCylinderWPts3.m:7-36): x0 = [origin, direction], f0 = objective at x0
__device__ void fit_init(const double *sP, int n, double R, int lane, double *sD, int *sNb, double *x0, double &f0_out)
{
    Pts P{sP, n};
    // ctr = mean(Pts3,2); covariance; rdir = pca 3rd axis with z > 0 (fitCylinderWPts3.m:7-19)
    double ctr[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        double acc = 0.0;
        for (int k = lane; k < n; k += 64) acc = acc + sP[3 * k + c];
        ctr[c] = wave_sum(acc) / (double)n;
    }
    double Cv[9];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = a; b < 3; b++) {
            double acc = 0.0;
            for (int k = lane; k < n; k += 64) acc = acc + (sP[3 * k + a] - ctr[a]) * (sP[3 * k + b] - ctr[b]);
            double v = wave_sum(acc) / (double)(n - 1);
            Cv[a * 3 + b] = v;
            Cv[b * 3 + a] = v;
        }
    double w3[3], V3[9];
    eig3(Cv, w3, V3);
    double rdir[3] = {V3[0], V3[3], V3[6]};
    if (rdir[2] < 0) { rdir[0] = -rdir[0]; rdir[1] = -rdir[1]; rdir[2] = -rdir[2]; }

    // i = argmin dist to line(ctr, ctr + rdir)  (first minimum)
    int im;
    {
        double p2[3] = {ctr[0] + rdir[0], ctr[1] + rdir[1], ctr[2] + rdir[2]};
        double v[3] = {p2[0] - ctr[0], p2[1] - ctr[1], p2[2] - ctr[2]};
        double nv2 = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2];
        double bd = DBL_MAX;
        int bi = INT_MAX;
        for (int k = lane; k < n; k += 64) {
            double d = dist_pt_line(sP + 3 * k, ctr, v, nv2);
            if (d < bd) { bd = d; bi = k; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            double od = __shfl_xor(bd, off, 64);
            int oi = __shfl_xor(bi, off, 64);
            if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
        }
        im = bi;
    }
    double e0 = ctr[0] - sP[3 * im], e1 = ctr[1] - sP[3 * im + 1], e2 = ctr[2] - sP[3 * im + 2];
    double d2s = sqrt((e0 * e0 + e1 * e1) + e2 * e2);

    // estCurvatures at point im only: 20-NN (ties by index), plane, local frame, quadric, 2x2 eig
    const int K = n < 20 ? n : 20;
    for (int k = lane; k < n; k += 64) {
        double a = sP[3 * k] - sP[3 * im], b = sP[3 * k + 1] - sP[3 * im + 1], c = sP[3 * k + 2] - sP[3 * im + 2];
        sD[k] = (a * a + b * b) + c * c;
    }
    __syncthreads();
    for (int r = 0; r < K; r++) {
        double bd = DBL_MAX;
        int bi = INT_MAX;
        for (int k = lane; k < n; k += 64) {
            double d = sD[k];
            if (d >= 0 && d < bd) { bd = d; bi = k; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            double od = __shfl_xor(bd, off, 64);
            int oi = __shfl_xor(bi, off, 64);
            if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
        }
        if (lane == 0) { sNb[r] = bi; sD[bi] = -1.0; }
        __syncthreads();
    }
    double dir0[3];
    {
        double mu[3] = {0, 0, 0};
        for (int k = 0; k < K; k++)
            for (int c = 0; c < 3; c++) mu[c] = mu[c] + sP[3 * sNb[k] + c];
        for (int c = 0; c < 3; c++) mu[c] = mu[c] / (double)K;
        double C2[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < K; k++) {
            double e[3] = {sP[3 * sNb[k]] - mu[0], sP[3 * sNb[k] + 1] - mu[1], sP[3 * sNb[k] + 2] - mu[2]};
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int b = 0; b < 3; b++) C2[a * 3 + b] = C2[a * 3 + b] + e[a] * e[b];
        }
#pragma unroll
        for (int a = 0; a < 9; a++) C2[a] = C2[a] / (double)(K - 1);
        double w[3], V[9];
        eig3(C2, w, V);
        double z[3] = {V[0], V[3], V[6]};
        if (z[2] < 0) { z[0] = -z[0]; z[1] = -z[1]; z[2] = -z[2]; }  // normal away from the camera (documented)
        double x[3] = {1, 0, 0};
        if (fabs(z[0]) > 0.9) { x[0] = 0; x[1] = 1; }
        double y[3] = {z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0]};
        double xx[3] = {y[1] * z[2] - y[2] * z[1], y[2] * z[0] - y[0] * z[2], y[0] * z[1] - y[1] * z[0]};
        double M[25], rhs[5], co[5];
        for (int a = 0; a < 25; a++) M[a] = 0;
        for (int a = 0; a < 5; a++) rhs[a] = 0;
        for (int k = 0; k < K; k++) {
            double e[3] = {sP[3 * sNb[k]] - mu[0], sP[3 * sNb[k] + 1] - mu[1], sP[3 * sNb[k] + 2] - mu[2]};
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
        double a = co[0] * 2, b = co[1], c = co[2] * 2;
        double hd = (a - c) / 2.0, mid = (a + c) / 2.0, rad = sqrt(hd * hd + b * b);
        double lam = mid - rad;  // eig ascending: V(:,1)
        double v0, v1;
        if (fabs(lam - a) >= fabs(lam - c)) { v0 = b; v1 = lam - a; }
        else { v0 = lam - c; v1 = b; }
        double nn = sqrt(v0 * v0 + v1 * v1);
        if (nn == 0) { v0 = 1; v1 = 0; nn = 1; }
        v0 = v0 / nn;
        v1 = v1 / nn;
        for (int k = 0; k < 3; k++) dir0[k] = xx[k] * v0 + y[k] * v1;
    }

#pragma unroll
    for (int c = 0; c < 3; c++) {
        x0[c] = ctr[c] + rdir[c] * (R - d2s);
        x0[3 + c] = dir0[c];
    }
    f0_out = cyl_objective(x0, P, R, lane);
}

// fminsearch (MATLAB order) from x0
__device__ void fit_nm(const double *sP, int n, double R, int lane, double tolx, double tolf, int maxiter, int maxfun,
                       const double *x0, double f0, double *xf, double &ffinal, int &itercount, int &func_evals)
{
    Pts P{sP, n};
    // ---- fminsearch (MATLAB order).  simplex is wave-uniform, kept in registers.
    constexpr int N = 6;
    double v[N + 1][N], fv[N + 1];
#pragma unroll
    for (int k = 0; k < N; k++) v[0][k] = x0[k];
    fv[0] = f0;
#pragma unroll
    for (int j = 0; j < N; j++) {
#pragma unroll
        for (int k = 0; k < N; k++) v[j + 1][k] = x0[k];
        if (v[j + 1][j] != 0) v[j + 1][j] = (1 + 0.05) * v[j + 1][j];
        else v[j + 1][j] = 0.00025;
        fv[j + 1] = cyl_objective(v[j + 1], P, R, lane);
    }
    func_evals = N + 1;
    itercount = 1;
#define CSWAPV(a)                                                                       \
    if (fv[a] > fv[a + 1]) {                                                            \
        double t_ = fv[a]; fv[a] = fv[a + 1]; fv[a + 1] = t_;                           \
        _Pragma("unroll") for (int k = 0; k < N; k++) { double u_ = v[a][k]; v[a][k] = v[a + 1][k]; v[a + 1][k] = u_; } \
    }
#define SORT_SIMPLEX()                                                                  \
    CSWAPV(0)                                                                           \
    CSWAPV(1) CSWAPV(0)                                                                 \
    CSWAPV(2) CSWAPV(1) CSWAPV(0)                                                       \
    CSWAPV(3) CSWAPV(2) CSWAPV(1) CSWAPV(0)                                             \
    CSWAPV(4) CSWAPV(3) CSWAPV(2) CSWAPV(1) CSWAPV(0)                                   \
    CSWAPV(5) CSWAPV(4) CSWAPV(3) CSWAPV(2) CSWAPV(1) CSWAPV(0)
    SORT_SIMPLEX()
    while (func_evals < maxfun && itercount < maxiter) {
        double df = 0, dx = 0, vmax = v[0][0];
#pragma unroll
        for (int j = 1; j <= N; j++) {
            double a = fabs(fv[0] - fv[j]);
            if (a > df) df = a;
#pragma unroll
            for (int k = 0; k < N; k++) {
                double b = fabs(v[j][k] - v[0][k]);
                if (b > dx) dx = b;
            }
        }
#pragma unroll
        for (int k = 1; k < N; k++)
            if (v[0][k] > vmax) vmax = v[0][k];
        double tf = 10 * eps_of(fv[0]), tx = 10 * eps_of(vmax);
        if (df <= (tolf > tf ? tolf : tf) && dx <= (tolx > tx ? tolx : tx)) break;

        double xbar[N], xr[N], xt[N];
#pragma unroll
        for (int k = 0; k < N; k++) {
            double s = v[0][k];
#pragma unroll
            for (int j = 1; j < N; j++) s = s + v[j][k];
            xbar[k] = s / (double)N;
        }
#pragma unroll
        for (int k = 0; k < N; k++) xr[k] = 2.0 * xbar[k] - 1.0 * v[N][k];
        double fxr = cyl_objective(xr, P, R, lane);
        func_evals++;
        bool shrink = false;
        if (fxr < fv[0]) {
#pragma unroll
            for (int k = 0; k < N; k++) xt[k] = 3.0 * xbar[k] - 2.0 * v[N][k];
            double fxe = cyl_objective(xt, P, R, lane);
            func_evals++;
            if (fxe < fxr) {
#pragma unroll
                for (int k = 0; k < N; k++) v[N][k] = xt[k];
                fv[N] = fxe;
            } else {
#pragma unroll
                for (int k = 0; k < N; k++) v[N][k] = xr[k];
                fv[N] = fxr;
            }
        } else if (fxr < fv[N - 1]) {
#pragma unroll
            for (int k = 0; k < N; k++) v[N][k] = xr[k];
            fv[N] = fxr;
        } else if (fxr < fv[N]) {
#pragma unroll
            for (int k = 0; k < N; k++) xt[k] = 1.5 * xbar[k] - 0.5 * v[N][k];
            double fxc = cyl_objective(xt, P, R, lane);
            func_evals++;
            if (fxc <= fxr) {
#pragma unroll
                for (int k = 0; k < N; k++) v[N][k] = xt[k];
                fv[N] = fxc;
            } else shrink = true;
        } else {
#pragma unroll
            for (int k = 0; k < N; k++) xt[k] = 0.5 * xbar[k] + 0.5 * v[N][k];
            double fxcc = cyl_objective(xt, P, R, lane);
            func_evals++;
            if (fxcc < fv[N]) {
#pragma unroll
                for (int k = 0; k < N; k++) v[N][k] = xt[k];
                fv[N] = fxcc;
            } else shrink = true;
        }
        if (shrink) {
#pragma unroll
            for (int j = 1; j <= N; j++) {
#pragma unroll
                for (int k = 0; k < N; k++) v[j][k] = v[0][k] + 0.5 * (v[j][k] - v[0][k]);
                fv[j] = cyl_objective(v[j], P, R, lane);
            }
            func_evals += N;
        }
        SORT_SIMPLEX()
        itercount++;
    }
#undef SORT_SIMPLEX
#undef CSWAPV

#pragma unroll
    for (int k = 0; k < 6; k++) xf[k] = v[0][k];
    ffinal = fv[0];
}

__device__ void fit_lm(const double *sP, int n, double R, int lane, double tolx, double tolf, int maxiter,
                       const double *x0, double f0, double *xf, double &ffinal, int &itercount, int &func_evals)
{
    Pts P{sP, n};
    // ---- Levenberg-Marquardt on the same objective (north_star's "Gauss-Newton/LM inner loop"; NOT what the
    // reference runs -- fitCylinderWPts3.m:38 uses fminsearch -- validated against the Nelder-Mead result).
    // r_i = d_i - R;  dr/do = -e/d;  dr/dv = -(alpha/d) e  with  e = (P-o) - v alpha, alpha = ((P-o).v)/|v|^2.
    // The 6-parameter form has two gauge directions (origin along the axis, |v|): the damping term handles them.
    double x[6];
#pragma unroll
    for (int k = 0; k < 6; k++) x[k] = x0[k];
    double fx = f0, lambda = 1e-3;
    itercount = 0;
    func_evals = 1;
    for (; itercount < maxiter && itercount < 200;) {
        double A[21], g[6];
#pragma unroll
        for (int k = 0; k < 21; k++) A[k] = 0.0;
#pragma unroll
        for (int k = 0; k < 6; k++) g[k] = 0.0;
        {
            double p2[3] = {x[0] + x[3], x[1] + x[4], x[2] + x[5]};
            double vv[3] = {p2[0] - x[0], p2[1] - x[1], p2[2] - x[2]};
            double nv2 = (vv[0] * vv[0] + vv[1] * vv[1]) + vv[2] * vv[2];
            for (int k = lane; k < n; k += 64) {
                const double *pt = sP + 3 * k;
                double al = (((pt[0] - x[0]) * vv[0] + (pt[1] - x[1]) * vv[1]) + (pt[2] - x[2]) * vv[2]) / nv2;
                double e[3] = {pt[0] - (x[0] + vv[0] * al), pt[1] - (x[1] + vv[1] * al), pt[2] - (x[2] + vv[2] * al)};
                double dd = sqrt((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]);
                if (dd > 0) {
                    double r = dd - R, c1 = -1.0 / dd, c2 = -(al / dd);
                    double j[6] = {c1 * e[0], c1 * e[1], c1 * e[2], c2 * e[0], c2 * e[1], c2 * e[2]};
                    int q = 0;
#pragma unroll
                    for (int a = 0; a < 6; a++) {
#pragma unroll
                        for (int b = a; b < 6; b++) { A[q] = A[q] + j[a] * j[b]; q++; }
                        g[a] = g[a] + j[a] * r;
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 21; k++) A[k] = wave_sum(A[k]);
#pragma unroll
            for (int k = 0; k < 6; k++) g[k] = wave_sum(g[k]);
        }
        itercount++;
        bool accepted = false;
        double dmax = 0, fprev = fx;
        for (int tr = 0; tr < 12 && !accepted; tr++) {
            double M[36], rhs[6], dl[6];
            {
                int q = 0;
                double trA = 0;
#pragma unroll
                for (int a = 0; a < 6; a++)
#pragma unroll
                    for (int b = a; b < 6; b++) { M[a * 6 + b] = A[q]; M[b * 6 + a] = A[q]; if (a == b) trA = trA + A[q]; q++; }
#pragma unroll
                for (int a = 0; a < 6; a++) { M[a * 6 + a] = M[a * 6 + a] + lambda * M[a * 6 + a] + 1e-12 * trA; rhs[a] = -g[a]; }
            }
            // Gaussian elimination with partial pivoting (wave-uniform)
            bool singular = false;
            for (int c = 0; c < 6; c++) {
                int pv = c;
                for (int r = c + 1; r < 6; r++)
                    if (fabs(M[r * 6 + c]) > fabs(M[pv * 6 + c])) pv = r;
                if (M[pv * 6 + c] == 0) { singular = true; break; }
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
#pragma unroll
            for (int k = 0; k < 6; k++) xn[k] = x[k] + dl[k];
            double fn = cyl_objective(xn, P, R, lane);
            func_evals++;
            if (fn < fx) {
                dmax = 0;
#pragma unroll
                for (int k = 0; k < 6; k++) { dmax = fmax(dmax, fabs(dl[k])); x[k] = xn[k]; }
                fx = fn;
                lambda = fmax(lambda / 10, 1e-12);
                accepted = true;
            } else {
                lambda = lambda * 10;
            }
        }
        if (!accepted) break;
        if ((fprev - fx) <= tolf * 1e-3 * (1.0 + fx) && dmax <= tolx) break;
    }
#pragma unroll
    for (int k = 0; k < 6; k++) xf[k] = x[k];
    ffinal = fx;
}

// applyCylParamsPrior.m on both rows (ymin over the n points given), cylParams2T.m on the final row, outputs of frame f
__device__ void fit_write(int f, int lane, const double *sP, int n, const double *x0, const double *xf, double f0, double ffinal,
                          int itercount, int func_evals, double *__restrict__ o_raw, double *__restrict__ o_cyl,
                          double *__restrict__ o_T, double *__restrict__ o_fvals, int *__restrict__ o_iters,
                          int *__restrict__ o_status)
{
    // applyCylParamsPrior.m on both rows, cylParams2T.m on the final row
    double ymin = DBL_MAX;
    for (int k = lane; k < n; k += 64) ymin = fmin(ymin, sP[3 * k + 1]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ymin = fmin(ymin, __shfl_xor(ymin, off, 64));
    if (lane == 0) {
        double rows[2][6];
        for (int k = 0; k < 6; k++) { rows[0][k] = x0[k]; rows[1][k] = xf[k]; }
        for (int rI = 0; rI < 2; rI++) {
            for (int k = 0; k < 6; k++) o_raw[12 * f + 6 * rI + k] = rows[rI][k];
            double o[3] = {rows[rI][0], rows[rI][1], rows[rI][2]}, d[3] = {rows[rI][3], rows[rI][4], rows[rI][5]};
            if (d[1] < 0) { d[0] = -d[0]; d[1] = -d[1]; d[2] = -d[2]; }
            double t = 0;
            if (!(fabs(d[1]) < DBL_EPSILON)) t = (ymin - o[1]) / d[1];
            for (int c = 0; c < 3; c++) { rows[rI][c] = o[c] + t * d[c]; rows[rI][3 + c] = d[c]; }
            for (int k = 0; k < 6; k++) o_cyl[12 * f + 6 * rI + k] = rows[rI][k];
        }
        const double *cy = rows[1];
        double y[3] = {cy[3], cy[4], cy[5]};
        double ny = sqrt((y[0] * y[0] + y[1] * y[1]) + y[2] * y[2]);
        for (int c = 0; c < 3; c++) y[c] = y[c] / ny;
        double z[3] = {0 * y[2] - 0 * y[1], 0 * y[0] - 1 * y[2], 1 * y[1] - 0 * y[0]};
        double nz = sqrt((z[0] * z[0] + z[1] * z[1]) + z[2] * z[2]);
        for (int c = 0; c < 3; c++) z[c] = z[c] / nz;
        double xv[3] = {y[1] * z[2] - y[2] * z[1], y[2] * z[0] - y[0] * z[2], y[0] * z[1] - y[1] * z[0]};
        double nx = sqrt((xv[0] * xv[0] + xv[1] * xv[1]) + xv[2] * xv[2]);
        for (int c = 0; c < 3; c++) xv[c] = xv[c] / nx;
        double *T = o_T + 16 * f;
        for (int r = 0; r < 3; r++) { T[r * 4] = xv[r]; T[r * 4 + 1] = y[r]; T[r * 4 + 2] = z[r]; T[r * 4 + 3] = cy[r]; }
        T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1;
        o_fvals[2 * f] = f0;
        o_fvals[2 * f + 1] = ffinal;
        o_iters[2 * f] = itercount;
        o_iters[2 * f + 1] = func_evals;
        o_status[f] = CPE_ST_OK;
    }
}

__device__ void fit_write_few_points(int f, int lane, double *__restrict__ o_raw, double *__restrict__ o_cyl, double *__restrict__ o_T,
                                     double *__restrict__ o_fvals, int *__restrict__ o_iters, int *__restrict__ o_status)
{
    if (lane == 0) {
        o_status[f] = CPE_ST_FEW_POINTS;
        o_iters[2 * f] = 0; o_iters[2 * f + 1] = 0;
        o_fvals[2 * f] = 0; o_fvals[2 * f + 1] = 0;
        for (int k = 0; k < 12; k++) { o_raw[12 * f + k] = 0; o_cyl[12 * f + k] = 0; }
        for (int k = 0; k < 16; k++) o_T[16 * f + k] = 0;
    }
}

template <int MODE>
__global__ __launch_bounds__(64) void k_fit_cylinder(const double *__restrict__ X, const int *__restrict__ cnt,
                                                     double R, double tolx, double tolf, int maxiter,
                                                     int maxfun, double *__restrict__ o_raw,
                                                     double *__restrict__ o_cyl, double *__restrict__ o_T,
                                                     double *__restrict__ o_fvals, int *__restrict__ o_iters,
                                                     int *__restrict__ o_status)
{
    // One wavefront per frame, a chain of dependent f64 operations 200-300 simplex iterations long: beside the wide kernels of
    // the other chunk in flight it waited its turn at every instruction (2.2 ms alone, 12 ms in the overlapped trace).  Its
    // few waves ask for the highest issue priority of their SIMD; the wide kernels fill what is left (2.2 ms again in the
    // overlapped trace; the bench rate does not move -- A/B on one box: 3166 / 3199 / 3198 with, 3175 / 3203 / 3182 without --,
    // the latency of a chunk does).  The same on the other narrow kernels of the chain changed nothing measurable and was dropped.
    __builtin_amdgcn_s_setprio(3);
    double *sP = fit_dyn, *sD = fit_dyn + MAXP * 3;   // dynamic LDS, FIT_LDS_BYTES
    __shared__ int sNb[20];
    const int f = blockIdx.x, lane = threadIdx.x;
    const int n = min(max(cnt[f], 0), MAXP);
    if (n < 3) { fit_write_few_points(f, lane, o_raw, o_cyl, o_T, o_fvals, o_iters, o_status); return; }
    const double *Xf = X + (size_t)f * MAXP * 3;
    for (int i = lane; i < 3 * n; i += 64) sP[i] = Xf[i];
    __syncthreads();
    double x0[6], f0, xf[6], ffinal;
    int itercount, func_evals;
    fit_init(sP, n, R, lane, sD, sNb, x0, f0);
    if constexpr (MODE == 0) fit_nm(sP, n, R, lane, tolx, tolf, maxiter, maxfun, x0, f0, xf, ffinal, itercount, func_evals);
    else fit_lm(sP, n, R, lane, tolx, tolf, maxiter, x0, f0, xf, ffinal, itercount, func_evals);
    fit_write(f, lane, sP, n, x0, xf, f0, ffinal, itercount, func_evals, o_raw, o_cyl, o_T, o_fvals, o_iters, o_status);
}

// ---- BUILD-DEFINED (BASELINE config 5; nothing like it in the reference): RANSAC around the fit -------
// H hypotheses per frame, one wavefront per frame.  Hypothesis 0 uses all points; hypothesis h > 0 keeps point k with
// probability S/n, decided by a counter-based hash of (seed, frame, h, k), so every lane decides for its own points and the
// subset does not depend on any sequential generator state.  A hypothesis = `hyp_iters` LM iterations on its subset
// (compacted into LDS in point order) from the all-points initial cylinder, scored by the number of points with
// | dist(point, axis) - R | < tau over ALL points; the first hypothesis with the largest count wins and the final fit
// (Nelder-Mead or LM) runs on its inliers.  Same arithmetic, same order as oracle/src/orc_fit.c:orc_fit_cylinder_ransac.
__device__ __forceinline__ unsigned long long ransac_hash(unsigned long long seed, unsigned long long frame, unsigned long long h,
                                                          unsigned long long k)
{
    unsigned long long z = (seed ^ (frame * 0xD1B54A32D192ED03ULL) ^ (h << 32) ^ k) + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// |dist - R| < tau for point k of sP under the cylinder x
__device__ __forceinline__ bool ransac_inlier(const double *sP, int k, const double *x, double R, double tau)
{
    double p2[3] = {x[0] + x[3], x[1] + x[4], x[2] + x[5]};
    double v[3] = {p2[0] - x[0], p2[1] - x[1], p2[2] - x[2]};
    double nv2 = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2];
    return fabs(dist_pt_line(sP + 3 * k, x, v, nv2) - R) < tau;
}

template <int MODE>
__global__ __launch_bounds__(64) void k_fit_ransac(const double *__restrict__ X, const int *__restrict__ cnt, double R, int H, int S,
                                                   double tau, unsigned long long seed, unsigned long long frame0, int hyp_iters,
                                                   double tolx, double tolf, int maxiter, int maxfun,
                                                   double *__restrict__ o_raw, double *__restrict__ o_cyl, double *__restrict__ o_T,
                                                   double *__restrict__ o_fvals, int *__restrict__ o_iters, int *__restrict__ o_status,
                                                   int *__restrict__ o_ninl, uint8_t *__restrict__ o_mask)
{
    double *sP = fit_dyn, *sQ = fit_dyn + MAXP * 3, *sD = fit_dyn + MAXP * 6;   // dynamic LDS, RANSAC_LDS_BYTES
    __shared__ int sNb[20];
    const int f = blockIdx.x, lane = threadIdx.x;
    const int n = min(max(cnt[f], 0), MAXP);
    uint8_t *mk = o_mask + (size_t)f * MAXP;
    for (int k = lane; k < MAXP; k += 64) mk[k] = 0;
    if (n < 3) {
        fit_write_few_points(f, lane, o_raw, o_cyl, o_T, o_fvals, o_iters, o_status);
        if (lane == 0) o_ninl[f] = 0;
        return;
    }
    const double *Xf = X + (size_t)f * MAXP * 3;
    for (int i = lane; i < 3 * n; i += 64) sP[i] = Xf[i];
    __syncthreads();
    double x0[6], f0;
    fit_init(sP, n, R, lane, sD, sNb, x0, f0);
    const double q = (double)S / (double)n;
    const unsigned long long frame = frame0 + (unsigned long long)f;
    int best_cnt = -1;
    double best_x[6] = {0, 0, 0, 0, 0, 0};
    // keep[k] -> sQ, in point order; returns the subset size (wave-uniform)
    auto compact = [&](auto keep) {
        int nq = 0;
        __syncthreads();
        for (int base = 0; base < n; base += 64) {
            const int k = base + lane;
            const bool take = k < n && keep(k);
            const unsigned long long b = __ballot(take);
            if (take) {
                const int pos = nq + __popcll(b & ((1ull << lane) - 1ull));
                sQ[3 * pos] = sP[3 * k]; sQ[3 * pos + 1] = sP[3 * k + 1]; sQ[3 * pos + 2] = sP[3 * k + 2];
            }
            nq += __popcll(b);
        }
        __syncthreads();
        return nq;
    };
    for (int h = 0; h < H; h++) {
        const int nq = compact([&](int k) {
            return h == 0 || ((double)(ransac_hash(seed, frame, (unsigned long long)h, (unsigned long long)k) >> 11) * 0x1.0p-53) < q;
        });
        if (nq < 6) continue;
        double xh[6], fh;
        int it, ev;
        const double fq0 = cyl_objective(x0, Pts{sQ, nq}, R, lane);
        fit_lm(sQ, nq, R, lane, tolx, tolf, hyp_iters, x0, fq0, xh, fh, it, ev);
        int c = 0;
        for (int k = lane; k < n; k += 64) c += ransac_inlier(sP, k, xh, R, tau) ? 1 : 0;
        c = wave_sum_i(c);
        if (c > best_cnt) {
            best_cnt = c;
#pragma unroll
            for (int k = 0; k < 6; k++) best_x[k] = xh[k];
        }
    }
    if (best_cnt < 0) {
#pragma unroll
        for (int k = 0; k < 6; k++) best_x[k] = x0[k];
    }
    int nq = compact([&](int k) { return ransac_inlier(sP, k, best_x, R, tau); });
    const int n_inl = nq;
    if (nq < 6) {   // too few inliers to fit: all points, and say so in the mask
        nq = compact([&](int) { return true; });
        for (int k = lane; k < n; k += 64) mk[k] = 1;
    } else {
        for (int k = lane; k < n; k += 64) mk[k] = ransac_inlier(sP, k, best_x, R, tau) ? 1 : 0;
    }
    const double fs = cyl_objective(best_x, Pts{sQ, nq}, R, lane);
    double xf[6], ffinal;
    int itercount, func_evals;
    if constexpr (MODE == 0) fit_nm(sQ, nq, R, lane, tolx, tolf, maxiter, maxfun, best_x, fs, xf, ffinal, itercount, func_evals);
    else fit_lm(sQ, nq, R, lane, tolx, tolf, maxiter, best_x, fs, xf, ffinal, itercount, func_evals);
    fit_write(f, lane, sQ, nq, x0, xf, f0, ffinal, itercount, func_evals, o_raw, o_cyl, o_T, o_fvals, o_iters, o_status);
    if (lane == 0) o_ninl[f] = n_inl;
}

}  // namespace

namespace {
// triangulate(matchedPoints1, matchedPoints2, stereoParams) for n frames of already matched pairs (fitSingleCylinder.m:15-17):
// one wavefront per frame, one lane per point; meanError through the same 64-lane tree as everywhere else
__global__ __launch_bounds__(64) void k_triangulate(const double *__restrict__ p1, const double *__restrict__ p2,
                                                    const int *__restrict__ cnt, const double *__restrict__ K1,
                                                    const double *__restrict__ K2, const double *__restrict__ T21,
                                                    double *__restrict__ o_X, double *__restrict__ o_err, double *__restrict__ o_mean_err)
{
    const int f = blockIdx.x, lane = threadIdx.x;
    const int m = min(max(cnt[f], 0), MAXP);
    double P1[12], P2[12];
    {
        const double I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        double k1[9], k2[9], tt[16];
        for (int k = 0; k < 9; k++) { k1[k] = K1[k]; k2[k] = K2[k]; }
        for (int k = 0; k < 16; k++) tt[k] = T21[k];
        make_P(k1, I4, P1);
        make_P(k2, tt, P2);
    }
    double es = 0.0;
    for (int k = lane; k < m; k += 64) {
        const size_t o = (size_t)f * MAXP + k;
        double X[3], e;
        triangulate_one(P1, P2, p1[2 * o], p1[2 * o + 1], p2[2 * o], p2[2 * o + 1], X, e);
        o_X[3 * o] = X[0]; o_X[3 * o + 1] = X[1]; o_X[3 * o + 2] = X[2];
        o_err[o] = e;
        es = es + e;
    }
    es = wave_sum(es);
    if (lane == 0) o_mean_err[f] = m > 0 ? es / (double)m : 0.0;
}
}  // namespace

extern "C" size_t cpe_fit_workspace_bytes(int32_t n)
{
    // index tables of the selector + room for the by-products cpe_choose_idx_batch does not hand out (X, err, mean_err)
    return (size_t)(n > 0 ? n : 0) * (2 * TBL * TBL * sizeof(int) + (size_t)MAXP * 4 * sizeof(double) + sizeof(double));
}

extern "C" int32_t cpe_select_triangulate_batch(const double *xy1, const int32_t *id1, const int32_t *cnt1,
                                                const double *xy2, const int32_t *id2, const int32_t *cnt2,
                                                int32_t n, const double *K1, const double *K2, const double *T21,
                                                int32_t selector, int32_t patch, double th, void *ws, size_t ws_bytes,
                                                double *p1, double *p2, int32_t *idx, double *X, double *err,
                                                int32_t *m, double *mean_err, int32_t *flags, void *stream)
{
    CPE_CHECK_ARG(xy1 && id1 && cnt1 && xy2 && id2 && cnt2 && K1 && K2 && T21 && p1 && p2 && idx && X && err && m &&
                      mean_err && flags,
                  "cpe_select_triangulate_batch: null pointer");
    CPE_CHECK_ARG(n >= 0, "cpe_select_triangulate_batch: n < 0");
    CPE_CHECK_ARG(selector >= 0 && selector <= 2, "cpe_select_triangulate_batch: selector must be 0,1,2");
    CPE_CHECK_ARG(patch >= 1 && patch <= 8, "cpe_select_triangulate_batch: patch must be 1..8");
    if (n == 0) return CPE_OK;
    if (!ws || ws_bytes < cpe_fit_workspace_bytes(n)) {
        cpe::set_error("cpe_select_triangulate_batch: workspace too small (%zu < %zu)", ws_bytes, cpe_fit_workspace_bytes(n));
        return CPE_ERR_WORKSPACE;
    }
    CPE_LAUNCH_BEGIN();
    CPE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_select_triangulate), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SEL_LDS_BYTES));
    CPE_KLAUNCH(k_select_triangulate, dim3(n), dim3(64), SEL_LDS_BYTES, (hipStream_t)stream, xy1, id1, cnt1, xy2, id2, cnt2,
                       K1, K2, T21, selector, patch, th, (int *)ws, p1, p2, idx, X, err, m, mean_err, flags);
    CPE_CHECK_LAUNCH("k_select_triangulate");
    return CPE_OK;
}

extern "C" int32_t cpe_choose_idx_batch(const double *xy1, const int32_t *id1, const int32_t *cnt1, const double *xy2,
                                        const int32_t *id2, const int32_t *cnt2, int32_t n, const double *K1, const double *K2,
                                        const double *T21, int32_t patch, double th, void *ws, size_t ws_bytes, double *p1,
                                        double *p2, int32_t *idx, int32_t *m, int32_t *flags, void *stream)
{
    CPE_CHECK_ARG(n >= 0, "cpe_choose_idx_batch: n < 0");
    if (n == 0) return CPE_OK;
    if (!ws || ws_bytes < cpe_fit_workspace_bytes(n)) {
        cpe::set_error("cpe_choose_idx_batch: workspace too small (%zu < %zu)", ws_bytes, cpe_fit_workspace_bytes(n));
        return CPE_ERR_WORKSPACE;
    }
    // the selector triangulates every candidate anyway: its points and errors land in the workspace behind the index tables
    double *X = reinterpret_cast<double *>(static_cast<char *>(ws) + (size_t)n * 2 * TBL * TBL * sizeof(int));
    double *err = X + (size_t)n * MAXP * 3, *mean_err = err + (size_t)n * MAXP;
    return cpe_select_triangulate_batch(xy1, id1, cnt1, xy2, id2, cnt2, n, K1, K2, T21, CPE_SEL_CHOOSE_IDX, patch, th, ws, ws_bytes,
                                        p1, p2, idx, X, err, m, mean_err, flags, stream);
}

extern "C" int32_t cpe_triangulate_batch(const double *p1, const double *p2, const int32_t *cnt, int32_t n, const double *K1,
                                         const double *K2, const double *T21, double *X, double *err, double *mean_err,
                                         void *stream)
{
    CPE_CHECK_ARG(p1 && p2 && cnt && K1 && K2 && T21 && X && err && mean_err, "cpe_triangulate_batch: null pointer");
    CPE_CHECK_ARG(n >= 0, "cpe_triangulate_batch: n < 0");
    if (n == 0) return CPE_OK;
    CPE_LAUNCH_BEGIN();
    CPE_KLAUNCH(k_triangulate, dim3(n), dim3(64), 0, (hipStream_t)stream, p1, p2, cnt, K1, K2, T21, X, err, mean_err);
    CPE_CHECK_LAUNCH("k_triangulate");
    return CPE_OK;
}

extern "C" int32_t cpe_fit_cylinder_batch(const double *X, const int32_t *cnt, int32_t n, double radius,
                                          const CpeFitParams *params, double *cyl_raw, double *cyl, double *T,
                                          double *fvals, int32_t *iters, int32_t *status, void *stream)
{
    CPE_CHECK_ARG(X && cnt && cyl_raw && cyl && T && fvals && iters && status, "cpe_fit_cylinder_batch: null pointer");
    CPE_CHECK_ARG(n >= 0, "cpe_fit_cylinder_batch: n < 0");
    CpeFitParams p = {1e-5, 1e-5, 100000, 100000, CPE_FIT_NELDER_MEAD, 0};
    if (params) p = *params;
    CPE_CHECK_ARG(p.tol_x >= 0 && p.tol_f >= 0 && p.max_iter > 0 && p.max_fun_evals > 0,
                  "cpe_fit_cylinder_batch: bad CpeFitParams");
    if (n == 0) return CPE_OK;
    CPE_LAUNCH_BEGIN();
    CPE_CHECK_ARG(p.mode == CPE_FIT_NELDER_MEAD || p.mode == CPE_FIT_LM, "cpe_fit_cylinder_batch: unknown mode %d", p.mode);
    CPE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fit_cylinder<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FIT_LDS_BYTES));
    CPE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fit_cylinder<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FIT_LDS_BYTES));
    if (p.mode == CPE_FIT_LM)
        CPE_KLAUNCH(k_fit_cylinder<1>, dim3(n), dim3(64), FIT_LDS_BYTES, (hipStream_t)stream, X, cnt, radius, p.tol_x, p.tol_f, p.max_iter,
                    p.max_fun_evals, cyl_raw, cyl, T, fvals, iters, status);
    else
        CPE_KLAUNCH(k_fit_cylinder<0>, dim3(n), dim3(64), FIT_LDS_BYTES, (hipStream_t)stream, X, cnt, radius, p.tol_x, p.tol_f, p.max_iter,
                    p.max_fun_evals, cyl_raw, cyl, T, fvals, iters, status);
    CPE_CHECK_LAUNCH("k_fit_cylinder");
    return CPE_OK;
}

extern "C" int32_t cpe_fit_cylinder_ransac_batch(const double *X, const int32_t *cnt, int32_t n, double radius,
                                                 const CpeFitParams *params, const CpeRansacParams *ransac, double *cyl_raw,
                                                 double *cyl, double *T, double *fvals, int32_t *iters, int32_t *status,
                                                 int32_t *n_inliers, uint8_t *inlier_mask, void *stream)
{
    CPE_CHECK_ARG(X && cnt && cyl_raw && cyl && T && fvals && iters && status && n_inliers && inlier_mask,
                  "cpe_fit_cylinder_ransac_batch: null pointer");
    CPE_CHECK_ARG(n >= 0, "cpe_fit_cylinder_ransac_batch: n < 0");
    CpeFitParams p = {1e-5, 1e-5, 100000, 100000, CPE_FIT_LM, 0};
    if (params) p = *params;
    CpeRansacParams r = {64, 12, 0.5, 0, 0, 8, 0};
    if (ransac) r = *ransac;
    CPE_CHECK_ARG(p.tol_x >= 0 && p.tol_f >= 0 && p.max_iter > 0 && p.max_fun_evals > 0, "cpe_fit_cylinder_ransac_batch: bad CpeFitParams");
    CPE_CHECK_ARG(p.mode == CPE_FIT_NELDER_MEAD || p.mode == CPE_FIT_LM, "cpe_fit_cylinder_ransac_batch: unknown mode %d", p.mode);
    CPE_CHECK_ARG(r.hypotheses >= 1 && r.hypotheses <= 4096 && r.sample >= 6 && r.tau > 0 && r.hyp_iters >= 1,
                  "cpe_fit_cylinder_ransac_batch: bad CpeRansacParams (hypotheses 1..4096, sample >= 6, tau > 0, hyp_iters >= 1)");
    if (n == 0) return CPE_OK;
    CPE_LAUNCH_BEGIN();
    CPE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fit_ransac<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RANSAC_LDS_BYTES));
    CPE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fit_ransac<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RANSAC_LDS_BYTES));
    if (p.mode == CPE_FIT_LM)
        CPE_KLAUNCH(k_fit_ransac<1>, dim3(n), dim3(64), RANSAC_LDS_BYTES, (hipStream_t)stream, X, cnt, radius, r.hypotheses, r.sample, r.tau,
                    (unsigned long long)r.seed, (unsigned long long)r.frame0, r.hyp_iters, p.tol_x, p.tol_f, p.max_iter, p.max_fun_evals,
                    cyl_raw, cyl, T, fvals, iters, status, n_inliers, inlier_mask);
    else
        CPE_KLAUNCH(k_fit_ransac<0>, dim3(n), dim3(64), RANSAC_LDS_BYTES, (hipStream_t)stream, X, cnt, radius, r.hypotheses, r.sample, r.tau,
                    (unsigned long long)r.seed, (unsigned long long)r.frame0, r.hyp_iters, p.tol_x, p.tol_f, p.max_iter, p.max_fun_evals,
                    cyl_raw, cyl, T, fvals, iters, status, n_inliers, inlier_mask);
    CPE_CHECK_LAUNCH("k_fit_ransac");
    return CPE_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Row f-1: objective of the multi-frame AGV-pose fit, utils/fitCylinderWPts3sAngs.m:82-94 (`dist`):
//   v = sum_i mean((d_i - R)^2),  d_i = getDistPts3ToLine(Pts3s{i}, line of T * TAGVcyls{i})
// One wavefront per frame computes its term (same 64-lane reduction tree as the per-frame fit); the host-side
// Nelder-Mead (cpe_amd/multiframe.py, MATLAB fminsearch order) adds the F terms in frame order.
namespace {
__global__ __launch_bounds__(64) void k_multi_frame_terms(const double *__restrict__ X, const int *__restrict__ cnt,
                                                          const double *__restrict__ TAGV, const double *__restrict__ T,
                                                          double R, double *__restrict__ terms)
{
    const int f = blockIdx.x, lane = threadIdx.x;
    const int n = min(max(cnt[f], 0), MAXP);
    if (n == 0) { if (lane == 0) terms[f] = 0.0; return; }
    // T_C1_cyl = T * TAGVcyls{i}: only column 2 (axis) and column 4 (origin) are used
    const double *A = TAGV + 16 * (size_t)f;
    double org[3], dy[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        dy[r] = ((T[r * 4] * A[1] + T[r * 4 + 1] * A[5]) + T[r * 4 + 2] * A[9]) + T[r * 4 + 3] * A[13];
        org[r] = ((T[r * 4] * A[3] + T[r * 4 + 1] * A[7]) + T[r * 4 + 2] * A[11]) + T[r * 4 + 3] * A[15];
    }
    double p2[3] = {org[0] + dy[0], org[1] + dy[1], org[2] + dy[2]};
    double v[3] = {p2[0] - org[0], p2[1] - org[1], p2[2] - org[2]};
    double nv2 = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2];
    const double *P = X + (size_t)f * MAXP * 3;
    double acc = 0.0;
    for (int k = lane; k < n; k += 64) {
        double d = dist_pt_line(P + 3 * k, org, v, nv2);
        double w = d - R;
        acc = acc + w * w;
    }
    acc = wave_sum(acc);
    if (lane == 0) terms[f] = acc / (double)n;
}
}  // namespace

extern "C" int32_t cpe_multi_frame_terms(const double *X, const int32_t *cnt, int32_t n, const double *TAGVcyl,
                                         const double *T, double radius, double *terms, void *stream)
{
    CPE_CHECK_ARG(X && cnt && TAGVcyl && T && terms && n >= 0, "cpe_multi_frame_terms: bad argument");
    if (n == 0) return CPE_OK;
    CPE_LAUNCH_BEGIN();
    CPE_KLAUNCH(k_multi_frame_terms, dim3(n), dim3(64), 0, (hipStream_t)stream, X, cnt, TAGVcyl, T, radius, terms);
    CPE_CHECK_LAUNCH("k_multi_frame_terms");
    return CPE_OK;
}
