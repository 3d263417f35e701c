// Stage a-1 -- fused pre-process kernel for gfx950, row-marching form.
//   reference: utils/util_cylinder.py:1769-1802 (load_and_preprocess_image), :1734-1738
//   (detect_ridges -> skimage hessian_matrix / hessian_matrix_eigvals), :1740-1766 (Sauvola).
//
// One 512-thread workgroup owns a column strip of SW = 128 output columns of one frame and marches down it K = 4 rows
// per step.  Every stage of the chain keeps only the rows its consumer still needs, as a ring in LDS (77 KB per
// workgroup: two workgroups share a CU):
//   gray   8 x 176 u8    x = sx0-24 ..   rows r-2 .. r+2 of the 5x5 binomial       (LDS-DMA, an interval ahead)
//   blur5 32 x 176 u8    x = sx0-22 ..   integer, exact, BORDER_REFLECT_101; 0 outside the image (scipy: mode='constant')
//   V      4 x 170 f64   x = sx0-21 ..   Gaussian sigma 3 along y (scipy applies axis 0 first), 25 taps
//   G      8 x 146 f64   x = sx0-9  ..   Gaussian along x; rows r-2 .. r+2 of the two np.gradient passes
//   b     15 x 142 f64   x = sx0-7  ..   smaller Hessian eigenvalue at clamped coordinates (BORDER_REPLICATE of the box)
//   rs  2x18 x 128 f64                   15-tap row sums of b and b*b (cv2 RowSum, restarted every 8 columns)
// and the column sums of the box filter are cv2's ColumnSum itself: a running SUM per column (two registers of the
// thread that owns the column), s0 = SUM + Sp, out = s0 * scale, SUM = s0 - Sm, from the top of the frame to the bottom.
// No output row is evaluated twice (the 64x64 tiles of the first version evaluated the y-Gaussian on 82x106 points per
// tile); along x the strip's halo costs (128 + 42) / 128 on the y-Gaussian and (128 + 18) / 128 on the x-Gaussian.
//
// A step is two barrier intervals; the stages alternate between them and run side by side on different waves, each on
// the rows its producer finished in the interval before (stage p of step s works on rows s*K - L_p ...):
//   interval A: wave 0 row sums (P5) | waves 1-2 blur5 (P1) | waves 3-7 x-Gaussian (P3)
//   interval B: waves 0-1 column sums + Sauvola + mask store (P6) | waves 2-4 y-Gaussian (P2) | waves 5-7 eigenvalue (P4) | wave 7 gray (P0)
// (with two workgroups per CU a wave gets about a quarter of its SIMD: an interval lasts as long as its longest role,
//  ~16 cycles per instruction of it; the roles are cut so that the longest ones of A and B are as short as they get)
// The kernel is bound by vector-ALU issue (rocprofv3 SQ_ACTIVE_INST_VALU: 3/4 of all SIMD cycles), so the work per stage
// is laid out for few instructions: column owners with scalar ring offsets, SWAR integer arithmetic in the binomial.
// The f64 operation order is the one scipy / numpy / OpenCV execute (centre tap, then symmetric pairs from the outside
// in; mul and add kept separate: built with -ffp-contract=off), so the mask is bit-identical to the CPU oracle, which
// in turn is bit-identical to the real skimage for b.
// HBM traffic per frame: h*w read (+ 48 halo columns per strip, served by L2) + h*w written.
#include "cpe_internal.h"
#include <algorithm>
#include <cstddef>
#include <type_traits>

namespace {

constexpr int SW = 128, K = 4, NT = 512;
// row lag of every stage behind the step counter: stage p of step s produces rows s*K - Lp .. + K-1
constexpr int L0 = K;                // gray rows stored into LDS (end of B)
constexpr int L1 = L0 + K + 2;       // blur5 (A): reads gray r-2 .. r+2
constexpr int L2 = L1 + 12;          // V (B): reads blur5 r-12 .. r+12
constexpr int L3 = L2 + K;           // G (A): reads V of the same rows
constexpr int L4 = L3 + 2;           // b (B): reads G r-2 .. r+2
constexpr int L5 = L4 + K;           // rs (A): reads b of the same rows
constexpr int L6 = L5 + 7;           // mask (B): reads rs r-7 .. r+7
constexpr int GYW = SW + 48, GYR = 8;          // gray ring: K + 4 rows
constexpr int B5W = 176, B5R = 32;             // blur5 ring (K + 24 rows live): column u <-> x = sx0 - 22 + u
constexpr int VW = SW + 42, VST = 172;         // V: K rows
constexpr int GW = SW + 18, GST = 148, GR = 8; // G ring: K + 4 rows
constexpr int EW = SW + 14, EST = 145, ER = 2 * K + 7;
constexpr int RST = 129, RR = K + 14, RPL = RR * RST + 2;
static_assert(K == 4 && GR == K + 4 && GYR == K + 4 && (L2 + 12) % 4 == 2 && L1 % 4 == 2 && L4 % 4 == 0, "ring sizes / block alignment");

__constant__ double c_gw[13] = {
    0x1.105a329f98197p-3, 0x1.01a25f86eb137p-3, 0x1.b42a57d56c0bep-4,
    0x1.4a614d1afd337p-4, 0x1.bfde9c12bec92p-5, 0x1.0fa58939b5290p-5,
    0x1.26defcaeb0201p-6, 0x1.1e6bccad344bap-7, 0x1.f1e9915139407p-9,
    0x1.8345966f69519p-10, 0x1.0d8a5ad43c165p-11, 0x1.4fbe39149e277p-13,
    0x1.763a210dfb305p-15};

struct Smem {
    alignas(16) uint8_t gray[GYR * GYW + 16];   // first: the LDS-DMA destination base (M0) is a 16-bit byte address.
                                                // + slack: the last strip of P1 reads one dword past its row
    alignas(16) uint8_t b5[B5R * B5W];
    alignas(16) double lut[256];          // (double)v * (1.0 / 255): img_as_float of a u8 value
    alignas(16) double v[K * VST];
    alignas(16) double g[GR * GST];
    alignas(16) double e[ER * EST];
    alignas(16) double rs[2 * RPL];
};
static_assert(offsetof(Smem, b5) < 65536, "LDS-DMA destination");
static_assert(sizeof(Smem) <= 80 * 1024, "two workgroups per CU");

// items per stage and step
constexpr int N1 = K * 22;             // blur5: 8 outputs per item
constexpr int N3 = K * (GW / 2);       // x-Gaussian: 2 outputs per item
constexpr int N2 = VW;                 // y-Gaussian: K outputs per item
constexpr int N0 = K * (GYW / 4);      // gray dwords
static_assert(N1 <= 128 && N3 <= 320 && N2 <= 192 && N0 > 128 && N0 <= 192 && EW == 128 + 14 && GW % 2 == 0, "wave roles");

// the blur5 ring slot of row r: rows 4m+2 .. 4m+5 share an aligned block of 4 slots (the y-Gaussian's 28-row window
// starts at such a row, the binomial writes such a block)
__device__ __forceinline__ int b5slot(int r) { return (r + 126) & (B5R - 1); }

// N square roots side by side.  The operations are exactly those of hipcc's own f64 sqrt (AMDGPU lowerFSQRTF64: v_rsq_f64, one
// Goldschmidt step on g ~ sqrt(x) and h ~ 1 / (2 sqrt(x)), two residual corrections, x returned for 0 and +inf), so every
// result has the bits `sqrt(x)` has (correctly rounded, the C oracle's libm result); written stage by stage over the N
// values because hipcc schedules N calls of sqrt() one after the other, each a chain of a dozen dependent instructions.
// Left out: the 2^256 pre-scaling hipcc applies to arguments below 2^-767.  The arguments here are sums / differences of
// squares of numbers that are sums of at most 2^10 doubles of magnitude 2^-45 .. 2^1: a non-zero one is above 2^-300.
// Zero: hipcc returns x itself for 0 and +inf (rsq(0) = inf would make g = NaN); here rsq sees max(x, 2^-600) instead, which
// changes nothing for the non-zero arguments and gives g = 0 * 2^300 = 0 for x = 0, and 0 stays 0 through every step.
// +inf and NaN cannot occur.
template <int N> __device__ __forceinline__ void sqrt_n(const double (&x)[N], double (&out)[N])
{
    double g[N], hh[N], r[N], d[N];
#pragma unroll
    for (int i = 0; i < N; i++) { const double y = __builtin_amdgcn_rsq(__builtin_fmax(x[i], 0x1p-600)); g[i] = x[i] * y; hh[i] = y * 0.5; }
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = __builtin_fma(-hh[i], g[i], 0.5);
#pragma unroll
    for (int i = 0; i < N; i++) { g[i] = __builtin_fma(g[i], r[i], g[i]); hh[i] = __builtin_fma(hh[i], r[i], hh[i]); }
#pragma unroll
    for (int i = 0; i < N; i++) d[i] = __builtin_fma(-g[i], g[i], x[i]);
#pragma unroll
    for (int i = 0; i < N; i++) g[i] = __builtin_fma(d[i], hh[i], g[i]);
#pragma unroll
    for (int i = 0; i < N; i++) d[i] = __builtin_fma(-g[i], g[i], x[i]);
#pragma unroll
    for (int i = 0; i < N; i++) g[i] = __builtin_fma(d[i], hh[i], g[i]);
#pragma unroll
    for (int i = 0; i < N; i++) out[i] = g[i];
}

#ifdef CPE_PRE_STAMPS
// diagnostic build only (tools/probe/pre_stamps.cpp): cycles per wave role in interval A, at the barrier behind it, in
// interval B and at its barrier, summed over steps and workgroups.  The product build has no stamp.
__device__ unsigned long long g_stamps[8][5];
#define STAMP(v) do { __builtin_amdgcn_sched_barrier(0); v = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(v) do { } while (0)
#endif

// FAST: the strip's gray window lies inside the frame and its rows are 4-byte aligned (LDS-DMA of dwords); the two forms
// are separate instantiations so that the byte loads of the other one (ordinary loads into registers, whose hazards the
// compiler guards with vmcnt waits) put no wait into this one, where a wait would drain the DMA early
template <bool FAST>
__device__ __forceinline__ void preprocess_strip(Smem &s, const uint8_t *__restrict__ img, int h, int w, int sx0, uint8_t *__restrict__ out)
{
    const int tid0 = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const double inv255 = 1.0 / 255;
    constexpr bool fast = FAST;

    for (int i = tid0; i < 256; i += NT) s.lut[i] = (double)i * inv255;
    for (int i = tid0; i < B5R * B5W / 4; i += NT) reinterpret_cast<uint32_t *>(s.b5)[i] = 0u;   // rows above the frame

    // P0: the next gray row block goes from global memory straight into the ring by LDS-DMA (global_load_lds: per-lane
    // source address, destination = wave-uniform LDS base + lane x size; no register holds the data, the barrier at the
    // end of the interval retires it): 3 dword instructions of wave 7, whose other roles are the lightest.  For strips
    // inside the frame with 4-byte aligned rows; the byte form of the instruction writes a zero-extended dword per lane,
    // so the other strips take the path below.
    typedef __attribute__((address_space(1))) const void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    auto gray_row = [&](int row0, int k) -> const uint8_t * {
        int r = row0 + k;
        if (row0 < 0 || row0 + K > h) r = cpe::reflect101(cpe::clampi(r, -2, h + 1), h);   // (uniform) first and last blocks only
        return img + (size_t)r * w;
    };
    auto gray_fetch = [&](int row0, int lane) {
        uint8_t *dst = &s.gray[((row0 + 64) & (GYR - 1)) * GYW];        // K ring rows, contiguous (row0 is a multiple of K)
        if (fast) {
            if (wave == 7) {
#pragma unroll
                for (int t = 0; t < 3; t++) {
                    const int q = t * 64 + lane;
                    if (q < N0) {
                        const int qk = q / (GYW / 4), qd = q - qk * (GYW / 4);
                        __builtin_amdgcn_global_load_lds((gptr_t)(gray_row(row0, qk) + sx0 - 24 + 4 * qd), (lptr_t)(dst + 256 * t), 4, 0, 0);
                    }
                }
            }
        }
    };
    // the other strips (frame border inside the window, or rows not 4-byte aligned): bytes from BORDER_REFLECT_101 addresses,
    // two per thread, loaded at the top of a step and stored into the ring in interval B
    auto gray_load_bytes = [&](int row0, int tid, uint8_t &v0, uint8_t &v1) {
        const int qk = tid / GYW, xq = tid - qk * GYW;
        v0 = gray_row(row0, qk)[cpe::reflect101(sx0 - 24 + xq, w)];
        if (tid < K * GYW - NT) {
            const int i = tid + NT, qk1 = i / GYW, xq1 = i - qk1 * GYW;
            v1 = gray_row(row0, qk1)[cpe::reflect101(sx0 - 24 + xq1, w)];
        }
    };
    auto gray_store_bytes = [&](int row0, int tid, uint8_t v0, uint8_t v1) {
        uint8_t *dst = &s.gray[((row0 + 64) & (GYR - 1)) * GYW];
        dst[tid] = v0;
        if (tid < K * GYW - NT) dst[tid + NT] = v1;
    };
    static_assert(K * GYW > NT && K * GYW <= 2 * NT && GYW % 4 == 0, "gray block: at most two bytes per thread");
    __syncthreads();

    const bool xinner = sx0 - 9 >= 0 && sx0 + SW + 8 <= w - 1;   // P4: every column of the strip's b range has x-2 .. x+2 inside
    const bool xin1 = sx0 - 21 >= 0 && sx0 + SW + 20 <= w - 1;   // P1: every blur5 column of the strip lies inside the frame
    double sum0 = 0.0, sum1 = 0.0;                                // P6: cv2 ColumnSum's SUM of this thread's column (b, b*b)

    // P4: smaller Hessian eigenvalue at clamped coordinates (two np.gradient passes over G, one-sided at the border)
    auto eig = [&](double m00, double m01, double m11) -> double {
        const double t1 = m01 * m01;
        const double t2 = 4.0 * t1;
        const double t3 = m00 - m11;
        const double t4 = t3 * t3;
        const double t5 = t2 + t4;
        const double t7 = sqrt(t5) / 2.0;
        const double t9 = (m00 + m11) / 2.0;
        return t9 - t7;
    };
    // general form of one pixel: np.gradient is (f[i+1] - f[i-1]) / 2 inside and f[1] - f[0] / f[n-1] - f[n-2] at the ends,
    // i.e. (f[min(i+1, n-1)] - f[max(i-1, 0)]) * (0.5 or 1.0): the same operations as the interior form with clamped
    // indices and per-lane factors, so frame borders cost a few integer operations instead of a divergent path
    auto p4_general = [&](int y, int j) {
        if (y < 0 || y >= h) return;
        const int x = cpe::clampi(sx0 - 7 + j, 0, w - 1);
        auto cf = [](int v, int n) -> double { return (v == 0 || v == n - 1) ? 1.0 : 0.5; };
        auto lo = [](int v) -> int { return v > 0 ? v - 1 : 0; };
        auto hi = [](int v, int n) -> int { return v < n - 1 ? v + 1 : n - 1; };
        auto gat = [&](int yy, int xx) -> double { return s.g[((yy + 64) & (GR - 1)) * GST + (xx - sx0 + 9)]; };
        auto gx = [&](int yy, int xx) -> double { return (gat(yy, hi(xx, w)) - gat(yy, lo(xx))) * cf(xx, w); };
        auto gy = [&](int yy, int xx) -> double { return (gat(hi(yy, h), xx) - gat(lo(yy), xx)) * cf(yy, h); };
        const int xp = hi(x, w), xm = lo(x), yp = hi(y, h), ym = lo(y);
        const double cx = cf(x, w), cy = cf(y, h);
        const double m00 = (gx(y, xp) - gx(y, xm)) * cx;
        const double m01 = (gx(yp, x) - gx(ym, x)) * cy;
        const double m11 = (gy(yp, x) - gy(ym, x)) * cy;
        s.e[(y % ER) * EST + j] = eig(m00, m01, m11);
    };

    const int nsteps = (h + L6 - K + K - 1) / K + 1;
#ifdef CPE_PRE_STAMPS
    unsigned long long tacc[5] = {0, 0, 0, 0, 0}, ta, tb, tc, tm, td, te;
#endif
    for (int st = 0; st < nsteps; st++) {
        const int base = st * K;
        // the per-thread LDS offsets are recomputed every step: hoisted out of the loop they are ~40 live registers, which spill
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63;
        uint8_t gb0 = 0, gb1 = 0;
        if constexpr (!FAST) gray_load_bytes(base - L0, tid, gb0, gb1);
        STAMP(ta);
        // ------------------------------------------------------------------ interval A
        if (wave == 0) {
            // P5: cv2 RowSum of b (plane 0) and b*b (plane 1): a direct 15-term sum for the first output of an
            // 8-column block (x a multiple of 8), then s += in - out.  Lanes: row fastest, then block.
            const int ry = lane & 3, gq = lane >> 2;
            const int y = base - L5 + ry;
            if (y >= 0 && y < h) {
                const double *eb = &s.e[(y % ER) * EST + 8 * gq];
                double cv[22];
#pragma unroll
                for (int k = 0; k < 22; k++) cv[k] = eb[k];
#pragma unroll
                for (int plane = 0; plane < 2; plane++) {
                    double *rp = &s.rs[plane * RPL + (y % RR) * RST + 8 * gq];
                    double acc = 0.0;
#pragma unroll
                    for (int j = 0; j < 15; j++) acc = acc + cv[j];
                    rp[0] = acc;
#pragma unroll
                    for (int o = 1; o < 8; o++) {
                        acc = acc + (cv[o + 14] - cv[o - 1]);
                        rp[o] = acc;
                    }
                    if (plane == 0) {
#pragma unroll
                        for (int k = 0; k < 22; k++) cv[k] = cv[k] * cv[k];
                    }
                }
            }
        } else if (wave < 3) {
            // P1: 5x5 binomial, exact integer: (sum + 128) >> 8; 0 outside the image.  blur5 column u reads gray columns
            // u .. u + 4; an item makes the 8 outputs u0 .. u0 + 7 of one row from 5 x 3 aligned dwords.
            const int it = (wave - 1) * 44 + lane;
            if (lane < 44) {
                const int ry = it / 22, u0 = (it - ry * 22) * 8;
                const int y = base - L1 + ry;
                const uint32_t *a32 = reinterpret_cast<const uint32_t *>(s.gray);
                uint32_t lo, hi;
                if (xin1 && base - L1 >= 0 && base - L1 + K - 1 < h) {
                    // every output of the step lies inside the frame: two 16-bit lanes per register (bytes 0, 2 and bytes
                    // 1, 3 of a dword), column sums <= 16 * 255, row sums <= 256 * 255 < 2^16: no carry between lanes
                    uint32_t E[3] = {0, 0, 0}, O[3] = {0, 0, 0};
#pragma unroll
                    for (int dy = 0; dy < 5; dy++) {
                        const int sl = (y - 2 + dy + 64) & (GYR - 1);
                        const uint32_t *qp = a32 + (sl * GYW + u0) / 4;
#pragma unroll
                        for (int q = 0; q < 3; q++) {
                            const uint32_t d = qp[q];
                            const uint32_t ev = d & 0x00ff00ffu, od = (d >> 8) & 0x00ff00ffu;
                            if (dy == 0 || dy == 4) { E[q] += ev; O[q] += od; }
                            else if (dy == 2) { E[q] += (ev << 2) + (ev << 1); O[q] += (od << 2) + (od << 1); }
                            else { E[q] += ev << 2; O[q] += od << 2; }
                        }
                    }
                    // columns c0 .. c11: E[q] = {c(4q), c(4q+2)}, O[q] = {c(4q+1), c(4q+3)}; X, Y: the pairs one column pair on
                    const uint32_t X0 = __builtin_amdgcn_alignbit(E[1], E[0], 16), Y0 = __builtin_amdgcn_alignbit(O[1], O[0], 16);
                    const uint32_t X1 = __builtin_amdgcn_alignbit(E[2], E[1], 16), Y1 = __builtin_amdgcn_alignbit(O[2], O[1], 16);
                    const uint32_t rnd = 0x00800080u;
                    const uint32_t t02 = E[0] + (O[0] << 2) + (X0 << 2) + (X0 << 1) + (Y0 << 2) + E[1] + rnd;   // outputs 0, 2
                    const uint32_t t13 = O[0] + (X0 << 2) + (Y0 << 2) + (Y0 << 1) + (E[1] << 2) + O[1] + rnd;   // outputs 1, 3
                    const uint32_t t46 = E[1] + (O[1] << 2) + (X1 << 2) + (X1 << 1) + (Y1 << 2) + E[2] + rnd;   // outputs 4, 6
                    const uint32_t t57 = O[1] + (X1 << 2) + (Y1 << 2) + (Y1 << 1) + (E[2] << 2) + O[2] + rnd;   // outputs 5, 7
                    lo = ((t02 >> 8) & 0x00ff00ffu) | (t13 & 0xff00ff00u);
                    hi = ((t46 >> 8) & 0x00ff00ffu) | (t57 & 0xff00ff00u);
                } else {
                    int col[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                    for (int dy = 0; dy < 5; dy++) {
                        const int ky = (dy == 0 || dy == 4) ? 1 : ((dy == 2) ? 6 : 4);
                        const int sl = (y - 2 + dy + 64) & (GYR - 1);
                        const uint32_t *qp = a32 + (sl * GYW + u0) / 4;
                        const uint32_t d0 = qp[0], d1 = qp[1], d2 = qp[2];
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            col[k] += ky * (int)((d0 >> (8 * k)) & 255);
                            col[4 + k] += ky * (int)((d1 >> (8 * k)) & 255);
                            col[8 + k] += ky * (int)((d2 >> (8 * k)) & 255);
                        }
                    }
                    const bool yin = y >= 0 && y < h;
                    lo = 0; hi = 0;
#pragma unroll
                    for (int o = 0; o < 8; o++) {
                        const int acc = col[o] + 4 * col[o + 1] + 6 * col[o + 2] + 4 * col[o + 3] + col[o + 4];
                        const int x = sx0 - 22 + u0 + o;
                        const uint32_t bv = (yin && x >= 0 && x < w) ? (uint32_t)((acc + 128) >> 8) : 0u;
                        if (o < 4) lo |= bv << (8 * o); else hi |= bv << (8 * (o - 4));
                    }
                }
                *reinterpret_cast<uint2 *>(&s.b5[b5slot(y) * B5W + u0]) = make_uint2(lo, hi);
            }
        } else {
            // P3: Gaussian along x, two neighbouring outputs per item from 26 inputs (13 x 16-byte reads)
            const int i = tid - 192;
            if (i < N3) {
                const int ry = i / (GW / 2), j0 = (i - ry * (GW / 2)) * 2;
                const int y = base - L3 + ry;
                const double2 *vp = reinterpret_cast<const double2 *>(&s.v[ry * VST + j0]);
                double cv[26];
#pragma unroll
                for (int k = 0; k < 13; k++) { const double2 d = vp[k]; cv[2 * k] = d.x; cv[2 * k + 1] = d.y; }
                double res[2];
#pragma unroll
                for (int o = 0; o < 2; o++) {
                    double tsum = cv[o + 12] * c_gw[0];
#pragma unroll
                    for (int j = 12; j >= 1; j--) {
                        const double sm = cv[o + 12 - j] + cv[o + 12 + j];
                        const double pr = sm * c_gw[j];
                        tsum = tsum + pr;
                    }
                    res[o] = tsum;
                }
                const int sl = (y + 64) & (GR - 1);
                *reinterpret_cast<double2 *>(&s.g[sl * GST + j0]) = make_double2(res[0], res[1]);
            }
        }
        STAMP(tb);
        __syncthreads();
        STAMP(tc);
        STAMP(tm);
        // ------------------------------------------------------------------ interval B
        if constexpr (FAST) gray_fetch(base - L0, lane);
        else gray_store_bytes(base - L0, tid, gb0, gb1);
        if (wave < 2) {
            // P6: column sums (cv2 ColumnSum, running down the whole frame), Sauvola threshold, compare, store
            const int x = sx0 + tid;
            const double *r0 = &s.rs[tid], *r1 = &s.rs[RPL + tid];
            const int y0 = base - L6;
            auto p6_row = [&](int op, int om, int oe) -> uint8_t {
                const double s0 = sum0 + r0[op], s1 = sum1 + r1[op];
                const double m = s0 * (1.0 / 225.0), msq = s1 * (1.0 / 225.0);
                sum0 = s0 - r0[om];
                sum1 = s1 - r1[om];
                double var = msq - m * m;
                if (var < 0) var = 0;
                const double sd = sqrt(var);
                const double T = m * (1 + 0.5 * ((sd / 128) - 1));
                const double bv = s.e[oe + tid + 7];
                return (bv > T) ? 0 : 255;
            };
            if (y0 >= 8 && y0 + K - 1 + 7 <= h - 1) {      // no clamping, no start of the column: straight-line code
                int sp = (y0 + 7) % RR, sm = (y0 - 7) % RR, se = y0 % ER;
                double m[K], var[K], sd[K], bv[K];
#pragma unroll
                for (int ry = 0; ry < K; ry++) {
                    const double s0 = sum0 + r0[sp * RST], s1 = sum1 + r1[sp * RST];
                    m[ry] = s0 * (1.0 / 225.0);
                    const double msq = s1 * (1.0 / 225.0);
                    sum0 = s0 - r0[sm * RST];
                    sum1 = s1 - r1[sm * RST];
                    double vv = msq - m[ry] * m[ry];
                    if (vv < 0) vv = 0;
                    var[ry] = vv;
                    bv[ry] = s.e[se * EST + tid + 7];
                    sp = sp + 1 == RR ? 0 : sp + 1; sm = sm + 1 == RR ? 0 : sm + 1; se = se + 1 == ER ? 0 : se + 1;
                }
                sqrt_n<K>(var, sd);
                if (x < w) {
#pragma unroll
                    for (int ry = 0; ry < K; ry++) {
                        const double T = m[ry] * (1 + 0.5 * ((sd[ry] / 128) - 1));
                        out[(size_t)(y0 + ry) * w + x] = (bv[ry] > T) ? 0 : 255;
                    }
                }
            } else {
                for (int ry = 0; ry < K; ry++) {
                    const int y = y0 + ry;
                    if (y < 0 || y >= h) continue;
                    if (y == 0) {   // sumCount == 0: SUM takes the first ksize - 1 rows (7 replicas of row 0, rows 0 .. 6)
                        sum0 = 0.0; sum1 = 0.0;
                        for (int j = -7; j < 7; j++) {
                            const int o = (cpe::clampi(j, 0, h - 1) % RR) * RST;
                            sum0 = sum0 + r0[o];
                            sum1 = sum1 + r1[o];
                        }
                    }
                    const uint8_t r = p6_row(((y + 7 < h ? y + 7 : h - 1) % RR) * RST, ((y - 7 > 0 ? y - 7 : 0) % RR) * RST, (y % ER) * EST);
                    if (x < w) out[(size_t)y * w + x] = r;
                }
            }
        } else if (wave < 5) {
            // P2: Gaussian along y, K outputs of one column from K + 24 inputs (7 aligned blocks of the blur5 ring)
            const int rx = tid - 128;
            if (rx < N2) {
                const int blk0 = b5slot(base - L2 - 12) >> 2;
                double cv[K + 24];
#pragma unroll
                for (int m = 0; m < 7; m++) {
                    const uint8_t *bp = &s.b5[((blk0 + m) & 7) * 4 * B5W + rx + 1];
#pragma unroll
                    for (int k = 0; k < 4; k++) cv[4 * m + k] = s.lut[bp[k * B5W]];
                }
#pragma unroll
                for (int o = 0; o < K; o++) {
                    double tsum = cv[o + 12] * c_gw[0];
#pragma unroll
                    for (int j = 12; j >= 1; j--) {
                        const double sm = cv[o + 12 - j] + cv[o + 12 + j];
                        const double pr = sm * c_gw[j];
                        tsum = tsum + pr;
                    }
                    s.v[o * VST + rx] = tsum;
                }
            }
        } else {
            // P4: in waves 5 and 6 a thread owns column j of the strip's b range and makes its K rows: 28 reads of the whole G
            // ring (8 rows: y0-2 .. y0+5) instead of 9 per pixel, ring offsets are scalars.  The 14 columns left over are
            // single pixels on 56 lanes of wave 7 (a third of the instructions of a column owner).
            const int j = tid - 320;
            const int y0 = base - L4;
            if (wave == 7) {
                const int rr = lane / 14, jj = 128 + lane - rr * 14;
                if (lane < 56) p4_general(y0 + rr, jj);
            } else {
                if (xinner && y0 - 2 >= 0 && y0 + K - 1 + 2 <= h - 1) {
                    const int s0 = (y0 - 2 + 64) & (GR - 1);
                    double gc[8], gl1[6], gr1[6], gl2[4], gr2[4];
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const double *gp = &s.g[((s0 + k) & (GR - 1)) * GST + j + 2];
                        gc[k] = gp[0];
                        if (k >= 1 && k <= 6) { gl1[k - 1] = gp[-1]; gr1[k - 1] = gp[1]; }
                        if (k >= 2 && k <= 5) { gl2[k - 2] = gp[-2]; gr2[k - 2] = gp[2]; }
                    }
                    double t5[K], t9[K], t7[K];
#pragma unroll
                    for (int r = 0; r < K; r++) {
                        const double c = gc[r + 2];
                        const double m00 = ((gr2[r] - c) / 2.0 - (c - gl2[r]) / 2.0) / 2.0;
                        const double m01 = ((gr1[r + 2] - gl1[r + 2]) / 2.0 - (gr1[r] - gl1[r]) / 2.0) / 2.0;
                        const double m11 = ((gc[r + 4] - c) / 2.0 - (c - gc[r]) / 2.0) / 2.0;
                        const double t1 = m01 * m01;
                        const double t2 = 4.0 * t1;
                        const double t3 = m00 - m11;
                        const double t4 = t3 * t3;
                        t5[r] = t2 + t4;
                        t9[r] = (m00 + m11) / 2.0;
                    }
                    sqrt_n<K>(t5, t7);
                    int se = (y0 + 4 * ER) % ER;
#pragma unroll
                    for (int r = 0; r < K; r++) {
                        s.e[se * EST + j] = t9[r] - t7[r] / 2.0;
                        se = se + 1 == ER ? 0 : se + 1;
                    }
                } else {
                    for (int r = 0; r < K; r++) p4_general(y0 + r, j);
                }
            }
        }
        STAMP(td);
        __syncthreads();
        STAMP(te);
#ifdef CPE_PRE_STAMPS
        tacc[0] += tb - ta; tacc[1] += tc - tb; tacc[2] += tm - tc; tacc[3] += td - tm; tacc[4] += te - td;
#endif
    }
#ifdef CPE_PRE_STAMPS
    if ((tid0 & 63) == 0)
        for (int k = 0; k < 5; k++) atomicAdd(&g_stamps[wave][k], tacc[k]);
#endif
}

__global__ __launch_bounds__(NT, 4) void k_preprocess(const uint8_t *__restrict__ gray, int h, int w, int strips,
                                                      uint8_t *__restrict__ mask)
{
    __shared__ Smem s;
    const int frame = blockIdx.x / strips, strip = blockIdx.x - frame * strips;
    const int sx0 = strip * SW;
    const uint8_t *img = gray + (size_t)frame * h * w;
    uint8_t *out = mask + (size_t)frame * h * w;
    const bool fast = ((w & 3) == 0) && ((((size_t)img) & 3) == 0) && sx0 - 24 >= 0 && sx0 - 24 + GYW <= w;
    if (fast) preprocess_strip<true>(s, img, h, w, sx0, out);
    else preprocess_strip<false>(s, img, h, w, sx0, out);
}

}  // namespace

extern "C" int32_t cpe_preprocess_batch(const uint8_t *gray, int32_t n, int32_t h, int32_t w,
                                        uint8_t *mask, void *stream)
{
    CPE_CHECK_ARG(gray && mask, "cpe_preprocess_batch: null pointer");
    CPE_CHECK_ARG(n >= 0 && h >= 8 && w >= 8, "cpe_preprocess_batch: need n>=0, h,w>=8 (got %d,%d,%d)", n, h, w);
    if (n == 0) return CPE_OK;
    const int strips = (w + SW - 1) / SW;
    CPE_CHECK_ARG((long long)n * strips < (1ll << 31), "cpe_preprocess_batch: too many strips (%d x %d)", n, strips);
    CPE_LAUNCH_BEGIN();
    CPE_KLAUNCH(k_preprocess, dim3((unsigned)(n * strips)), dim3(NT), 0, (hipStream_t)stream, gray, h, w, strips, mask);
    CPE_CHECK_LAUNCH("k_preprocess");
    return CPE_OK;
}
