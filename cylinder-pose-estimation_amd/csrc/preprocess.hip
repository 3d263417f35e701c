// Stage a-1 -- fused pre-process kernel for gfx950.
//   reference: utils/util_cylinder.py:1769-1802 (load_and_preprocess_image), :1734-1738
//   (detect_ridges -> skimage hessian_matrix / hessian_matrix_eigvals), :1740-1766 (Sauvola).
//
// One workgroup (1024 threads = 16 waves, one per CU: the tile buffers take ~147 KB of the
// 160 KB LDS) produces a 64x64 tile of the binary ridge mask.  Everything between the u8 frame
// read and the u8 mask write lives in LDS:
//   A  gray   (TY+46)x(TX+48) u8   halo 23 = 2 (blur5) + 12 (Gauss) + 2 (gradients) + 7 (box), rows start 4-byte aligned
//   B  blur5  (TY+42)x(TX+42) u8   integer, exact, BORDER_REFLECT_101; 0 outside the image
//   C  V      (TY+18)x(TX+42) f64  Gaussian along y (scipy applies axis 0 first), zero padded
//   D  G      (TY+18)x(TX+18) f64  Gaussian along x
//   E  b      (TY+14)x(TX+14) f64  smaller Hessian eigenvalue, BORDER_REPLICATE for the box
//   F  rs     (TY+14)x TX     f64  15-tap row sums (first of b, then of b*b)
// The f64 operation order is the one scipy/numpy execute (centre tap, then symmetric pairs from
// the outside in; mul and add kept separate: built with -ffp-contract=off), so the mask is
// bit-identical to the CPU oracle, which in turn is bit-identical to the real skimage for b.
// HBM traffic per frame: h*w read (+ tile halo re-reads served by L2) + h*w written.
#include "cpe_internal.h"
#include <algorithm>

namespace {

constexpr int TX = 64, TY = 64, NT = 1024;
constexpr int RA = 23, RB = 21, RG = 9, RE = 7;
constexpr int AW = TX + 2 * RA + 2, AH = TY + 2 * RA;   // 112 x 110: column 0 is x = gx0 - 24 (a 4-byte boundary)
constexpr int BW = TX + 2 * RB, BH = TY + 2 * RB;   // 106
constexpr int VW = BW, VH = TY + 2 * RG;            // 106 x 82
constexpr int GW_ = TX + 2 * RG, GH = TY + 2 * RG;  // 82 x 82
constexpr int EW = TX + 2 * RE, EH = TY + 2 * RE;   // 78 x 78
// LDS row strides (in doubles) of b and of the row sums: with lanes = rows, an odd stride spreads a 32-lane group's 8-byte
// accesses over all 64 banks (stride 78 / 64 cost the row-sum phases 2x on reads and 8x on writes)
constexpr int EWP = EW + 1, RSP = TX + 1;
constexpr int VWP = VW + 1, GWP = GW_ + 1;          // the same for V and G (x-Gaussian: lanes = rows)

__constant__ double c_gw[13] = {
    0x1.105a329f98197p-3, 0x1.01a25f86eb137p-3, 0x1.b42a57d56c0bep-4,
    0x1.4a614d1afd337p-4, 0x1.bfde9c12bec92p-5, 0x1.0fa58939b5290p-5,
    0x1.26defcaeb0201p-6, 0x1.1e6bccad344bap-7, 0x1.f1e9915139407p-9,
    0x1.8345966f69519p-10, 0x1.0d8a5ad43c165p-11, 0x1.4fbe39149e277p-13,
    0x1.763a210dfb305p-15};

struct Smem {
    uint8_t a[AH * AW + 16];   // + slack: the last strip of phase B reads one dword past its row
    uint8_t b5[BH * BW];
    double buf1[VH * VWP];  // V, later b
    double buf2[GH * GWP];  // G, later row sums
};
static_assert(sizeof(Smem) <= 160 * 1024, "LDS budget");
static_assert(EH * EWP <= VH * VWP && EH * RSP <= GH * GWP, "buffer reuse");

// np.gradient of G (LDS tile, global coords) -- one-sided at the image border
struct GView {
    const double *g;
    int x0, y0, w, h;  // global coords of local (0,0); image size
    __device__ __forceinline__ double at(int y, int x) const { return g[(y - y0) * GWP + (x - x0)]; }
    __device__ __forceinline__ double gx(int y, int x) const
    {
        if (x == 0) return at(y, 1) - at(y, 0);
        if (x == w - 1) return at(y, w - 1) - at(y, w - 2);
        return (at(y, x + 1) - at(y, x - 1)) / 2.0;
    }
    __device__ __forceinline__ double gy(int y, int x) const
    {
        if (y == 0) return at(1, x) - at(0, x);
        if (y == h - 1) return at(h - 1, x) - at(h - 2, x);
        return (at(y + 1, x) - at(y - 1, x)) / 2.0;
    }
};

// A workgroup works through RUN consecutive tiles (same frame row mostly: neighbouring tiles share halo columns in L1 / L2).
// The gray window of the NEXT tile is requested from HBM while the current tile is in its f64 phases (the window buffer is
// free after phase B) and stored into LDS late in the iteration, so its latency hides behind the arithmetic instead of
// being waited for with one workgroup per CU and nothing else to run.  Runs are short on purpose: a grid of fully persistent
// workgroups (one per CU for the whole launch) was 8 % faster alone but starved the other chains of the call, which share
// the GPU with this kernel through the dispatcher's interleaving of workgroups.
constexpr int RUN = 8;
constexpr int A_DW = AH * (AW / 4);                 // dwords of the gray window
constexpr int A_PER = (A_DW + NT - 1) / NT;         // per thread
__global__ __launch_bounds__(NT) void k_preprocess(const uint8_t *__restrict__ gray, int h, int w,
                                                   int tiles_x, int tiles_y, long long total_tiles,
                                                   uint8_t *__restrict__ mask)
{
    __shared__ Smem s;
    const int tid = threadIdx.x;
    const int tiles = tiles_x * tiles_y;
    const int nslot = 1;
    const long long range_lo = (long long)blockIdx.x * RUN, slot = 0;
    const long long range_hi = range_lo + RUN < total_tiles ? range_lo + RUN : total_tiles;
    auto fast_tile = [&](long long tix, const uint8_t *&img_out, int &gx0_out, int &gy0_out) -> bool {
        const int frame = (int)(tix / tiles), t = (int)(tix - (long long)frame * tiles);
        gx0_out = (t % tiles_x) * TX; gy0_out = (t / tiles_x) * TY;
        img_out = gray + (size_t)frame * h * w;
        return ((w & 3) == 0) && ((((size_t)img_out) & 3) == 0) && gx0_out - 24 >= 0 && gx0_out - 24 + AW <= w &&
               gy0_out - RA >= 0 && gy0_out - RA + AH <= h;
    };
    bool have_a = false;                              // s.a already holds the window of the tile about to be processed
    for (long long tix = range_lo + slot; tix < range_hi; tix += nslot) {
    const int frame = (int)(tix / tiles);
    const int t = (int)(tix - (long long)frame * tiles);
    const int gx0 = (t % tiles_x) * TX, gy0 = (t / tiles_x) * TY;
    const uint8_t *img = gray + (size_t)frame * h * w;
    uint8_t *out = mask + (size_t)frame * h * w;

    // A: gray with reflect-101 addressing; LDS column j is x = gx0 - 24 + j.  Tiles whose window lies inside the
    // frame (and whose rows are 4-byte aligned in memory) move dwords, the others single bytes.
    if (!have_a) {
        const bool fast = ((w & 3) == 0) && ((((size_t)img) & 3) == 0) && gx0 - 24 >= 0 && gx0 - 24 + AW <= w &&
                          gy0 - RA >= 0 && gy0 - RA + AH <= h;
        if (fast) {
            const uint32_t *src = reinterpret_cast<const uint32_t *>(img + (size_t)(gy0 - RA) * w + (gx0 - 24));
            uint32_t *dst = reinterpret_cast<uint32_t *>(s.a);
            const int wq = w >> 2;
            for (int i = tid; i < A_DW; i += NT) {
                int ry = i / (AW / 4), q = i - ry * (AW / 4);
                dst[i] = src[(size_t)ry * wq + q];
            }
        } else {
            for (int i = tid; i < AH * AW; i += NT) {
                int ry = i / AW, rx = i - ry * AW;
                int y = cpe::reflect101(gy0 - RA + ry, h), x = cpe::reflect101(gx0 - 24 + rx, w);
                s.a[i] = img[(size_t)y * w + x];
            }
        }
    }
    __syncthreads();

    // B: 5x5 binomial, exact integer: (sum + 128) >> 8 ; 0 outside the image.
    // b5 column rx (x = gx0 - 21 + rx) reads LDS columns rx + 1 .. rx + 5.  A thread makes the 8 outputs
    // u0 .. u0 + 7 (u = rx + 1, u0 a multiple of 8) of one row from 5 x 3 aligned dwords.
    {
        constexpr int SB = (BW + 1 + 7) / 8;   // 14 strips per row
        const uint32_t *a32 = reinterpret_cast<const uint32_t *>(s.a);
        for (int i = tid; i < BH * SB; i += NT) {
            const int ry = i / SB, u0 = (i - ry * SB) * 8;
            const int y = gy0 - RB + ry;
            // separable (integers: any order is exact): the 12 column sums with weights 1 4 6 4 1 first, then the 8 outputs
            int col[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int dy = 0; dy < 5; dy++) {
                const int ky = (dy == 0 || dy == 4) ? 1 : ((dy == 2) ? 6 : 4);
                const uint32_t *q = a32 + ((ry + dy) * AW + u0) / 4;
                const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    col[k] += ky * (int)((d0 >> (8 * k)) & 255);
                    col[4 + k] += ky * (int)((d1 >> (8 * k)) & 255);
                    col[8 + k] += ky * (int)((d2 >> (8 * k)) & 255);
                }
            }
            int acc[8];
#pragma unroll
            for (int o = 0; o < 8; o++) acc[o] = col[o] + 4 * col[o + 1] + 6 * col[o + 2] + 4 * col[o + 3] + col[o + 4];
            const bool yin = y >= 0 && y < h;
#pragma unroll
            for (int o = 0; o < 8; o++) {
                const int rx = u0 + o - 1;
                if (rx >= 0 && rx < BW) {
                    const int x = gx0 - RB + rx;
                    s.b5[ry * BW + rx] = (uint8_t)((yin && x >= 0 && x < w) ? ((acc[o] + 128) >> 8) : 0);
                }
            }
        }
    }
    __syncthreads();
    // request the next tile's window (interior tiles only: the others are loaded at the top of their iteration)
    uint32_t pre[A_PER];
    bool pre_ok = false;
    {
        const long long nix = tix + nslot;
        if (nix < range_hi) {
            const uint8_t *nimg; int ngx0, ngy0;
            if (fast_tile(nix, nimg, ngx0, ngy0)) {
                pre_ok = true;
                const uint32_t *src = reinterpret_cast<const uint32_t *>(nimg + (size_t)(ngy0 - RA) * w + (ngx0 - 24));
                const int wq = w >> 2;
#pragma unroll
                for (int k = 0; k < A_PER; k++) {
                    int i = tid + k * NT;
                    asm volatile("" : "+v"(i));      // recompute the offsets per tile: hoisted out of the tile loop they
                                                     // cost 8 registers that spill (one scratch write per workgroup)
                    const int ry = i / (AW / 4), q = i - ry * (AW / 4);
                    pre[k] = i < A_DW ? src[(size_t)ry * wq + q] : 0u;
                }
            }
        }
    }

    const double inv255 = 1.0 / 255;
    // C: Gaussian along y.  V = 0 outside the image (the x pass zero-pads).
    // One thread = KC consecutive rows of one column: the KC + 24 inputs are read (and converted) once into registers
    // instead of 25 times; every output still adds its taps in scipy's order.
    {
        constexpr int KC = 10, GC = (VH + KC - 1) / KC;   // 9 row groups x 106 columns = 954 threads busy
        static_assert(GC * VW <= NT, "phase C fits one round");
        if (tid < GC * VW) {
            const int grp = tid / VW, rx = tid - grp * VW;
            const int ry0 = grp * KC;
            const int x = gx0 - RB + rx;
            double cv[KC + 24];
#pragma unroll
            for (int k = 0; k < KC + 24; k++) {
                const int r = ry0 + k;
                cv[k] = (r < BH) ? (double)s.b5[r * BW + rx] * inv255 : 0.0;
            }
#pragma unroll
            for (int o = 0; o < KC; o++) {
                const int ry = ry0 + o;
                if (ry < VH) {
                    const int y = gy0 - RG + ry;
                    double tsum = 0.0;
                    if (y >= 0 && y < h && x >= 0 && x < w) {
                        tsum = cv[o + 12] * c_gw[0];
#pragma unroll
                        for (int j = 12; j >= 1; j--) {
                            double sm = cv[o + 12 - j] + cv[o + 12 + j];
                            double pr = sm * c_gw[j];
                            tsum = tsum + pr;
                        }
                    }
                    s.buf1[ry * VWP + rx] = tsum;
                }
            }
        }
    }
    __syncthreads();

    // D: Gaussian along x: one thread = KD consecutive columns of one row
    {
        constexpr int KD = 7, GD = (GW_ + KD - 1) / KD;   // 12 column groups x 82 rows = 984 threads busy
        static_assert(GD * GH <= NT, "phase D fits one round");
        if (tid < GD * GH) {
            const int ry = tid % GH, grp = tid / GH;   // neighbouring lanes: neighbouring rows (odd strides: no bank conflicts)
            const int rx0 = grp * KD;
            double cv[KD + 24];
#pragma unroll
            for (int k = 0; k < KD + 24; k++) cv[k] = (rx0 + k < VW) ? s.buf1[ry * VWP + rx0 + k] : 0.0;
#pragma unroll
            for (int o = 0; o < KD; o++) {
                if (rx0 + o < GW_) {
                    double tsum = cv[o + 12] * c_gw[0];
#pragma unroll
                    for (int j = 12; j >= 1; j--) {
                        double sm = cv[o + 12 - j] + cv[o + 12 + j];
                        double pr = sm * c_gw[j];
                        tsum = tsum + pr;
                    }
                    s.buf2[ry * GWP + rx0 + o] = tsum;
                }
            }
        }
    }
    __syncthreads();

    // E: smaller Hessian eigenvalue at clamped coordinates (BORDER_REPLICATE of the box filter)
    {
        GView G{s.buf2, gx0 - RG, gy0 - RG, w, h};
        double *bb = s.buf1;
        // tiles whose E range keeps 2 pixels away from every frame border (uniform per workgroup): central differences
        // only, no clamping -- the same operations in the same order as the general path takes for such pixels
        const bool inner = gx0 - RE - 2 >= 0 && gx0 + TX + RE + 2 <= w - 1 && gy0 - RE - 2 >= 0 && gy0 + TY + RE + 2 <= h - 1;
        for (int i = tid; i < EH * EW; i += NT) {
            int ry = i / EW, rx = i - ry * EW;
            int y = cpe::clampi(gy0 - RE + ry, 0, h - 1), x = cpe::clampi(gx0 - RE + rx, 0, w - 1);
            double m00, m01, m11;
            if (inner) {
                const double *g = &s.buf2[(ry + RG - RE) * GWP + rx + RG - RE];   // G at (y, x)
                const double c = g[0];
                m00 = ((g[2] - c) / 2.0 - (c - g[-2]) / 2.0) / 2.0;
                m01 = ((g[GWP + 1] - g[GWP - 1]) / 2.0 - (g[-GWP + 1] - g[-GWP - 1]) / 2.0) / 2.0;
                m11 = ((g[2 * GWP] - c) / 2.0 - (c - g[-2 * GWP]) / 2.0) / 2.0;
            } else {
                if (x == 0) m00 = G.gx(y, 1) - G.gx(y, 0);
                else if (x == w - 1) m00 = G.gx(y, w - 1) - G.gx(y, w - 2);
                else m00 = (G.gx(y, x + 1) - G.gx(y, x - 1)) / 2.0;
                if (y == 0) {
                    m01 = G.gx(1, x) - G.gx(0, x);
                    m11 = G.gy(1, x) - G.gy(0, x);
                } else if (y == h - 1) {
                    m01 = G.gx(h - 1, x) - G.gx(h - 2, x);
                    m11 = G.gy(h - 1, x) - G.gy(h - 2, x);
                } else {
                    m01 = (G.gx(y + 1, x) - G.gx(y - 1, x)) / 2.0;
                    m11 = (G.gy(y + 1, x) - G.gy(y - 1, x)) / 2.0;
                }
            }
            double t1 = m01 * m01;
            double t2 = 4.0 * t1;
            double t3 = m00 - m11;
            double t4 = t3 * t3;
            double t5 = t2 + t4;
            double t7 = sqrt(t5) / 2.0;
            double t9 = (m00 + m11) / 2.0;
            bb[ry * EWP + rx] = t9 - t7;
        }
    }
    __syncthreads();
    if (pre_ok) {   // the window requested after phase B has arrived long ago
        uint32_t *dst = reinterpret_cast<uint32_t *>(s.a);
#pragma unroll
        for (int k = 0; k < A_PER; k++) { const int i = tid + k * NT; if (i < A_DW) dst[i] = pre[k]; }
    }
    have_a = pre_ok;

    // F/G: 15x15 box of b and b*b (row sums left->right, then column sums top->bottom).  Register windows again:
    // a thread makes KF neighbouring row sums from KF + 14 inputs, and the 4 column sums of its 4 output pixels
    // (same column, consecutive rows) from 18 row sums.
    const double *bb = s.buf1;
    double *rs = s.buf2;
    constexpr int KF = 8, GF = TX / KF;   // 8 groups x 78 rows = 624 threads busy
    constexpr int KG = TX * TY / NT;      // 4 output rows per thread
    const int otx = tid % TX, oty0 = (tid / TX) * KG;
    double mean[KG];
    auto row_sums = [&](bool squared) {
        if (tid < GF * EH) {
            const int ry = tid % EH, tx0 = (tid / EH) * KF;   // neighbouring lanes: neighbouring rows (see EWP)
            double cv[KF + 14];
#pragma unroll
            for (int k = 0; k < KF + 14; k++) {
                double v = bb[ry * EWP + tx0 + k];
                cv[k] = squared ? v * v : v;
            }
            static_assert(KF == 8 && TX % 8 == 0, "row-sum blocks are 8 columns, aligned to the image origin");
            // cv2's RowSum: a direct sum for the block's first output, then s += in - out (block = the KF outputs of
            // this thread: columns x0 .. x0 + 7 with x0 a multiple of 8 in image coordinates)
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < 15; j++) acc = acc + cv[j];
            rs[ry * RSP + tx0] = acc;
#pragma unroll
            for (int o = 1; o < KF; o++) {
                acc = acc + (cv[o + 14] - cv[o - 1]);
                rs[ry * RSP + tx0 + o] = acc;
            }
        }
    };
    auto col_sums = [&](double *res) {
        double cv[KG + 14];
#pragma unroll
        for (int k = 0; k < KG + 14; k++) cv[k] = rs[(oty0 + k) * RSP + otx];
        // cv2's ColumnSum: s0 = SUM + Sp, SUM = s0 - Sm (block = this thread's KG rows, y0 a multiple of 4)
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < 15; j++) acc = acc + cv[j];
        res[0] = acc * (1.0 / 225.0);
#pragma unroll
        for (int o = 1; o < KG; o++) {
            acc = (acc - cv[o - 1]) + cv[o + 14];
            res[o] = acc * (1.0 / 225.0);
        }
    };
    row_sums(false);
    __syncthreads();
    col_sums(mean);
    __syncthreads();
    row_sums(true);
    __syncthreads();
    double mean_sq[KG];
    col_sums(mean_sq);
#pragma unroll
    for (int o = 0; o < KG; o++) {
        const int ty = oty0 + o, tx = otx;
        const int y = gy0 + ty, x = gx0 + tx;
        double m = mean[o];
        double var = mean_sq[o] - m * m;
        if (var < 0) var = 0;
        double sd = sqrt(var);
        double T = m * (1 + 0.5 * ((sd / 128) - 1));
        double bv = bb[(ty + RE) * EWP + tx + RE];
        if (y < h && x < w) out[(size_t)y * w + x] = (bv > T) ? 0 : 255;
    }
    __syncthreads();   // the next tile's phases overwrite b5 / buf1 / buf2
    }
}

}  // namespace

extern "C" int32_t cpe_preprocess_batch(const uint8_t *gray, int32_t n, int32_t h, int32_t w,
                                        uint8_t *mask, void *stream)
{
    CPE_CHECK_ARG(gray && mask, "cpe_preprocess_batch: null pointer");
    CPE_CHECK_ARG(n >= 0 && h >= 8 && w >= 8, "cpe_preprocess_batch: need n>=0, h,w>=8 (got %d,%d,%d)", n, h, w);
    if (n == 0) return CPE_OK;
    int tiles_x = (w + TX - 1) / TX, tiles_y = (h + TY - 1) / TY;
    const long long blocks = (long long)n * tiles_x * tiles_y;
    const unsigned grid = (unsigned)((blocks + RUN - 1) / RUN);
    CPE_LAUNCH_BEGIN();
    CPE_KLAUNCH(k_preprocess, dim3(grid), dim3(NT), 0, (hipStream_t)stream, gray, h, w,
                       tiles_x, tiles_y, blocks, mask);
    CPE_CHECK_LAUNCH("k_preprocess");
    return CPE_OK;
}
