// Stage a-1 -- fused pre-process kernel for gfx950.
//   reference: utils/util_cylinder.py:1769-1802 (load_and_preprocess_image), :1734-1738
//   (detect_ridges -> skimage hessian_matrix / hessian_matrix_eigvals), :1740-1766 (Sauvola).
//
// One workgroup (1024 threads = 16 waves, one per CU: the tile buffers take ~147 KB of the
// 160 KB LDS) produces a 64x64 tile of the binary ridge mask.  Everything between the u8 frame
// read and the u8 mask write lives in LDS:
//   A  gray   (TY+46)x(TX+46) u8   halo 23 = 2 (blur5) + 12 (Gauss) + 2 (gradients) + 7 (box)
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

namespace {

constexpr int TX = 64, TY = 64, NT = 1024;
constexpr int RA = 23, RB = 21, RG = 9, RE = 7;
constexpr int AW = TX + 2 * RA, AH = TY + 2 * RA;   // 110
constexpr int BW = TX + 2 * RB, BH = TY + 2 * RB;   // 106
constexpr int VW = BW, VH = TY + 2 * RG;            // 106 x 82
constexpr int GW_ = TX + 2 * RG, GH = TY + 2 * RG;  // 82 x 82
constexpr int EW = TX + 2 * RE, EH = TY + 2 * RE;   // 78 x 78

__constant__ double c_gw[13] = {
    0x1.105a329f98197p-3, 0x1.01a25f86eb137p-3, 0x1.b42a57d56c0bep-4,
    0x1.4a614d1afd337p-4, 0x1.bfde9c12bec92p-5, 0x1.0fa58939b5290p-5,
    0x1.26defcaeb0201p-6, 0x1.1e6bccad344bap-7, 0x1.f1e9915139407p-9,
    0x1.8345966f69519p-10, 0x1.0d8a5ad43c165p-11, 0x1.4fbe39149e277p-13,
    0x1.763a210dfb305p-15};

struct Smem {
    uint8_t a[AH * AW];
    uint8_t b5[BH * BW];
    double buf1[VH * VW];  // V, later b
    double buf2[GH * GW_]; // G, later row sums
};
static_assert(sizeof(Smem) <= 160 * 1024, "LDS budget");
static_assert(EH * EW <= VH * VW && EH * TX <= GH * GW_, "buffer reuse");

// np.gradient of G (LDS tile, global coords) -- one-sided at the image border
struct GView {
    const double *g;
    int x0, y0, w, h;  // global coords of local (0,0); image size
    __device__ __forceinline__ double at(int y, int x) const { return g[(y - y0) * GW_ + (x - x0)]; }
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

__global__ __launch_bounds__(NT) void k_preprocess(const uint8_t *__restrict__ gray, int h, int w,
                                                   int tiles_x, int tiles_y,
                                                   uint8_t *__restrict__ mask)
{
    __shared__ Smem s;
    const int tid = threadIdx.x;
    // blockIdx.x -> (frame, tile); tiles of one frame are consecutive so they share L2 halos
    const int tiles = tiles_x * tiles_y;
    const int frame = blockIdx.x / tiles;
    const int t = blockIdx.x - frame * tiles;
    const int gx0 = (t % tiles_x) * TX, gy0 = (t / tiles_x) * TY;
    const uint8_t *img = gray + (size_t)frame * h * w;
    uint8_t *out = mask + (size_t)frame * h * w;

    // A: gray with reflect-101 addressing
    for (int i = tid; i < AH * AW; i += NT) {
        int ry = i / AW, rx = i - ry * AW;
        int y = cpe::reflect101(gy0 - RA + ry, h), x = cpe::reflect101(gx0 - RA + rx, w);
        s.a[i] = img[(size_t)y * w + x];
    }
    __syncthreads();

    // B: 5x5 binomial, exact integer: (sum + 128) >> 8 ; 0 outside the image
    for (int i = tid; i < BH * BW; i += NT) {
        int ry = i / BW, rx = i - ry * BW;
        int y = gy0 - RB + ry, x = gx0 - RB + rx;
        int v = 0;
        if (y >= 0 && y < h && x >= 0 && x < w) {
            const uint8_t *p = &s.a[ry * AW + rx];  // (ry+2-2, rx+2-2)
            int acc = 0;
#pragma unroll
            for (int dy = 0; dy < 5; dy++) {
                const int ky = (dy == 0 || dy == 4) ? 1 : ((dy == 2) ? 6 : 4);
                const uint8_t *q = p + dy * AW;
                acc += ky * (q[0] + 4 * q[1] + 6 * q[2] + 4 * q[3] + q[4]);
            }
            v = (acc + 128) >> 8;
        }
        s.b5[i] = (uint8_t)v;
    }
    __syncthreads();

    const double inv255 = 1.0 / 255;
    // C: Gaussian along y.  V = 0 outside the image (the x pass zero-pads).
    for (int i = tid; i < VH * VW; i += NT) {
        int ry = i / VW, rx = i - ry * VW;
        int y = gy0 - RG + ry, x = gx0 - RB + rx;
        double tsum = 0.0;
        if (y >= 0 && y < h && x >= 0 && x < w) {
            const uint8_t *p = &s.b5[(ry + 12) * BW + rx];  // b5 row of global y
            tsum = ((double)p[0] * inv255) * c_gw[0];
#pragma unroll
            for (int j = 12; j >= 1; j--) {
                double a = (double)p[-j * BW] * inv255;
                double b = (double)p[j * BW] * inv255;
                double sm = a + b;
                double pr = sm * c_gw[j];
                tsum = tsum + pr;
            }
        }
        s.buf1[i] = tsum;
    }
    __syncthreads();

    // D: Gaussian along x
    for (int i = tid; i < GH * GW_; i += NT) {
        int ry = i / GW_, rx = i - ry * GW_;
        const double *p = &s.buf1[ry * VW + rx + 12];
        double tsum = p[0] * c_gw[0];
#pragma unroll
        for (int j = 12; j >= 1; j--) {
            double sm = p[-j] + p[j];
            double pr = sm * c_gw[j];
            tsum = tsum + pr;
        }
        s.buf2[i] = tsum;
    }
    __syncthreads();

    // E: smaller Hessian eigenvalue at clamped coordinates (BORDER_REPLICATE of the box filter)
    {
        GView G{s.buf2, gx0 - RG, gy0 - RG, w, h};
        double *bb = s.buf1;
        for (int i = tid; i < EH * EW; i += NT) {
            int ry = i / EW, rx = i - ry * EW;
            int y = cpe::clampi(gy0 - RE + ry, 0, h - 1), x = cpe::clampi(gx0 - RE + rx, 0, w - 1);
            double m00, m01, m11;
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
            double t1 = m01 * m01;
            double t2 = 4.0 * t1;
            double t3 = m00 - m11;
            double t4 = t3 * t3;
            double t5 = t2 + t4;
            double t7 = sqrt(t5) / 2.0;
            double t9 = (m00 + m11) / 2.0;
            bb[i] = t9 - t7;
        }
    }
    __syncthreads();

    // F/G: 15x15 box of b and b*b (row sums left->right, then column sums top->bottom)
    const double *bb = s.buf1;
    double *rs = s.buf2;
    double mean[TX * TY / NT];
    for (int i = tid; i < EH * TX; i += NT) {
        int ry = i / TX, tx = i - ry * TX;
        const double *p = &bb[ry * EW + tx];
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < 15; j++) acc = acc + p[j];
        rs[i] = acc;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < TX * TY / NT; k++) {
        int i = tid + k * NT;
        int ty = i / TX, tx = i - ty * TX;
        const double *p = &rs[ty * TX + tx];
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < 15; j++) acc = acc + p[j * TX];
        mean[k] = acc * (1.0 / 225.0);
    }
    __syncthreads();
    for (int i = tid; i < EH * TX; i += NT) {
        int ry = i / TX, tx = i - ry * TX;
        const double *p = &bb[ry * EW + tx];
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < 15; j++) {
            double v = p[j];
            acc = acc + v * v;
        }
        rs[i] = acc;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < TX * TY / NT; k++) {
        int i = tid + k * NT;
        int ty = i / TX, tx = i - ty * TX;
        int y = gy0 + ty, x = gx0 + tx;
        const double *p = &rs[ty * TX + tx];
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < 15; j++) acc = acc + p[j * TX];
        double mean_sq = acc * (1.0 / 225.0);
        double m = mean[k];
        double var = mean_sq - m * m;
        if (var < 0) var = 0;
        double sd = sqrt(var);
        double T = m * (1 + 0.5 * ((sd / 128) - 1));
        double bv = bb[(ty + RE) * EW + tx + RE];
        if (y < h && x < w) out[(size_t)y * w + x] = (bv > T) ? 0 : 255;
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
    long long blocks = (long long)n * tiles_x * tiles_y;
    CPE_CHECK_ARG(blocks < (1LL << 31), "cpe_preprocess_batch: grid too large");
    CPE_LAUNCH_BEGIN();
    CPE_KLAUNCH(k_preprocess, dim3((unsigned)blocks), dim3(NT), 0, (hipStream_t)stream, gray, h, w,
                       tiles_x, tiles_y, mask);
    CPE_CHECK_LAUNCH("k_preprocess");
    return CPE_OK;
}
