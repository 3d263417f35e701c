// Stages a-2, a-4, a-5, a-6 on the GPU:
//   extract_joints (util_cylinder.py:1805-1827), the joint filter of find_cylinder_centroids_and_center
//   (:1913-1920), mask_roi_around_center (:1944-2007), expands_line_roi / expand_line_roi /
//   process_contour_info / create_rotated_line_kernel / get_pca_endpoints (:35-237).
// [ext] OpenCV / numpy-LAPACK semantics restated exactly as in oracle/src/orc_morph.c, orc_masks.c,
// orc_contours.c, orc_draw.c (those are the bit-level specification; parity unpinned vs cv2).
//
// The reference's dominant cost -- two full-frame dilations with a ~100x100 structuring element per
// line fragment -- becomes one workgroup per fragment end point working on the (15 + K)^2 support in LDS.
#include "cpe_dev.h"

namespace cpe {

int ccl_run(const uint8_t *img, int n, int h, int w, int thr, int invert, int conn8, int *L, int *roots, bool holes_only,
            uint8_t *touch, int count_mode, int *cnt, int use_rect, int *nrect, FrameState *st, hipStream_t s, int sparse = 0, int flags = 0, int cnt_sel = 0);
int ccl_ctl(FrameState *st, int *nrect, int n, int h, int w, int op, hipStream_t s);

namespace {

// a-2 in one kernel: binary -> open(20x1) -> hmask, open(1x20) -> vmask, joints = hmask & vmask.
// Erosion and dilation both take the in-image pixels of the window [p-10, p+9] (anchor 10 of a 20-tap line; the border
// never erodes and never dilates).  The kernel works on one-bit rows: a workgroup packs a band of R + 40 rows (all
// columns) into 64-pixel words in LDS, with pixels outside the image set.  A horizontal opening is shifts of a
// 192-bit window (word and both neighbours) with the window doubled 1 -> 2 -> 4 -> 8 -> 16 (+4) taps; the vertical one is
// the same doubling across rows, word by word, ping-ponged between two LDS buffers.  ~2 bit operations per pixel instead
// of 80 byte reads; the masks leave as bytes, 8 pixels per store.
struct W3 { unsigned long long a, b, c; };   // pixels [-64, -1], [0, 63], [64, 127] relative to the word; bit i = pixel i
template <int K> __device__ __forceinline__ W3 w3_up(const W3 &v, unsigned long long fill)     // r[p] = v[p + K]
{
    W3 r;
    r.a = (v.a >> K) | (v.b << (64 - K));
    r.b = (v.b >> K) | (v.c << (64 - K));
    r.c = (v.c >> K) | (fill << (64 - K));
    return r;
}
template <int K> __device__ __forceinline__ W3 w3_down(const W3 &v, unsigned long long fill)   // r[p] = v[p - K]
{
    W3 r;
    r.c = (v.c << K) | (v.b >> (64 - K));
    r.b = (v.b << K) | (v.a >> (64 - K));
    r.a = (v.a << K) | (fill >> (64 - K));
    return r;
}
__device__ __forceinline__ W3 w3_and(const W3 &x, const W3 &y) { return W3{x.a & y.a, x.b & y.b, x.c & y.c}; }
__device__ __forceinline__ W3 w3_or(const W3 &x, const W3 &y) { return W3{x.a | y.a, x.b | y.b, x.c | y.c}; }
// r[p] = AND (IS_AND) / OR of v[p - 10 .. p + 9]; exact for p in [-54, 118]
template <bool IS_AND> __device__ __forceinline__ W3 w3_window20(const W3 &v)
{
    const unsigned long long f = IS_AND ? ~0ull : 0ull;
    auto op = [](const W3 &x, const W3 &y) { return IS_AND ? w3_and(x, y) : w3_or(x, y); };
    const W3 t1 = op(v, w3_up<1>(v, f));
    const W3 t2 = op(t1, w3_up<2>(t1, f));
    const W3 t4 = op(t2, w3_up<4>(t2, f));
    const W3 t8 = op(t4, w3_up<8>(t4, f));
    const W3 t20 = op(t8, w3_up<12>(t4, f));     // taps 0..15 and 12..19
    return w3_down<10>(t20, f);
}
__device__ __forceinline__ unsigned long long row_valid_word(int j, int w)   // in-image pixels of word j of a row
{
    const int x0 = j * 64;
    if (j < 0 || x0 >= w) return 0ull;
    return (x0 + 64 <= w) ? ~0ull : ((1ull << (w - x0)) - 1ull);
}
__device__ __forceinline__ unsigned long long bytes_of_bits8(unsigned bits)   // bit i -> byte i = 0 / 255
{
    unsigned long long v = ((unsigned long long)(bits & 255u) * 0x0101010101010101ull) & 0x8040201008040201ull;
    v = ((v + 0x7f7f7f7f7f7f7f7full) & 0x8080808080808080ull) >> 7;
    return v * 255ull;
}

constexpr int OB_AP = 20;
template <int R>
__global__ __launch_bounds__(256) void k_open20_joints(const uint8_t *__restrict__ bin, int h, int w, int bands,
                                                       uint8_t *__restrict__ hm, uint8_t *__restrict__ vm,
                                                       uint8_t *__restrict__ jm, uint32_t *__restrict__ jbits)
{
    // jbits: the joints mask once more as a one-bit plane (the words are in LDS anyway): its labelling, its RETR_EXTERNAL
    // flood and the border tracer of the joint centroids read an eighth of the bytes
    constexpr int ROWS = R + 2 * OB_AP;
    extern __shared__ unsigned long long s_ob[];
    const int WW = (w + 63) >> 6;
    unsigned long long *buf0 = s_ob, *buf1 = s_ob + (size_t)ROWS * WW, *hbuf = s_ob + (size_t)2 * ROWS * WW;
    const int t = threadIdx.x;
    const int f = blockIdx.x / bands, band = blockIdx.x - f * bands;
    const int y0 = band * R;
    const size_t N = (size_t)h * w;
    const uint8_t *im = bin + f * N;
    const bool al8 = ((w & 7) == 0) && (((size_t)im & 7) == 0);
    // pack: LDS byte k of row r = pixels [8k, 8k + 8) of row y0 - 20 + r; outside the image: ones
    {
        uint8_t *pk = reinterpret_cast<uint8_t *>(buf0);
        const int per_row = WW * 8;
        for (int i = t; i < ROWS * per_row; i += 256) {
            const int r = i / per_row, k = i - r * per_row;
            const int y = y0 - OB_AP + r, x0 = k * 8;
            unsigned bits = 255u;
            if (y >= 0 && y < h && x0 < w) {
                const uint8_t *q = im + (size_t)y * w + x0;
                if (al8) {    // w % 8 == 0: the 8 pixels are all inside
                    unsigned long long v = *reinterpret_cast<const unsigned long long *>(q);
                    v = (((v & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | v) & 0x8080808080808080ull;   // non-zero bytes
                    bits = (unsigned)(((v >> 7) * 0x0102040810204080ull) >> 56);
                } else {
                    bits = 0;
                    for (int b = 0; b < 8; b++) bits |= ((x0 + b >= w) || q[b]) ? (1u << b) : 0u;
                }
            }
            pk[i] = (uint8_t)bits;
        }
    }
    __syncthreads();
    // horizontal opening of the R output rows
    for (int i = t; i < R * WW; i += 256) {
        const int tr = i / WW, j = i - tr * WW;
        const unsigned long long *row = buf0 + (size_t)(tr + OB_AP) * WW;
        const W3 x{j > 0 ? row[j - 1] : ~0ull, row[j], j + 1 < WW ? row[j + 1] : ~0ull};
        const W3 v{row_valid_word(j - 1, w), row_valid_word(j, w), row_valid_word(j + 1, w)};
        const W3 e = w3_and(w3_window20<true>(x), v);
        hbuf[i] = w3_window20<false>(e).b & v.b;
    }
    __syncthreads();
    // vertical opening: r indexes rows of the band; a missing row is the neutral element of the chain
    auto pass = [&](const unsigned long long *src, unsigned long long *dst, int step, bool is_and, const unsigned long long *src2) {
        // dst[r] = src[r] (op) src2[r + step]
        const unsigned long long fill = is_and ? ~0ull : 0ull;
        for (int i = t; i < ROWS * WW; i += 256) {
            const int r = i / WW;
            const unsigned long long p = src[i];
            const unsigned long long q = (r + step < ROWS) ? src2[i + (size_t)step * WW] : fill;
            dst[i] = is_and ? (p & q) : (p | q);
        }
        __syncthreads();
    };
    pass(buf0, buf1, 1, true, buf0);     // 2 taps
    pass(buf1, buf0, 2, true, buf1);     // 4
    pass(buf0, buf1, 4, true, buf0);     // 8   (buf1)
    pass(buf1, buf0, 8, true, buf1);     // 16  (buf0)
    // E[r] = AND of rows r .. r+19 = erosion at row y0 - 10 + r; rows outside the image: 0
    for (int i = t; i < ROWS * WW; i += 256) {
        const int r = i / WW;
        const unsigned long long q = (r + 12 < ROWS) ? buf1[i + (size_t)12 * WW] : ~0ull;
        const int y = y0 - 10 + r;
        buf0[i] = (y >= 0 && y < h) ? (buf0[i] & q) : 0ull;
    }
    __syncthreads();
    pass(buf0, buf1, 1, false, buf0);
    pass(buf1, buf0, 2, false, buf1);
    pass(buf0, buf1, 4, false, buf0);    // 8   (buf1)
    pass(buf1, buf0, 8, false, buf1);    // 16  (buf0)
    // D[r] = OR of E[r .. r+19] = dilation at row y0 + r: only rows r < R are read below, each by one thread
    // outputs: 8 pixels per step
    {
        const int per_row = WW * 8;
        for (int i = t; i < R * per_row; i += 256) {
            const int tr = i / per_row, k = i - tr * per_row;
            const int y = y0 + tr, x0 = k * 8;
            if (y >= h || x0 >= w) continue;
            const int j = k >> 3, sh = (k & 7) * 8;
            const size_t wi = (size_t)tr * WW + j;
            const unsigned long long d16 = buf0[wi] | ((tr + 12 < ROWS) ? buf1[wi + (size_t)12 * WW] : 0ull);
            const unsigned hb = (unsigned)(hbuf[wi] >> sh) & 255u;
            const unsigned vb = (unsigned)(d16 >> sh) & 255u;
            const size_t o = f * N + (size_t)y * w + x0;
            if (al8 && (((size_t)hm | (size_t)vm | (size_t)jm) & 7) == 0) {
                *reinterpret_cast<unsigned long long *>(hm + o) = bytes_of_bits8(hb);
                *reinterpret_cast<unsigned long long *>(vm + o) = bytes_of_bits8(vb);
                *reinterpret_cast<unsigned long long *>(jm + o) = bytes_of_bits8(hb & vb);
            } else {
                for (int b = 0; b < 8 && x0 + b < w; b++) {
                    hm[o + b] = ((hb >> b) & 1u) ? 255 : 0;
                    vm[o + b] = ((vb >> b) & 1u) ? 255 : 0;
                    jm[o + b] = (((hb & vb) >> b) & 1u) ? 255 : 0;
                }
            }
        }
        // pixel x is bit x + 32 of its plane row: word 0 and the words behind the last pixel are zero
        const int ws = bit_row_words(w);
        for (int i = t; i < R * (WW + 1); i += 256) {
            const int tr = i / (WW + 1), j = i - tr * (WW + 1);
            const int y = y0 + tr;
            if (y >= h) continue;
            uint32_t *row = jbits + ((size_t)f * h + y) * ws;
            if (j == WW) {
                row[0] = 0;
                for (int k = 1 + 2 * WW; k < ws; k++) row[k] = 0;
                continue;
            }
            const size_t wi = (size_t)tr * WW + j;
            unsigned long long m = hbuf[wi] & (buf0[wi] | ((tr + 12 < ROWS) ? buf1[wi + (size_t)12 * WW] : 0ull));
            if (64 * j + 64 > w) m &= (1ull << (w - 64 * j)) - 1ull;      // columns past the end of the row
            row[1 + 2 * j] = (uint32_t)m;
            row[2 + 2 * j] = (uint32_t)(m >> 32);
        }
    }
}

// dst = ((a | b) != 0 ? 255 : 0) & c with c = mask_contour, which is zero outside the region rectangle: rows outside it are
// written as zeros without reading anything.  grid = (ceil(N / 16384), n), 16 bytes per thread and step when rows allow.
__global__ __launch_bounds__(256) void k_or_and(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b,
                                                const uint8_t *__restrict__ c, int h, int w, const FrameState *__restrict__ st,
                                                uint8_t *__restrict__ dst)
{
    const int N = h * w;
    const size_t f = blockIdx.y;
    const int *r = st[f].rect;
    const bool none = st[f].status == CPE_ST_NO_REGION;
    const int y0 = r[1], y1 = r[1] + r[3] - 1;
    const size_t o = f * (size_t)N;
    const bool vec = ((((size_t)a | (size_t)b | (size_t)c | (size_t)dst) & 15) == 0) && (N % 16 == 0);
    for (int it = 0; it < 4; it++) {
        const int i0 = blockIdx.x * 16384 + it * 4096 + threadIdx.x * 16;
        if (i0 >= N) break;
        const int ya = i0 / w, yb = min(i0 + 15, N - 1) / w;
        const bool outside = none || yb < y0 || ya > y1;
        if (vec) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (!outside) {
                const uint4 va = *reinterpret_cast<const uint4 *>(a + o + i0), vb = *reinterpret_cast<const uint4 *>(b + o + i0);
                const uint4 vc = *reinterpret_cast<const uint4 *>(c + o + i0);
                auto f4 = [](uint32_t x, uint32_t y, uint32_t z) {
                    uint32_t r4 = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t on = (((x | y) >> (8 * k)) & 255u) ? 255u : 0u;
                        r4 |= (on & ((z >> (8 * k)) & 255u)) << (8 * k);
                    }
                    return r4;
                };
                v = make_uint4(f4(va.x, vb.x, vc.x), f4(va.y, vb.y, vc.y), f4(va.z, vb.z, vc.z), f4(va.w, vb.w, vc.w));
            }
            *reinterpret_cast<uint4 *>(dst + o + i0) = v;
        } else {
            for (int k = 0; k < 16 && i0 + k < N; k++) {
                const size_t gi = o + i0 + k;
                dst[gi] = outside ? 0 : (uint8_t)(((a[gi] ? 255 : 0) | (b[gi] ? 255 : 0)) & c[gi]);
            }
        }
    }
}

// a-5 tail + a-6 head in one kernel:  roi = open3x3(mask & circle_mask & mask_contour)  (util_cylinder.py:1995-2005),
// base = close3x3(roi) (:150-152); out-of-image pixels never erode / dilate.  Bit rows like k_open20_joints: a workgroup
// holds a band of RB_R + 8 rows (all columns) as 64-pixel words in LDS and runs the four 3x3 passes on words (3 taps
// along the row by shifts, 3 rows by AND / OR).  mask_contour is zero outside the region rectangle, so only its pixels
// are read: everything else packs as zero, and bands further than 4 rows from it are written as zeros straight away.
constexpr int RB_R = 64, RB_AP = 4, RB_ROWS = RB_R + 2 * RB_AP;
__global__ __launch_bounds__(256) void k_roi_base(const uint8_t *__restrict__ m, const uint8_t *__restrict__ cm,
                                                  const uint8_t *__restrict__ mc, int h, int w, int bands,
                                                  const FrameState *__restrict__ st, uint8_t *__restrict__ roi,
                                                  uint8_t *__restrict__ base, uint32_t *__restrict__ bbits)
{
    // bbits: `base` once more as a one-bit plane (build_bitplanes layout, one plane per frame): the words are in LDS anyway,
    // and the fragments' labelling, flood and border tracer read the plane
    extern __shared__ unsigned long long s_rb[];
    const int WW = (w + 63) >> 6;
    unsigned long long *buf0 = s_rb, *buf1 = s_rb + (size_t)RB_ROWS * WW;
    const int t = threadIdx.x;
    const int f = blockIdx.x / bands, band = blockIdx.x - f * bands;
    const int y0 = band * RB_R;
    const size_t N = (size_t)h * w;
    const int *rc = st[f].rect;
    const int rx0 = rc[0], ry0 = rc[1], rx1 = rc[0] + rc[2] - 1, ry1 = rc[1] + rc[3] - 1;
    const bool al8 = ((w & 7) == 0) && ((((size_t)m | (size_t)cm | (size_t)mc | (size_t)roi | (size_t)base) & 7) == 0);
    const int per_row = WW * 8;
    auto store8 = [&](uint8_t *dst, size_t o, int x0, unsigned bits) {
        if (al8) *reinterpret_cast<unsigned long long *>(dst + o) = bytes_of_bits8(bits);
        else for (int b = 0; b < 8 && x0 + b < w; b++) dst[o + b] = ((bits >> b) & 1u) ? 255 : 0;
    };
    if (st[f].status == CPE_ST_NO_REGION || y0 > ry1 + RB_AP || y0 + RB_R - 1 < ry0 - RB_AP) {
        for (int i = t; i < RB_R * per_row; i += 256) {
            const int tr = i / per_row, k = i - tr * per_row;
            const int y = y0 + tr, x0 = k * 8;
            if (y >= h || x0 >= w) continue;
            const size_t o = f * N + (size_t)y * w + x0;
            store8(roi, o, x0, 0u); store8(base, o, x0, 0u);
        }
        const int ws = bit_row_words(w);
        for (int i = t; i < RB_R * ws; i += 256) {
            const int tr = i / ws, y = y0 + tr;
            if (y < h) bbits[((size_t)f * h + y) * ws + (i - tr * ws)] = 0u;
        }
        return;
    }
    // pack (m & cm & mc) != 0 of the rectangle's pixels; everything else (and outside the image) is zero
    {
        uint8_t *pk = reinterpret_cast<uint8_t *>(buf0);
        for (int i = t; i < RB_ROWS * per_row; i += 256) {
            const int r = i / per_row, k = i - r * per_row;
            const int y = y0 - RB_AP + r, x0 = k * 8;
            unsigned bits = 0u;
            if (y >= ry0 && y <= ry1 && y >= 0 && y < h && x0 < w && x0 + 7 >= rx0 && x0 <= rx1) {
                const size_t o = f * N + (size_t)y * w + x0;
                if (al8) {
                    unsigned long long v = *reinterpret_cast<const unsigned long long *>(m + o) &
                                           *reinterpret_cast<const unsigned long long *>(cm + o) &
                                           *reinterpret_cast<const unsigned long long *>(mc + o);
                    v = (((v & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | v) & 0x8080808080808080ull;
                    bits = (unsigned)(((v >> 7) * 0x0102040810204080ull) >> 56);
                } else {
                    for (int b = 0; b < 8 && x0 + b < w; b++) bits |= ((m[o + b] & cm[o + b]) & mc[o + b]) ? (1u << b) : 0u;
                }
            }
            pk[i] = (uint8_t)bits;
        }
    }
    __syncthreads();
    // one 3x3 pass on words; pixels outside the image (and rows / words outside the band) are the neutral element
    auto pass = [&](const unsigned long long *src, unsigned long long *dst, bool dil) {
        const unsigned long long fill = dil ? 0ull : ~0ull;
        for (int i = t; i < RB_ROWS * WW; i += 256) {
            const int r = i / WW, j = i - r * WW;
            unsigned long long acc = fill;
#pragma unroll
            for (int dr = -1; dr <= 1; dr++) {
                const int rr = r + dr, y = y0 - RB_AP + rr;
                unsigned long long hres = fill;
                if (rr >= 0 && rr < RB_ROWS && y >= 0 && y < h) {
                    const unsigned long long *row = src + (size_t)rr * WW;
                    const unsigned long long va = row_valid_word(j - 1, w), vb = row_valid_word(j, w), vc = row_valid_word(j + 1, w);
                    unsigned long long xa = j > 0 ? row[j - 1] : 0ull, xb = row[j], xc = j + 1 < WW ? row[j + 1] : 0ull;
                    if (!dil) { xa |= ~va; xb |= ~vb; xc |= ~vc; }
                    const unsigned long long up = (xb >> 1) | (xc << 63), down = (xb << 1) | (xa >> 63);
                    hres = dil ? (xb | up | down) : (xb & up & down);
                }
                acc = dil ? (acc | hres) : (acc & hres);
            }
            const int y = y0 - RB_AP + r;
            dst[i] = (y >= 0 && y < h) ? (acc & row_valid_word(j, w)) : 0ull;
        }
        __syncthreads();
    };
    auto emit = [&](const unsigned long long *src, uint8_t *dst) {
        for (int i = t; i < RB_R * per_row; i += 256) {
            const int tr = i / per_row, k = i - tr * per_row;
            const int y = y0 + tr, x0 = k * 8;
            if (y >= h || x0 >= w) continue;
            const unsigned bits = (unsigned)(src[(size_t)(tr + RB_AP) * WW + (k >> 3)] >> ((k & 7) * 8)) & 255u;
            store8(dst, f * N + (size_t)y * w + x0, x0, bits);
        }
    };
    pass(buf0, buf1, false);   // erode
    pass(buf1, buf0, true);    // dilate -> roi
    emit(buf0, roi);
    pass(buf0, buf1, true);    // dilate
    pass(buf1, buf0, false);   // erode -> base
    emit(buf0, base);
    {   // pixel x is bit x + 32 of its plane row: word 0 and the words behind the last pixel are zero
        const int ws = bit_row_words(w);
        for (int i = t; i < RB_R * (WW + 1); i += 256) {
            const int tr = i / (WW + 1), j = i - tr * (WW + 1);
            const int y = y0 + tr;
            if (y >= h) continue;
            uint32_t *row = bbits + ((size_t)f * h + y) * ws;
            if (j == WW) {
                row[0] = 0;
                for (int k = 1 + 2 * WW; k < ws; k++) row[k] = 0;
                continue;
            }
            const unsigned long long m64 = buf0[(size_t)(tr + RB_AP) * WW + j];     // (columns past the row's end are 0: row_valid_word)
            row[1 + 2 * j] = (uint32_t)m64;
            row[2 + 2 * j] = (uint32_t)(m64 >> 32);
        }
    }
}

// ---- joints: polygon-moment centroids inside the region rectangle, in cv2.findContours order ----------
__global__ __launch_bounds__(64) void k_joint_centroids(const uint32_t *__restrict__ jbits, int h, int w,
                                                        const int *__restrict__ roots, FrameState *__restrict__ st,
                                                        int *__restrict__ jtmp /* n*MAXJ*3 */,
                                                        const unsigned long long *__restrict__ outside, size_t plane_words)
{
    __shared__ unsigned long long s_win[BW_ROWS * 64];   // the tracer's bit window, one column per lane (cpe_dev.h: BitWin)
    const int f = blockIdx.y;
    if (st[f].status != CPE_ST_OK) return;
    const int ncomp = min(st[f].n_roots_p, MAXROOTS);
    for (int k = blockIdx.x * 64 + threadIdx.x; k < ncomp; k += gridDim.x * 64) {
        const int root = roots[(size_t)f * MAXROOTS + k];
        if (!comp_is_external(outside + f * plane_words, w, root, 0)) continue;   // RETR_EXTERNAL (:1817): inside a hole of another blob
        const int ws = bit_row_words(w);
        BitWin nz{jbits + (size_t)f * h * ws, ws, h, s_win + threadIdx.x};
        StatVisitor sv;
        if (!trace_border(nz, root % w, root / w, false, sv, 4 * (w + h) + 65536)) { set_overflow(st[f], OVF_TRACE); continue; }
        sv.finish();
        double m00, m10, m01;
        moments_from_sums(sv.a00, sv.a10, sv.a01, m00, m10, m01);
        if (m00 == 0) continue;
        int cx = (int)(m10 / m00), cy = (int)(m01 / m00);
        atomicAdd(&st[f].n_joints_all, 1);
        const int *r = st[f].rect;
        if (!(r[0] <= cx && cx < r[0] + r[2] && r[1] <= cy && cy < r[1] + r[3])) continue;
        int q = atomicAdd(&st[f].n_joints, 1);
        if (q >= MAXJ) { set_overflow(st[f], OVF_JOINTS); continue; }
        int *o = jtmp + ((size_t)f * MAXJ + q) * 3;
        o[0] = cx; o[1] = cy; o[2] = root;
    }
}

__global__ __launch_bounds__(256) void k_joint_sort(FrameState *__restrict__ st, const int *__restrict__ jtmp,
                                                    int *__restrict__ joints /* n*MAXJ*2 */)
{
    const int f = blockIdx.x;
    const int nj = min(st[f].n_joints, MAXJ);
    const int *src = jtmp + (size_t)f * MAXJ * 3;
    int *dst = joints + (size_t)f * MAXJ * 2;
    for (int i = threadIdx.x; i < nj; i += 256) {
        int ki = src[3 * i + 2], rank = 0;
        for (int j = 0; j < nj; j++) rank += (src[3 * j + 2] > ki) ? 1 : 0;  // latest discovery first
        dst[2 * rank] = src[3 * i];
        dst[2 * rank + 1] = src[3 * i + 1];
    }
    __syncthreads();
    if (threadIdx.x == 0) st[f].n_joints = nj;
}

// ---- separable fixed-point Gaussian blur on u8 (cv2.GaussianBlur (7,7) and (19,19), sigma 0) ------------
struct Taps { int k[19]; int r; int shift; };

// Both passes in one kernel: a 64x32 tile with an r-pixel apron goes through LDS, the u16 row sums never reach HBM.
// Neither consumer needs the whole plane, so tiles that cannot matter are skipped:
//   BLUR_SPOT (19x19): only `blurred > 240` is ever asked (util_cylinder.py:1958).  The taps sum to 2^16, so the
//     result is <= the largest input: a tile whose apron holds no pixel > 240 is written as 0.
//   BLUR_RECT (7x7): only the box means around the intersection points are read (find_center_point, :1548-1560), and
//     those points lie inside the region rectangle: tiles away from rect +- (half + 1) are left untouched.
constexpr int BT_X = 64, BT_Y = 32;
// one BT_X x BT_Y tile: both passes through LDS (workgroup-wide; ends with the stores, no barrier after them)
template <int R>
__device__ __forceinline__ void blur_tile(const uint8_t *__restrict__ im, int h, int w, int gx0, int gy0, const Taps &t,
                                          uint8_t *__restrict__ out, uint8_t *s_in, uint16_t *s_h)
{
    constexpr int IW = BT_X + 2 * R, IH = BT_Y + 2 * R;
    const int tid = threadIdx.x;
    for (int i = tid; i < IH * IW; i += 256) {
        int ry = i / IW, rx = i - ry * IW;
        s_in[i] = im[(size_t)reflect101(gy0 - R + ry, h) * w + reflect101(gx0 - R + rx, w)];
    }
    __syncthreads();
    for (int i = tid; i < IH * BT_X; i += 256) {
        int ry = i / BT_X, rx = i - ry * BT_X;
        const uint8_t *p = &s_in[ry * IW + rx];
        int sacc = 0;
#pragma unroll
        for (int j = 0; j <= 2 * R; j++) sacc += t.k[j] * p[j];
        s_h[i] = (uint16_t)sacc;
    }
    __syncthreads();
    for (int i = tid; i < BT_Y * BT_X; i += 256) {
        int ry = i / BT_X, rx = i - ry * BT_X;
        const uint16_t *p = &s_h[ry * BT_X + rx];
        int sacc = 0;
#pragma unroll
        for (int j = 0; j <= 2 * R; j++) sacc += t.k[j] * (int)p[j * BT_X];
        if (gy0 + ry < h && gx0 + rx < w)
            out[(size_t)(gy0 + ry) * w + gx0 + rx] = (uint8_t)((sacc + (1 << (t.shift - 1))) >> t.shift);
    }
}

// BLUR_RECT (7 x 7): one workgroup per tile, tiles away from the region rectangle are left untouched
template <int R>
__global__ __launch_bounds__(256) void k_blur_fused(const uint8_t *__restrict__ src, int h, int w, int tiles_x, int tiles_y, Taps t,
                                                    const FrameState *__restrict__ st, uint8_t *__restrict__ dst)
{
    constexpr int IW = BT_X + 2 * R, IH = BT_Y + 2 * R;
    __shared__ uint8_t s_in[IH * IW];
    __shared__ uint16_t s_h[IH * BT_X];
    const int tiles = tiles_x * tiles_y;
    const int f = blockIdx.x / tiles, tt = blockIdx.x - f * tiles;
    const int gx0 = (tt % tiles_x) * BT_X, gy0 = (tt / tiles_x) * BT_Y;
    const size_t N = (size_t)h * w;
    const FrameState &S = st[f];
    if (S.status != CPE_ST_OK) return;
    int half = (int)(S.r0 / 5.0);
    if (half < 3) half = 3;
    if (half > 10) half = half + 5;
    half = max(half, (int)(S.r0 / 4.5));   // the planar script's window (util_plane.py:1280)
    const int m = half + 1;
    if (gx0 > S.rect[0] + S.rect[2] + m || gx0 + BT_X < S.rect[0] - m || gy0 > S.rect[1] + S.rect[3] + m ||
        gy0 + BT_Y < S.rect[1] - m)
        return;
    blur_tile<R>(src + f * N, h, w, gx0, gy0, t, dst + f * N, s_in, s_h);
}

// BLUR_SPOT (19 x 19) in two steps.  k_spot_scan reads the frame once (16-byte loads, a band of BT_Y rows per workgroup) and
// flags the tiles that hold a pixel > 240; their bounding box (+ 16 px) becomes st[].srect, the window of the spot
// labelling.  k_blur19_spot (again a band per workgroup) writes zeros over every tile with no flagged tile among its 3 x 3
// neighbours -- the taps sum to 2^16, so the blur is <= the largest input of its 19 x 19 window and cannot exceed 240
// there -- and looks at the others (blur19_tile_spot).  One read and one write of the frame instead of 1.6 reads through
// LDS per tile.  The blurred plane is only ever asked `> 240` (util_cylinder.py:1958): where that cannot hold it is 0.
__global__ __launch_bounds__(256) void k_spot_scan(const uint8_t *__restrict__ gray, int h, int w, int tiles_x, int tiles_y,
                                                   FrameState *__restrict__ st, uint8_t *__restrict__ flags, size_t flag_stride)
{
    __shared__ int s_f[128];
    const int f = blockIdx.x / tiles_y, ty = blockIdx.x - f * tiles_y, tid = threadIdx.x;
    const int gy0 = ty * BT_Y, rows = min(BT_Y, h - gy0);
    if (tid < 128) s_f[tid] = 0;
    __syncthreads();
    const uint8_t *im = gray + (size_t)f * h * w + (size_t)gy0 * w;
    if ((w & 15) == 0 && (((size_t)im) & 15) == 0) {
        const int cw = w >> 4;                     // the rows of a band are contiguous: chunk i = (row i / cw, 16 columns from 16 (i % cw))
        // a byte > 240 <=> (byte + 15) carries into bit 8 of its own 9-bit field: test the two byte pairs apart
        auto gt = [](uint32_t d) { return (((d & 0x00FF00FFu) + 0x000F000Fu) | (((d >> 8) & 0x00FF00FFu) + 0x000F000Fu)) & 0x01000100u; };
        for (int i = tid; i < rows * cw; i += 256) {
            const uint4 v = reinterpret_cast<const uint4 *>(im)[i];
            if (gt(v.x) | gt(v.y) | gt(v.z) | gt(v.w)) s_f[((i % cw) * 16) / BT_X] = 1;
        }
    } else {
        for (int i = tid; i < rows * w; i += 256)
            if (im[i] > 240) s_f[(i % w) / BT_X] = 1;
    }
    __syncthreads();
    uint8_t *fl = flags + f * flag_stride + (size_t)ty * tiles_x;
    for (int tx = tid; tx < tiles_x; tx += 256) {
        fl[tx] = (uint8_t)s_f[tx];
        if (s_f[tx]) {
            FrameState &S = st[f];
            atomicMin(&S.srect[0], max(tx * BT_X - 16, 0)); atomicMin(&S.srect[1], max(gy0 - 16, 0));
            atomicMax(&S.srect[2], min(tx * BT_X + BT_X + 15, w - 1)); atomicMax(&S.srect[3], min(gy0 + BT_Y + 15, h - 1));
        }
    }
}

// One live tile of the 19 x 19 blur.  Row sums first: a thread makes 8 neighbouring sums of a row from 8 dwords of the LDS
// window (v_alignbyte_b32 picks the four bytes that start at any byte, v_dot4_u32_u8 multiplies them with four taps: 10
// instructions per sum instead of 19 byte reads and 19 multiply-adds).  The column pass is a weighted mean of row sums
// (its taps add up to 256): if no row sum of the tile's rows exceeds 240 * 256 no blurred value exceeds 240, and the tile
// keeps the zeros it already holds -- the case of nearly every live tile (the dots where two laser lines cross are > 240
// but only ~5 px wide; it takes the saturated spot to lift a 19-tap mean over 240).  Returns after the stores, no barrier.
__device__ __forceinline__ void blur19_tile_spot(const uint8_t *__restrict__ im, int h, int w, int gx0, int gy0, const Taps &t,
                                                 uint8_t *__restrict__ out, uint32_t *s_in32, uint16_t *s_h, int *s_max)
{
    constexpr int R = 9, IH = BT_Y + 2 * R, QW = (BT_X + 24) / 4;   // LDS window: columns gx0 - 12 .. gx0 + BT_X + 11 (dword aligned)
    const int tid = threadIdx.x;
    const int ax0 = gx0 - 12, ay0 = gy0 - R;
    if (tid == 0) *s_max = 0;
    if (ax0 >= 0 && ax0 + 4 * QW <= w && ay0 >= 0 && ay0 + IH <= h && (w & 3) == 0 && (((size_t)im) & 3) == 0) {
        for (int i = tid; i < IH * QW; i += 256) {
            const int ry = i / QW, q = i - ry * QW;
            s_in32[i] = *reinterpret_cast<const uint32_t *>(im + (size_t)(ay0 + ry) * w + ax0 + 4 * q);
        }
    } else {
        uint8_t *s_in8 = reinterpret_cast<uint8_t *>(s_in32);
        for (int i = tid; i < IH * QW * 4; i += 256) {
            const int ry = i / (4 * QW), b = i - ry * 4 * QW;
            s_in8[i] = im[(size_t)reflect101(ay0 + ry, h) * w + reflect101(ax0 + b, w)];
        }
    }
    __syncthreads();
    uint32_t T[5];
#pragma unroll
    for (int q = 0; q < 5; q++)
        T[q] = (uint32_t)t.k[4 * q] | ((uint32_t)t.k[4 * q + 1] << 8) | ((uint32_t)t.k[4 * q + 2] << 16) | ((q < 4 ? (uint32_t)t.k[4 * q + 3] : 0u) << 24);
    int mx = 0;
    for (int it = tid; it < IH * (BT_X / 8); it += 256) {
        const int ry = it / (BT_X / 8), c0 = (it - ry * (BT_X / 8)) * 8;
        const uint32_t *dp = s_in32 + ry * QW + (c0 >> 2);   // tile column c is window byte c + 3 .. the sum of column c0 + o reads bytes c0 + o + 3 .. + 21
        uint32_t d[8];
#pragma unroll
        for (int k = 0; k < 8; k++) d[k] = dp[k];
#pragma unroll
        for (int o = 0; o < 8; o++) {
            const int sb = o + 3, idx = sb >> 2, sh = sb & 3;
            uint32_t acc = 0;
#pragma unroll
            for (int q = 0; q < 5; q++)
                acc = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d[idx + q + 1], d[idx + q], sh), T[q], acc, false);
            s_h[ry * BT_X + c0 + o] = (uint16_t)acc;
            mx = max(mx, (int)acc);
        }
    }
    for (int off = 32; off >= 1; off >>= 1) mx = max(mx, __shfl_xor(mx, off, 64));
    if ((tid & 63) == 0) atomicMax(s_max, mx);
    __syncthreads();
    if (*s_max <= 240 * 256) return;               // (uniform) every blurred value of the tile is <= 240: it stays 0
    for (int i = tid; i < BT_Y * BT_X; i += 256) {
        int ry = i / BT_X, rx = i - ry * BT_X;
        const uint16_t *p = &s_h[ry * BT_X + rx];
        int sacc = 0;
#pragma unroll
        for (int j = 0; j <= 2 * R; j++) sacc += t.k[j] * (int)p[j * BT_X];
        if (gy0 + ry < h && gx0 + rx < w)
            out[(size_t)(gy0 + ry) * w + gx0 + rx] = (uint8_t)((sacc + (1 << (t.shift - 1))) >> t.shift);
    }
}

__global__ __launch_bounds__(256) void k_blur19_spot(const uint8_t *__restrict__ src, int h, int w, int tiles_x, int tiles_y, Taps t,
                                                     const uint8_t *__restrict__ flags, size_t flag_stride, uint8_t *__restrict__ dst)
{
    constexpr int R = 9, IH = BT_Y + 2 * R, QW = (BT_X + 24) / 4;
    __shared__ uint32_t s_in32[IH * QW];
    __shared__ uint16_t s_h[IH * BT_X];
    __shared__ uint8_t s_live[128];
    __shared__ int s_max;
    const int f = blockIdx.x / tiles_y, ty = blockIdx.x - f * tiles_y, tid = threadIdx.x;
    const int gy0 = ty * BT_Y, rows = min(BT_Y, h - gy0);
    const size_t N = (size_t)h * w;
    const uint8_t *fl = flags + f * flag_stride;
    for (int tx = tid; tx < tiles_x; tx += 256) {
        int any = 0;
        for (int yy = max(ty - 1, 0); yy <= min(ty + 1, tiles_y - 1); yy++)
            for (int xx = max(tx - 1, 0); xx <= min(tx + 1, tiles_x - 1); xx++) any |= fl[(size_t)yy * tiles_x + xx];
        s_live[tx] = (uint8_t)any;
    }
    // zeros over the whole band; the few tiles whose blur can exceed 240 are written again below
    uint8_t *out = dst + f * N + (size_t)gy0 * w;
    if ((w & 15) == 0 && (((size_t)out) & 15) == 0) {
        for (int i = tid; i < rows * (w >> 4); i += 256) reinterpret_cast<uint4 *>(out)[i] = make_uint4(0, 0, 0, 0);
    } else {
        for (int i = tid; i < rows * w; i += 256) out[i] = 0;
    }
    __syncthreads();                               // (waits for this wavefront's zero stores: the stores below come after them)
    for (int tx = 0; tx < tiles_x; tx++) {
        if (!s_live[tx]) continue;                 // (uniform)
        blur19_tile_spot(src + f * N, h, w, tx * BT_X, gy0, t, dst + f * N, s_in32, s_h, &s_max);
        __syncthreads();
    }
}

// ---- saturated spot -> minEnclosingCircle -> ellipse erased from a 255 plane -----------------------
__global__ __launch_bounds__(64) void k_spot_area(const uint8_t *__restrict__ g19, int h, int w,
                                                  const int *__restrict__ roots, FrameState *__restrict__ st,
                                                  unsigned long long *__restrict__ best)
{
    const int f = blockIdx.y;
    const size_t N = (size_t)h * w;
    const int ncomp = min(st[f].n_roots_s, MAXROOTS);   // (runs beside the region stage: no look at st[].status here)
    for (int k = blockIdx.x * 64 + threadIdx.x; k < ncomp; k += gridDim.x * 64) {
        const int root = roots[(size_t)f * MAXROOTS + k];
        ThreshPred nz{g19 + f * N, w, h, 240};
        StatVisitor sv;
        if (!trace_border(nz, root % w, root / w, false, sv, 8 * (w + h) + (1 << 20))) { set_overflow(st[f], OVF_TRACE); continue; }
        sv.finish();
        long long a2 = sv.a00 < 0 ? -sv.a00 : sv.a00;
        // max(contours, key=contourArea): first maximum in list order = latest discovered among equals; area 0 counts
        unsigned long long key = ((unsigned long long)(a2 + 1) << 24) | (unsigned long long)(root & 0xFFFFFF);
        atomicMax(&best[f], key);
    }
}

struct VertVisitor {
    int *v;
    int cap, n = 0;
    __device__ __forceinline__ void point(int x, int y, bool vertex)
    {
        if (!vertex) return;
        if (n < cap) { v[2 * n] = x; v[2 * n + 1] = y; }
        n++;
    }
    __device__ __forceinline__ bool stop() const { return false; }
};

__device__ __forceinline__ float normf2(float x, float y) { return (float)sqrt((double)x * x + (double)y * y); }
constexpr float MEC_EPS = 1.0e-4f;

__device__ void circle3(const float *px, const float *py, float &cx_, float &cy_, float &r_)
{
    float v1x = px[1] - px[0], v1y = py[1] - py[0];
    float v2x = px[2] - px[0], v2y = py[2] - py[0];
    float m1x = (px[0] + px[1]) / 2.0f, m1y = (py[0] + py[1]) / 2.0f;
    float c1 = m1x * v1x + m1y * v1y;
    float m2x = (px[0] + px[2]) / 2.0f, m2y = (py[0] + py[2]) / 2.0f;
    float c2 = m2x * v2x + m2y * v2y;
    float det = v1x * v2y - v1y * v2x;
    if (fabsf(det) <= MEC_EPS) {
        float d1 = (px[0] - px[1]) * (px[0] - px[1]) + (py[0] - py[1]) * (py[0] - py[1]);
        float d2 = (px[0] - px[2]) * (px[0] - px[2]) + (py[0] - py[2]) * (py[0] - py[2]);
        float d3 = (px[1] - px[2]) * (px[1] - px[2]) + (py[1] - py[2]) * (py[1] - py[2]);
        float mx = d1 > d2 ? d1 : d2;
        mx = mx > d3 ? mx : d3;
        r_ = sqrtf(mx) * 0.5f + MEC_EPS;
        if (d1 >= d2 && d1 >= d3) { cx_ = (px[0] + px[1]) * 0.5f; cy_ = (py[0] + py[1]) * 0.5f; }
        else if (d2 >= d1 && d2 >= d3) { cx_ = (px[0] + px[2]) * 0.5f; cy_ = (py[0] + py[2]) * 0.5f; }
        else { cx_ = (px[1] + px[2]) * 0.5f; cy_ = (py[1] + py[2]) * 0.5f; }
        return;
    }
    float cx = (c1 * v2y - c2 * v1y) / det;
    float cy = (v1x * c2 - v2x * c1) / det;
    cx_ = cx; cy_ = cy;
    cx -= px[0]; cy -= py[0];
    r_ = (float)sqrt((double)(cx * cx + cy * cy)) + MEC_EPS;
}

__device__ void mec_third(const int *p, int i, int j, float &cx, float &cy, float &r)
{
    cx = (float)(p[2 * j] + p[2 * i]) / 2.0f;
    cy = (float)(p[2 * j + 1] + p[2 * i + 1]) / 2.0f;
    r = normf2((float)(p[2 * j] - p[2 * i]), (float)(p[2 * j + 1] - p[2 * i + 1])) / 2.0f + MEC_EPS;
    for (int k = 0; k < j; k++) {
        float dx = cx - (float)p[2 * k], dy = cy - (float)p[2 * k + 1];
        if (normf2(dx, dy) < r) continue;
        float fx[3] = {(float)p[2 * i], (float)p[2 * j], (float)p[2 * k]};
        float fy[3] = {(float)p[2 * i + 1], (float)p[2 * j + 1], (float)p[2 * k + 1]};
        float ncx, ncy, nr = 0;
        circle3(fx, fy, ncx, ncy, nr);
        if (nr > 0) { r = nr; cx = ncx; cy = ncy; }
    }
}
__device__ void mec_second(const int *p, int i, float &cx, float &cy, float &r)
{
    cx = (float)(p[0] + p[2 * i]) / 2.0f;
    cy = (float)(p[1] + p[2 * i + 1]) / 2.0f;
    r = normf2((float)(p[0] - p[2 * i]), (float)(p[1] - p[2 * i + 1])) / 2.0f + MEC_EPS;
    for (int j = 1; j < i; j++) {
        float dx = cx - (float)p[2 * j], dy = cy - (float)p[2 * j + 1];
        if (normf2(dx, dy) < r) continue;
        float ncx, ncy, nr = 0;
        mec_third(p, i, j, ncx, ncy, nr);
        if (nr > 0) { r = nr; cx = ncx; cy = ncy; }
    }
}
__device__ void min_enclosing_circle(const int *p, int n, float &cx, float &cy, float &r)
{
    cx = cy = r = 0;
    if (n == 0) return;
    if (n == 1) { cx = (float)p[0]; cy = (float)p[1]; r = MEC_EPS; return; }
    if (n == 2) {
        cx = ((float)p[0] + (float)p[2]) / 2.0f;
        cy = ((float)p[1] + (float)p[3]) / 2.0f;
        double dx = p[0] - p[2], dy = p[1] - p[3];
        r = (float)(sqrt(dx * dx + dy * dy) / 2.0) + MEC_EPS;
        return;
    }
    cx = (float)(p[0] + p[2]) / 2.0f;
    cy = (float)(p[1] + p[3]) / 2.0f;
    r = normf2((float)(p[0] - p[2]), (float)(p[1] - p[3])) / 2.0f + MEC_EPS;
    for (int i = 2; i < n; i++) {
        float dx = (float)p[2 * i] - cx, dy = (float)p[2 * i + 1] - cy;
        float d = normf2(dx, dy);
        if (d < r) continue;
        float ncx, ncy, nr = 0;
        mec_second(p, i, ncx, ncy, nr);
        if (nr > 0) { r = nr; cx = ncx; cy = ncy; }
    }
}

constexpr int XY_SHIFT = 16;
constexpr int XY_ONE = 1 << XY_SHIFT;

__device__ __forceinline__ void putz(uint8_t *img, int h, int w, int x, int y)
{
    if (x >= 0 && x < w && y >= 0 && y < h) img[(size_t)y * w + x] = 0;
}

__device__ void line2z(uint8_t *img, int h, int w, long long p1x, long long p1y, long long p2x, long long p2y)
{
    long long dx = p2x - p1x, dy = p2y - p1y;
    long long j = dx < 0 ? -1 : 0, ax = (dx ^ j) - j;
    long long i = dy < 0 ? -1 : 0, ay = (dy ^ i) - i;
    long long x_step, y_step;
    int ecount;
    if (ax > ay) {
        dy = (dy ^ j) - j;
        p1x ^= p2x & j; p2x ^= p1x & j; p1x ^= p2x & j;
        p1y ^= p2y & j; p2y ^= p1y & j; p1y ^= p2y & j;
        x_step = XY_ONE;
        y_step = dy * (1 << XY_SHIFT) / (ax | 1);
        ecount = (int)((p2x - p1x) >> XY_SHIFT);
    } else {
        dx = (dx ^ i) - i;
        p1x ^= p2x & i; p2x ^= p1x & i; p1x ^= p2x & i;
        p1y ^= p2y & i; p2y ^= p1y & i; p1y ^= p2y & i;
        x_step = dx * (1 << XY_SHIFT) / (ay | 1);
        y_step = XY_ONE;
        ecount = (int)((p2y - p1y) >> XY_SHIFT);
    }
    (void)x_step; (void)y_step;
    p1x += (XY_ONE >> 1);
    p1y += (XY_ONE >> 1);
    putz(img, h, w, (int)((p2x + (XY_ONE >> 1)) >> XY_SHIFT), (int)((p2y + (XY_ONE >> 1)) >> XY_SHIFT));
    if (ax > ay) {
        p1x >>= XY_SHIFT;
        while (ecount >= 0) {
            putz(img, h, w, (int)p1x, (int)(p1y >> XY_SHIFT));
            p1x++;
            p1y += y_step;
            ecount--;
        }
    } else {
        p1y >>= XY_SHIFT;
        while (ecount >= 0) {
            putz(img, h, w, (int)(p1x >> XY_SHIFT), (int)p1y);
            p1x += x_step;
            p1y++;
            ecount--;
        }
    }
}

// FillConvexPoly(shift = 16, colour 0), the scan-line half: sink(y, x1, x2) receives every row span (x1 <= x2, clipped to
// the image) in order of y.  OpenCV also draws the polygon's edges (Line2 from vertex to vertex): poly_edges_z below.
template <class Sink>
__device__ void convex_poly_rows(int h, int w, const long long *vx, const long long *vy, int npts, Sink sink)
{
    const int shift = XY_SHIFT;
    struct { int idx, di; long long x, dx; int ye; } edge[2];
    int delta = 1 << shift >> 1;
    int i, y, imin = 0, edges = npts;
    long long xmin, xmax, ymin, ymax;
    const int delta1 = XY_ONE >> 1, delta2 = XY_ONE >> 1;
    xmin = xmax = vx[0];
    ymin = ymax = vy[0];
    for (i = 0; i < npts; i++) {
        long long px = vx[i], py = vy[i];
        if (py < ymin) { ymin = py; imin = i; }
        if (py > ymax) ymax = py;
        if (px > xmax) xmax = px;
        if (px < xmin) xmin = px;
    }
    xmin = (xmin + delta) >> shift;
    xmax = (xmax + delta) >> shift;
    ymin = (ymin + delta) >> shift;
    ymax = (ymax + delta) >> shift;
    if (npts < 3 || (int)xmax < 0 || (int)ymax < 0 || (int)xmin >= w || (int)ymin >= h) return;
    if (ymax > h - 1) ymax = h - 1;
    edge[0].idx = edge[1].idx = imin;
    edge[0].ye = edge[1].ye = y = (int)ymin;
    edge[0].di = 1;
    edge[1].di = npts - 1;
    edge[0].x = edge[1].x = -XY_ONE;
    edge[0].dx = edge[1].dx = 0;
    do {
        for (i = 0; i < 2; i++) {
            if (y >= edge[i].ye) {
                int idx0 = edge[i].idx, di = edge[i].di;
                int idx = idx0 + di;
                if (idx >= npts) idx -= npts;
                int ty = 0;
                for (; edges-- > 0;) {
                    ty = (int)((vy[idx] + delta) >> shift);
                    if (ty > y) {
                        long long xs = vx[idx0], xe = vx[idx];
                        edge[i].ye = ty;
                        edge[i].dx = ((xe - xs) * 2 + (ty - y)) / (2 * (ty - y));
                        edge[i].x = xs;
                        edge[i].idx = idx;
                        break;
                    }
                    idx0 = idx;
                    idx += di;
                    if (idx >= npts) idx -= npts;
                }
            }
        }
        if (edges < 0) break;
        if (y >= 0) {
            int left = 0, right = 1;
            if (edge[0].x > edge[1].x) { left = 1; right = 0; }
            int xx1 = (int)((edge[left].x + delta1) >> XY_SHIFT);
            int xx2 = (int)((edge[right].x + delta2) >> XY_SHIFT);
            if (xx2 >= 0 && xx1 < w) {
                if (xx1 < 0) xx1 = 0;
                if (xx2 >= w) xx2 = w - 1;
                sink(y, xx1, xx2);
            }
        }
        edge[0].x += edge[0].dx;
        edge[1].x += edge[1].dx;
    } while (++y <= (int)ymax);
}

// cv2.circle(img, (cx, cy), radius, 0, thickness=-1): the midpoint spans of OpenCV's Circle() (as in k_discs)
__device__ void circle_fill_z(uint8_t *im, int h, int w, int cx, int cy, int radius)
{
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        int ys[4] = {cy - dy, cy + dy, cy - dx, cy + dx};
        int xa[4] = {cx - dx, cx - dx, cx - dy, cx - dy};
        int xb[4] = {cx + dx, cx + dx, cx + dy, cx + dy};
        for (int q = 0; q < 4; q++) {
            if (ys[q] < 0 || ys[q] >= h) continue;
            int x1 = max(xa[q], 0), x2 = min(xb[q], w - 1);
            for (int x = x1; x <= x2; x++) im[(size_t)ys[q] * w + x] = 0;
        }
        dy++;
        err += plus;
        plus += 2;
        int mask = (err <= 0) - 1;
        err -= minus & mask;
        dx += mask;
        minus -= mask & 2;
    }
}

// cv2.ellipse(img, (cx, cy), (a, b), 0, 0, 360, 0, -1): ellipse2Poly with the angle step OpenCV derives from the axes
// (at most 360 / 5 + 2 vertices), rounded to 16 fractional bits as EllipseEx does; the polygon is then filled
constexpr int ELL_MAXV = 80;
__device__ int ellipse_vertices(int cx, int cy, int a, int b, long long *vx, long long *vy)
{
    long long ccx = (long long)cx << XY_SHIFT, ccy = (long long)cy << XY_SHIFT;
    long long aw = llabs((long long)a << XY_SHIFT), ah = llabs((long long)b << XY_SHIFT);
    int delta = (int)(((aw > ah ? aw : ah) + (XY_ONE >> 1)) >> XY_SHIFT);
    delta = delta < 3 ? 90 : delta < 10 ? 30 : delta < 15 ? 18 : 5;
    int nv = 0;
    long long prevx = -1, prevy = -1;
    bool have_prev = false;
    for (int i = 0; i < 360 + delta; i += delta) {
        int ang = i > 360 ? 360 : i;
        double sx = (double)(float)sin((450 - ang) * 0.017453292519943295769236907684886);
        double sy = (double)(float)sin(ang * 0.017453292519943295769236907684886);
        if ((450 - ang) % 90 == 0) { int q = ((450 - ang) / 90) % 4; sx = q == 0 ? 0 : q == 1 ? 1 : q == 2 ? 0 : -1; }
        if (ang % 90 == 0) { int q = (ang / 90) % 4; sy = q == 0 ? 0 : q == 1 ? 1 : q == 2 ? 0 : -1; }
        double x = (double)aw * sx, y = (double)ah * sy;
        double ptx = (double)ccx + x, pty = (double)ccy + y;
        long long qx = (long long)(int)rint(ptx / XY_ONE) << XY_SHIFT, qy = (long long)(int)rint(pty / XY_ONE) << XY_SHIFT;
        qx += (int)rint(ptx - (double)qx);
        qy += (int)rint(pty - (double)qy);
        if (!have_prev || qx != prevx || qy != prevy) {
            vx[nv] = qx; vy[nv] = qy; nv++;
            prevx = qx; prevy = qy; have_prev = true;
        }
    }
    if (nv == 1) { vx[0] = vx[1] = ccx; vy[0] = vy[1] = ccy; nv = 2; }
    return nv;
}

// one wavefront per frame: contour of the largest saturated blob -> circle -> ellipse erased from cm (pre-set to 255).
// Lane 0 follows the border, runs minEnclosingCircle and makes the ellipse's vertices and row spans (a few hundred
// sequential steps, all in LDS: no per-lane scratch); the pixels -- the polygon's edges, one lane per edge, and its rows,
// 64 pixels per store -- are written by the whole wavefront.
constexpr int SPOT_ROWS = 1024;   // row spans kept in LDS; a taller ellipse is filled by lane 0 alone
__global__ __launch_bounds__(64) void k_spot_ellipse(const uint8_t *__restrict__ g19, int n, int h, int w,
                                                     const unsigned long long *__restrict__ best, FrameState *__restrict__ st,
                                                     int *__restrict__ verts /* n*MAXV*2 */, uint8_t *__restrict__ cm, int planar)
{
    const int f = blockIdx.x, lane = threadIdx.x;
    FrameState &S = st[f];
    __shared__ long long s_vx[ELL_MAXV], s_vy[ELL_MAXV];
    __shared__ int s_nv, s_rows, s_y0;
    __shared__ int2 s_span[SPOT_ROWS];
    const size_t N = (size_t)h * w;
    uint8_t *img = cm + f * N;
    if (lane == 0) {
        s_nv = 0; s_rows = 0; s_y0 = 0;
        if (best[f] == 0) S.spot_fail = 1;
        else {
            const int root = (int)(best[f] & 0xFFFFFF);
            ThreshPred nz{g19 + f * N, w, h, 240};
            VertVisitor vv{verts + (size_t)f * MAXV * 2, MAXV};
            trace_border(nz, root % w, root / w, false, vv, 8 * (w + h) + (1 << 20));
            if (vv.n > MAXV) { set_overflow(S, OVF_VERTS); vv.n = MAXV; }
            float cx, cy, rad;
            min_enclosing_circle(vv.v, vv.n, cx, cy, rad);
            int icx = (int)cx, icy = (int)cy;
            int cr0 = (int)rad;
            int cr = rad < 30 ? cr0 + 20 : cr0 + 5;
            int minor = cr + 20 > 1 ? cr + 20 : 1;
            int a = (int)rint((cr + 40) / 2.0), b = (int)rint(minor / 2.0);
            if (planar) { a = cr0; b = cr0; circle_fill_z(img, h, w, icx, icy, cr0); }   // util_plane.py:2733-2792: plain circle
            else {
                const int nv = ellipse_vertices(icx, icy, a, b, s_vx, s_vy);
                s_nv = nv;
                int rows = 0, y0 = 0;
                convex_poly_rows(h, w, s_vx, s_vy, nv, [&](int y, int x1, int x2) {
                    if (rows == 0) y0 = y;
                    if (rows < SPOT_ROWS) s_span[rows] = make_int2(x1, x2);
                    else for (int x = x1; x <= x2; x++) img[(size_t)y * w + x] = 0;
                    rows++;
                });
                s_rows = rows < SPOT_ROWS ? rows : SPOT_ROWS; s_y0 = y0;
            }
            S.r0 = cr0;
            S.spot[0] = icx; S.spot[1] = icy; S.spot[2] = a; S.spot[3] = b;
        }
    }
    __syncthreads();
    const int nv = s_nv;
    for (int e = lane; e < nv; e += 64) {          // FillConvexPoly draws every edge (vertex e-1 -> vertex e) with Line2
        const int p = e == 0 ? nv - 1 : e - 1;
        line2z(img, h, w, s_vx[p], s_vy[p], s_vx[e], s_vy[e]);
    }
    const int rows = s_rows, y0 = s_y0;
    for (int r = 0; r < rows; r++) {
        const int2 sp = s_span[r];
        uint8_t *rowp = img + (size_t)(y0 + r) * w;
        for (int x = sp.x + lane; x <= sp.y; x += 64) rowp[x] = 0;
    }
}

// ---- line-fragment expansion ------------------------------------------------------------------------

__device__ __forceinline__ double sign1(double a, double b) { return b >= 0 ? fabs(a) : -fabs(a); }

// numpy.linalg.eig of a symmetric 2x2 as LAPACK dgeev (dlanv2) orders and signs it
__device__ void eig2_lapack(double a, double b, double c, double d, double *w, double *V)
{
    double cs, sn;
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
    double v1x = cs, v1y = sn;
    double x0 = (a != d) ? -b / (a - d) : 0.0, x1 = 1.0;
    double v2x = cs * x0 - sn * x1, v2y = sn * x0 + cs * x1;
    double n2 = sqrt(v2x * v2x + v2y * v2y);
    v2x /= n2; v2y /= n2;
    V[0] = v1x; V[1] = v2x; V[2] = v1y; V[3] = v2y;
}

struct SegVisitor {
    float *pts;  // maxv x 2
    int maxv;
    int n = 0;
    __device__ __forceinline__ void point(int x, int y, bool vertex)
    {
        if (!vertex) return;
        if (n < maxv) { pts[2 * n] = (float)x; pts[2 * n + 1] = (float)y; }
        n++;
    }
    // expand_line_roi skips contours with more than max_pixels vertices (util_cylinder.py:169): no need to finish those
    __device__ __forceinline__ bool stop() const { return n > maxv; }
};

// one thread per fragment: CHAIN_APPROX_SIMPLE vertices (MINV..MAXVS: 5..200 for the cylinder script, 8..700 for the planar
// one, expand_line_roi's min_pixels / max_pixels) -> PCA end points, angle, length
constexpr int SEG_LPW = 4;
template <int MINV, int MAXVS>
__global__ __launch_bounds__(64) void k_seg_trace(const uint32_t *__restrict__ base_bits, int h, int w, int which,
                                                  const int *__restrict__ roots, int cnt_sel, FrameState *__restrict__ st,
                                                  SegRec *__restrict__ segs /* n*MAXSEG */,
                                                  const unsigned long long *__restrict__ outside, size_t plane_words)
{
    const int f = blockIdx.y;
    if (st[f].status != CPE_ST_OK) return;
    __shared__ unsigned long long s_win[BW_ROWS * 64];
    __shared__ float s_pts[SEG_LPW][2 * MAXVS];
    const int ncomp = min(*root_counter(st[f], cnt_sel), MAXROOTS);
    // Few, long borders: a wavefront steps at the pace of its slowest lane, and with 64 borders in flight nearly every
    // step waits for some lane's window refill (one memory round trip).  SEG_LPW borders per wavefront keep most steps
    // LDS-only; the other lanes idle.
    if (threadIdx.x >= SEG_LPW) return;
    for (int k = blockIdx.x * SEG_LPW + threadIdx.x; k < ncomp; k += gridDim.x * SEG_LPW) {
    const int root = roots[(size_t)f * MAXROOTS + k];
    if (!comp_is_external(outside + f * plane_words, w, root, window_x0(st[f], 2))) continue;   // RETR_EXTERNAL (:161)
    const int ws = bit_row_words(w);
    BitWin nz{base_bits + (size_t)f * h * ws, ws, h, s_win + threadIdx.x};
    float *pts = s_pts[threadIdx.x];     // the border's vertices live in LDS: 1.6 KB (5.6 KB planar) per lane were scratch
    SegVisitor sv{pts, MAXVS};
    if (!trace_border(nz, root % w, root / w, false, sv, 8 * (w + h) + (1 << 20))) { set_overflow(st[f], OVF_TRACE); continue; }
    const int n = sv.n;
    if (n < MINV || n > MAXVS) continue;
    // get_pca_endpoints
    float mx = 0, my = 0;
    for (int i = 0; i < n; i++) { mx += pts[2 * i]; my += pts[2 * i + 1]; }
    mx = mx / (float)n; my = my / (float)n;
    double ax = 0, ay = 0;
    for (int i = 0; i < n; i++) { ax += (double)(pts[2 * i] - mx); ay += (double)(pts[2 * i + 1] - my); }
    ax /= n; ay /= n;
    double sxx = 0, sxy = 0, syy = 0;
    for (int i = 0; i < n; i++) {
        double dx = (double)(pts[2 * i] - mx) - ax, dy = (double)(pts[2 * i + 1] - my) - ay;
        sxx += dx * dx; sxy += dx * dy; syy += dy * dy;
    }
    double fct = 1.0 / (n - 1);
    sxx *= fct; sxy *= fct; syy *= fct;
    double wv[2], V[4];
    eig2_lapack(sxx, sxy, sxy, syy, wv, V);
    int kk = wv[1] > wv[0] ? 1 : 0;
    double ux = V[0 + kk], uy = V[2 + kk];
    int imin = 0, imax = 0;
    double pmin = 0, pmax = 0;
    for (int i = 0; i < n; i++) {
        double pr = (double)(pts[2 * i] - mx) * ux + (double)(pts[2 * i + 1] - my) * uy;
        if (i == 0 || pr < pmin) { pmin = pr; imin = i; }
        if (i == 0 || pr > pmax) { pmax = pr; imax = i; }
    }
    float p1x = pts[2 * imin], p1y = pts[2 * imin + 1], p2x = pts[2 * imax], p2y = pts[2 * imax + 1];
    float dx = p2x - p1x, dy = p2y - p1y;
    float length = (float)hypot((double)dx, (double)dy);
    if (length < 1e-8) continue;
    float at = (float)atan2((double)dy, (double)dx);
    float deg = at * (float)(180.0 / 3.14159265358979323846);
    int q = atomicAdd(&st[f].n_seg[which], 1);
    if (q >= MAXSEG) { set_overflow(st[f], OVF_SEGS); continue; }
    SegRec &r = segs[(size_t)f * MAXSEG + q];
    r.p1x = p1x; r.p1y = p1y; r.p2x = p2x; r.p2y = p2y; r.angle = -deg; r.len = length; r.valid = 1;
    }
}

// median angle (np.median of float32) and maximum length per frame
__global__ __launch_bounds__(256) void k_seg_global(FrameState *__restrict__ st, int which, const SegRec *__restrict__ segs)
{
    const int f = blockIdx.x, t = threadIdx.x;
    __shared__ float s_lo, s_hi;
    __shared__ unsigned int s_len;
    const int nv = min(st[f].n_seg[which], MAXSEG);
    const SegRec *S = segs + (size_t)f * MAXSEG;
    if (t == 0) { s_lo = 0; s_hi = 0; s_len = 0; }
    __syncthreads();
    const int k_hi = nv / 2, k_lo = (nv & 1) ? nv / 2 : nv / 2 - 1;
    for (int i = t; i < nv; i += 256) {
        float ai = S[i].angle;
        int rank = 0;
        for (int j = 0; j < nv; j++) {
            float aj = S[j].angle;
            rank += (aj < ai || (aj == ai && j < i)) ? 1 : 0;
        }
        if (rank == k_lo) s_lo = ai;
        if (rank == k_hi) s_hi = ai;
        atomicMax(&s_len, __float_as_uint(S[i].len));  // lengths are positive: uint order == float order
    }
    __syncthreads();
    if (t == 0) {
        st[f].gang[which] = (nv & 1) ? s_hi : (float)(((double)s_lo + (double)s_hi) / 2.0);
        st[f].glen[which] = __uint_as_float(s_len);
    }
}

// one workgroup per fragment end point: 15x15 patch of the mask dilated by the rotated line kernel
// (reflected, as cv2.dilate does), eroded 3x3, OR-ed into exp.  Works on the (15 + ks)^2 support in LDS.
constexpr int EXP_MAXKS = 176;       // cylinder script: kernel 91 + r0
constexpr int EXP_MAXKS_PLANE = 208;  // planar script: fixed 201
template <int MAXKS>
__global__ __launch_bounds__(256) void k_seg_expand(const uint8_t *__restrict__ base, int h, int w, int which,
                                                    FrameState *__restrict__ st, const SegRec *__restrict__ segs,
                                                    uint8_t *__restrict__ exp, int fixed_ks)
{
    constexpr int EXP_REG = 15 + MAXKS + 2;
    __shared__ uint8_t dil[EXP_REG * EXP_REG];
    __shared__ short koff[4096][2];
    __shared__ short ppix[225][2];
    __shared__ int s_nk, s_np;
    const int f = blockIdx.y, t = threadIdx.x;
    FrameState &S = st[f];
    if (S.status != CPE_ST_OK) return;
    const int njobs = 2 * min(S.n_seg[which], MAXSEG);
    for (int job = blockIdx.x; job < njobs; job += gridDim.x) {   // (fragment, end point) pairs in turns
    const int seg = job >> 1, e = job & 1;
    const SegRec r = segs[(size_t)f * MAXSEG + seg];
    const float glen = S.glen[which], gang = S.gang[which];
    if ((double)r.len > 0.8 * (double)glen) continue;
    const int ks = fixed_ks > 0 ? fixed_ks : 91 + S.r0;
    if (ks > MAXKS) { if (t == 0) set_overflow(S, OVF_KERNEL); return; }
    __syncthreads();   // the previous job's readers of dil / koff / ppix are done
    const float ak = fabsf(r.angle - gang) > 5.0f ? gang : r.angle;
    const int a = ks / 2, half = 7;
    const size_t N = (size_t)h * w;
    const uint8_t *bm = base + f * N;
    const int cx = (int)rint((double)(e ? r.p2x : r.p1x)), cy = (int)rint((double)(e ? r.p2y : r.p1y));
    const int x1 = max(cx - half, 0), x2 = min(cx + half + 1, w), y1 = max(cy - half, 0), y2 = min(cy + half + 1, h);
    if (t == 0) { s_nk = 0; s_np = 0; }
    __syncthreads();
    // rotated line kernel: getRotationMatrix2D + warpAffine(INTER_NEAREST) of the centre row
    {
        const int c = ks / 2;
        double ar = (double)ak * 3.1415926535897932384626433832795 / 180.0;
        double alpha = cos(ar), beta = sin(ar);
        double M[6] = {alpha, beta, (1 - alpha) * c - beta * c, -beta, alpha, beta * c + (1 - alpha) * c};
        double D = M[0] * M[4] - M[1] * M[3];
        D = D != 0 ? 1. / D : 0;
        double A11 = M[4] * D, A22 = M[0] * D;
        double iM[6];
        iM[0] = A11; iM[1] = M[1] * (-D); iM[3] = M[3] * (-D); iM[4] = A22;
        double b1 = -iM[0] * M[2] - iM[1] * M[5];
        double b2 = -iM[3] * M[2] - iM[4] * M[5];
        iM[2] = b1; iM[5] = b2;
        for (int i = t; i < ks * ks; i += 256) {
            int y = i / ks, x = i - y * ks;
            int X0 = (int)rint((iM[1] * y + iM[2]) * 1024) + 512;
            int Y0 = (int)rint((iM[4] * y + iM[5]) * 1024) + 512;
            int adelta = (int)rint(iM[0] * x * 1024), bdelta = (int)rint(iM[3] * x * 1024);
            int X = (X0 + adelta) >> 10, Y = (Y0 + bdelta) >> 10;
            if (X >= 0 && X < ks && Y == c) {
                int q = atomicAdd(&s_nk, 1);
                if (q < 4096) { koff[q][0] = (short)(x - a); koff[q][1] = (short)(y - a); }
            }
        }
    }
    for (int i = t; i < 225; i += 256) {
        int py = y1 + i / 15, px = x1 + i % 15;
        if (py < y2 && px < x2 && bm[(size_t)py * w + px]) {
            int q = atomicAdd(&s_np, 1);
            ppix[q][0] = (short)px; ppix[q][1] = (short)py;
        }
    }
    // support box (global coords) with a 1-px margin for the erosion
    const int bx1 = x1 - a - 1, by1 = y1 - a - 1;
    const int bw = (x2 - x1) + 2 * a + 2, bh = (y2 - y1) + 2 * a + 2;
    for (int i = t; i < bw * bh; i += 256) dil[i] = 0;
    __syncthreads();
    const int nk = min(s_nk, 4096), np = s_np;
    if (s_nk > 4096 && t == 0) set_overflow(S, OVF_EXPAND);
    for (int i = t; i < nk * np; i += 256) {
        int p = i / nk, k = i - p * nk;
        int xx = ppix[p][0] - koff[k][0], yy = ppix[p][1] - koff[k][1];
        if (xx >= 0 && xx < w && yy >= 0 && yy < h) dil[(yy - by1) * bw + (xx - bx1)] = 1;
    }
    __syncthreads();
    uint8_t *ex = exp + f * N;
    for (int i = t; i < bw * bh; i += 256) {
        int ly = i / bw, lx = i - ly * bw;
        int gy = by1 + ly, gx = bx1 + lx;
        if (gx < 0 || gx >= w || gy < 0 || gy >= h) continue;
        bool all = true;
        for (int dy = -1; dy <= 1 && all; dy++)
            for (int dx = -1; dx <= 1; dx++) {
                int yy = gy + dy, xx = gx + dx;
                if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
                int ly2 = ly + dy, lx2 = lx + dx;
                bool v = (ly2 >= 0 && ly2 < bh && lx2 >= 0 && lx2 < bw) ? dil[ly2 * bw + lx2] != 0 : false;
                if (!v) { all = false; break; }
            }
        if (all) ex[(size_t)gy * w + gx] = 255;
    }
    }
}

// the three chains meet: the region stage's verdict comes first (a frame without a region never looked for its spot
// in the sequential order: mask_roi_around_center is not reached, util_cylinder.py:1900 raises before)
__global__ void k_masks_reset(FrameState *st, int n)
{
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    FrameState &S = st[f];
    S.n_joints = 0; S.n_joints_all = 0; S.n_seg[0] = 0; S.n_seg[1] = 0;
    S.gang[0] = S.gang[1] = 0; S.glen[0] = S.glen[1] = 0;
    if (S.status == CPE_ST_NO_REGION) { S.r0 = 0; S.spot[0] = S.spot[1] = S.spot[2] = S.spot[3] = 0; }
    else if (S.status == CPE_ST_OK && S.spot_fail) S.status = CPE_ST_NO_SPOT;
}

inline unsigned grid1(size_t total) { return (unsigned)((total + 255) / 256); }

}  // namespace


// cv2.GaussianBlur(img,(7,7),0) for indexing_data (util_cylinder.py:1433)
// indexing_data on a colour frame (util_cylinder.py:1433-1435): cv2.GaussianBlur(img, (7, 7), 0) channel by channel, then
// cv2.cvtColor(BGR2GRAY) of the blurred image.  Same tiles and the same skip rule as the grey blur; bgr: u8[n,h,w,3].
__global__ __launch_bounds__(256) void k_blur7_bgr(const uint8_t *__restrict__ bgr, int h, int w, int tiles_x, int tiles_y, Taps t,
                                                   const FrameState *__restrict__ st, uint8_t *__restrict__ dst)
{
    constexpr int R = 3, IW = BT_X + 2 * R, IH = BT_Y + 2 * R, PER = BT_X * BT_Y / 256;
    __shared__ uint8_t s_in[IH * IW];
    __shared__ uint16_t s_h[IH * BT_X];
    const int tid = threadIdx.x;
    const int tiles = tiles_x * tiles_y;
    const int f = blockIdx.x / tiles, tt = blockIdx.x - f * tiles;
    const int gx0 = (tt % tiles_x) * BT_X, gy0 = (tt / tiles_x) * BT_Y;
    const size_t N = (size_t)h * w;
    const FrameState &S = st[f];
    if (S.status != CPE_ST_OK) return;
    int half = (int)(S.r0 / 5.0);
    if (half < 3) half = 3;
    if (half > 10) half = half + 5;
    half = max(half, (int)(S.r0 / 4.5));
    const int m = half + 1;
    if (gx0 > S.rect[0] + S.rect[2] + m || gx0 + BT_X < S.rect[0] - m || gy0 > S.rect[1] + S.rect[3] + m || gy0 + BT_Y < S.rect[1] - m)
        return;
    const uint8_t *im = bgr + f * N * 3;
    uint32_t ch[PER][3];
    for (int c = 0; c < 3; c++) {
        for (int i = tid; i < IH * IW; i += 256) {
            int ry = i / IW, rx = i - ry * IW;
            s_in[i] = im[((size_t)reflect101(gy0 - R + ry, h) * w + reflect101(gx0 - R + rx, w)) * 3 + c];
        }
        __syncthreads();
        for (int i = tid; i < IH * BT_X; i += 256) {
            int ry = i / BT_X, rx = i - ry * BT_X;
            const uint8_t *p = &s_in[ry * IW + rx];
            int sacc = 0;
#pragma unroll
            for (int j = 0; j <= 2 * R; j++) sacc += t.k[j] * p[j];
            s_h[i] = (uint16_t)sacc;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const int i = tid + q * 256;
            int ry = i / BT_X, rx = i - ry * BT_X;
            const uint16_t *p = &s_h[ry * BT_X + rx];
            int sacc = 0;
#pragma unroll
            for (int j = 0; j <= 2 * R; j++) sacc += t.k[j] * (int)p[j * BT_X];
            ch[q][c] = (uint32_t)((sacc + (1 << (t.shift - 1))) >> t.shift);
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < PER; q++) {
        const int i = tid + q * 256;
        int ry = i / BT_X, rx = i - ry * BT_X;
        if (gy0 + ry < h && gx0 + rx < w)
            dst[f * N + (size_t)(gy0 + ry) * w + gx0 + rx] = (uint8_t)((ch[q][0] * 3735u + ch[q][1] * 19235u + ch[q][2] * 9798u + 16384u) >> 15);
    }
}

int blur7_bgr(const uint8_t *bgr, int n, int h, int w, const FrameState *st, uint8_t *dst, hipStream_t s)
{
    Taps t7 = {{2, 7, 14, 18, 14, 7, 2}, 3, 12};
    CPE_LAUNCH_BEGIN();
    const int tiles_x = (w + BT_X - 1) / BT_X, tiles_y = (h + BT_Y - 1) / BT_Y;
    CPE_KLAUNCH(k_blur7_bgr, dim3((unsigned)(n * tiles_x * tiles_y)), dim3(256), 0, s, bgr, h, w, tiles_x, tiles_y, t7, st, dst);
    CPE_CHECK_LAUNCH("blur7_bgr");
    return CPE_OK;
}

int blur7_u8(const uint8_t *src, int n, int h, int w, const FrameState *st, uint8_t *dst, hipStream_t s)
{
    Taps t7 = {{2, 7, 14, 18, 14, 7, 2}, 3, 12};
    CPE_LAUNCH_BEGIN();
    const int tiles_x = (w + BT_X - 1) / BT_X, tiles_y = (h + BT_Y - 1) / BT_Y;
    CPE_KLAUNCH((k_blur_fused<3>), dim3((unsigned)(n * tiles_x * tiles_y)), dim3(256), 0, s, src, h, w, tiles_x, tiles_y, t7,
                st, dst);
    CPE_CHECK_LAUNCH("blur7_u8");
    return CPE_OK;
}

// a-2: hmask, vmask, joints mask and its components (no dependence on the region stage: own stream)
int joints_mask_stage(int n, int h, int w, const MaskBuffers &B, FrameState *st, hipStream_t s)
{
    CPE_LAUNCH_BEGIN();
    {
        // band height by LDS budget: (2 (R + 40) + R) words of 8 bytes per 64 columns
        const int WW = (w + 63) / 64;
        const size_t lds64 = (size_t)(2 * (64 + 2 * OB_AP) + 64) * WW * 8, lds32 = (size_t)(2 * (32 + 2 * OB_AP) + 32) * WW * 8;
        // more than the default 64 KB of dynamic LDS.  The attribute belongs to the (function, device) pair, so it is set on
        // every call for the device the launch goes to (cheap, and safe with several GPUs / host threads in one process)
        CPE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_open20_joints<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CPE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_open20_joints<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (lds64 <= 96 * 1024) {
            const int bands = (h + 63) / 64;
            CPE_KLAUNCH(k_open20_joints<64>, dim3((unsigned)(n * bands)), dim3(256), lds64, s, (const uint8_t *)B.binary, h, w, bands,
                        B.hmask, B.vmask, B.joints_mask, B.jbits);
        } else {
            CPE_CHECK_ARG(lds32 <= 160 * 1024, "joints_mask_stage: frame too wide (%d columns)", w);
            const int bands = (h + 31) / 32;
            CPE_KLAUNCH(k_open20_joints<32>, dim3((unsigned)(n * bands)), dim3(256), lds32, s, (const uint8_t *)B.binary, h, w, bands,
                        B.hmask, B.vmask, B.joints_mask, B.jbits);
        }
    }
    CPE_CHECK_LAUNCH("joints_mask_stage");
    int rc;
    rc = ccl_roots_bits(B.jbits, n, h, w, B.lab_p, B.roots_p, 0, st, s, 1);          // labelling on the one-bit plane
    if (rc == CPE_ERR_ARG) rc = ccl_run(B.joints_mask, n, h, w, 0, 0, 1, B.lab_p, B.roots_p, false, nullptr, 0, nullptr, 0, nullptr, st, s, 1, 1, 1);
    if (rc != CPE_OK) return rc;
    // RETR_EXTERNAL (:1817): outer background of the joints mask (whole frame), on this chain, beside the region stage
    const size_t fl_words = (size_t)h * ((w + 63) / 64);
    return outside_flood(B.joints_mask, n, h, w, st, 0, B.fl_j, B.fl_j + (size_t)n * fl_words, fl_words, s, B.jbits);
}

// a-5 head: saturated spot -> circle_mask, r0 (depends on the grey frame only: own stream)
int spot_stage(const uint8_t *gray, int n, int h, int w, const MaskBuffers &B, FrameState *st, hipStream_t s, int planar)
{
    const size_t total = (size_t)h * w * n;
    int rc;
    CPE_LAUNCH_BEGIN();
    // (B.best_s and st[].srect were reset by k_state_init)
    Taps t19 = {{1, 1, 3, 5, 10, 15, 20, 27, 30, 32, 30, 27, 20, 15, 10, 5, 3, 1, 1}, 9, 16};
    {
        const int tiles_x = (w + BT_X - 1) / BT_X, tiles_y = (h + BT_Y - 1) / BT_Y;
        CPE_CHECK_ARG(tiles_x <= 128 && (size_t)tiles_x * tiles_y <= (size_t)MAXROOTS * sizeof(int), "spot_stage: frame too large");
        // tile flags: in the root list of the spot labelling, which is written after the blur has read them
        uint8_t *flags = reinterpret_cast<uint8_t *>(B.roots_s);
        const size_t fstride = (size_t)MAXROOTS * sizeof(int);
        CPE_KLAUNCH(k_spot_scan, dim3((unsigned)(n * tiles_y)), dim3(256), 0, s, gray, h, w, tiles_x, tiles_y, st, flags, fstride);
        CPE_KLAUNCH(k_blur19_spot, dim3((unsigned)(n * tiles_y)), dim3(256), 0, s, gray, h, w, tiles_x, tiles_y, t19,
                    (const uint8_t *)flags, fstride, B.g19);
    }
    // labelling of blurred > 240 inside st[].srect (use_rect 3): the only place it can hold
    if ((rc = ccl_run(B.g19, n, h, w, 240, 0, 1, B.lab_s, B.roots_s, false, nullptr, 0, nullptr, 3, nullptr, st, s, 1, 1, 2)) != CPE_OK) return rc;
    CPE_KLAUNCH(k_spot_area, dim3(4, n), dim3(64), 0, s, B.g19, h, w, B.roots_s, st, B.best_s);
    (void)hipMemsetAsync(B.cm, 255, total, s);
    CPE_KLAUNCH(k_spot_ellipse, dim3(n), dim3(64), 0, s, B.g19, n, h, w, B.best_s, st, B.verts, B.cm, planar);
    CPE_CHECK_LAUNCH("spot_stage");
    return CPE_OK;
}

// a-2/a-4 (joint centroids inside rect), a-5 (roi masks), a-6 (expansion).  Needs st[].rect and mc, the joints
// components and the spot.
int masks_stage(const uint8_t *gray, int n, int h, int w, const MaskBuffers &B, FrameState *st, hipStream_t s,
                const RegionSide *side, int planar, hipStream_t sj)
{
    const size_t total = (size_t)h * w * n;
    int rc;
    CPE_LAUNCH_BEGIN();
    CPE_KLAUNCH(k_masks_reset, dim3((n + 63) / 64), dim3(64), 0, s, st, n);
    // scratch of the RETR_EXTERNAL tests: u64 planes carved from the one-bit planes the region stage is done with
    // (planes 0 and 1 hold the fragment masks below; a u64 plane of h * ceil(w/64) words fits one bit plane)
    const size_t bit_words = (size_t)n * h * bit_row_words(w);
    const size_t fl_words = (size_t)h * bit_row_words(w) / 2;          // u64 words per frame of one flood plane
    auto fl_plane = [&](int k) { return reinterpret_cast<unsigned long long *>(B.bits + (size_t)(2 + k) * ((bit_words + 1) & ~(size_t)1)); };
    // joints (their outer-background mask comes from the joints chain).  Only the lines kernel reads them: with helper
    // streams they are collected on `sj` beside the two fragment chains instead of in front of them
    const bool jside = side && sj != s;
    if (jside) { (void)hipEventRecord(side->clahe_done, s); (void)hipStreamWaitEvent(sj, side->clahe_done, 0); }
    {
        hipStream_t q = jside ? sj : s;
        CPE_KLAUNCH(k_joint_centroids, dim3(frame_waves(n, 16, MAXROOTS / 64), n), dim3(64), 0, q, (const uint32_t *)B.jbits, h, w, B.roots_p, st, B.jtmp,
                    (const unsigned long long *)(B.fl_j + (size_t)n * h * ((w + 63) / 64)), (size_t)h * ((w + 63) / 64));
        CPE_KLAUNCH(k_joint_sort, dim3(n), dim3(256), 0, q, st, B.jtmp, B.joints);
    }
    // a-5 tail, a-6 and the labelling of the expanded masks, once per line direction.  The two directions share
    // nothing but their inputs: the vertical one runs on the helper stream (if any) with the spot chain's label plane.
    const int rb_bands = (h + RB_R - 1) / RB_R;
    const size_t rb_lds = (size_t)2 * RB_ROWS * ((w + 63) / 64) * 8;
    CPE_CHECK_ARG(rb_lds <= 160 * 1024, "masks_stage: frame too wide (%d columns)", w);
    CPE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_roi_base), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if (side) { (void)hipEventRecord(side->clahe_done, s); (void)hipStreamWaitEvent(side->s, side->clahe_done, 0); }
    for (int which = 0; which < 2; which++) {
        hipStream_t q = (which && side) ? side->s : s;
        const uint8_t *lm = which ? B.vmask : B.hmask;
        uint8_t *roi = which ? B.roi_v : B.roi_h;
        uint8_t *base = which ? B.base_v : B.base_h;
        uint8_t *exp = which ? B.exp_v : B.exp_h;
        uint8_t *tmp = which ? B.tmpA : B.tmpB;
        int *lab = which ? B.lab_s : B.lab, *roots = which ? B.roots_s : B.roots;
        const int sel = which ? 2 : 0;
        uint32_t *bits = B.bits + (which ? bit_words : 0);
        SegRec *segs = B.segs + (size_t)which * n * MAXSEG;
        // roi = open3x3(mask & circle_mask & mask_contour), base = close3x3(roi)
        CPE_KLAUNCH(k_roi_base, dim3((unsigned)(n * rb_bands)), dim3(256), rb_lds, q, lm, (const uint8_t *)B.cm, (const uint8_t *)B.mc,
                    h, w, rb_bands, (const FrameState *)st, roi, base, bits);   // + base's one-bit plane
        rc = ccl_roots_bits(bits, n, h, w, lab, roots, 2, st, q, sel);         // labelling on the one-bit plane
        if (rc == CPE_ERR_ARG) rc = ccl_run(base, n, h, w, 0, 0, 1, lab, roots, false, nullptr, 0, nullptr, 2, nullptr, st, q, 1, 1, sel);
        if (rc != CPE_OK) return rc;
        unsigned long long *fl_bg = fl_plane(2 + 2 * which), *fl_out = fl_plane(3 + 2 * which);
        if ((rc = outside_flood(base, n, h, w, st, 2, fl_bg, fl_out, fl_words, q, bits)) != CPE_OK) return rc;
        if (planar) CPE_KLAUNCH((k_seg_trace<8, 700>), dim3(frame_waves(n, 32, 512), n), dim3(64), 0, q, (const uint32_t *)bits, h, w, which, (const int *)roots, sel, st, segs, (const unsigned long long *)fl_out, fl_words);
        else CPE_KLAUNCH((k_seg_trace<5, 200>), dim3(frame_waves(n, 32, 512), n), dim3(64), 0, q, (const uint32_t *)bits, h, w, which, (const int *)roots, sel, st, segs, (const unsigned long long *)fl_out, fl_words);
        CPE_KLAUNCH(k_seg_global, dim3(n), dim3(256), 0, q, st, which, (const SegRec *)segs);
        (void)hipMemsetAsync(tmp, 0, total, q);
        if (planar) CPE_KLAUNCH(k_seg_expand<EXP_MAXKS_PLANE>, dim3(frame_waves(n, 32, 256), n), dim3(256), 0, q, (const uint8_t *)base, h, w, which, st, (const SegRec *)segs, tmp, 201);
        else CPE_KLAUNCH(k_seg_expand<EXP_MAXKS>, dim3(frame_waves(n, 32, 256), n), dim3(256), 0, q, (const uint8_t *)base, h, w, which, st, (const SegRec *)segs, tmp, 0);
        CPE_KLAUNCH(k_or_and, dim3((unsigned)(((size_t)h * w + 16383) / 16384), n), dim3(256), 0, q, (const uint8_t *)tmp, (const uint8_t *)base,
                    (const uint8_t *)B.mc, h, w, (const FrameState *)st, exp);
        CPE_CHECK_LAUNCH("masks_stage expand");
        // cv2.connectedComponents of the expanded mask: unions only, k_lines resolves the joints' labels
        if ((rc = ccl_run(exp, n, h, w, 0, 0, 1, which ? B.lab_v : B.lab_h, nullptr, false, nullptr, 0, nullptr, 2, nullptr, st, q, 1, 2)) != CPE_OK)
            return rc;
    }
    if (side) { (void)hipEventRecord(side->traced, side->s); (void)hipStreamWaitEvent(s, side->traced, 0); }
    return CPE_OK;
}

}  // namespace cpe
