// Connected-component labelling on the GPU (replaces cv2.connectedComponents, util_cylinder.py:28, and
// is the front end of every cv2.findContours call site: components -> one border trace each).
//
// Set membership: in(p) = (img[p] > thr) != invert.  Foreground sets use 8-connectivity, background
// sets 4-connectivity (the pairing under which Suzuki-Abe outer / hole borders are defined).
// Passes over an int32 label plane (label = raster index of the component's first pixel):
//   init   : one wavefront per image row; every pixel points at the first pixel of its horizontal run
//            (ballot + bit scan, carried across 64-pixel chunks)
//   merge  : runs are united with the row above through atomicMin union-find (only where a run
//            starts on either side, so a long run costs O(1) unions per neighbour run)
//   touch  : (hole search only) components that reach the border of the working rectangle are flagged
//   finish : flatten + per-component pixel counts (aggregated per wavefront before the atomic) + root list
//            + bounding box of the set, all in one read of the label plane
// When rows start on 16-byte boundaries, init (sparse passes), merge and the roots-only list have word-level forms
// (k_ccl_init64 / k_ccl_merge64 / k_ccl_roots64): 64 pixels per thread and row as one bit mask, the label plane is
// touched only where a run starts or two runs start to overlap.
// A pass can be restricted to a per-frame rectangle (FrameState::crect).  The blob detector uses this:
// every hole of the binarisation at threshold t lies inside the bounding box of the bright pixels at t,
// which lies inside the box at t-10; a dark pixel on the box border is 4-connected to the outside, so
// "touches the box" == "is not a hole".  HBM bytes per pixel inside the rectangle: 1 (image) + 4 written,
// 1 + sparse label traffic, 4 read + 4 written.
#include "cpe_dev.h"

namespace cpe {

namespace {

__device__ __forceinline__ bool pred(const uint8_t *img, size_t i, int thr, int invert)
{
    return (((int)img[i] > thr) ? 1 : 0) != invert;
}

struct Rect { int x0, y0, x1, y1; };
__device__ __forceinline__ Rect get_rect(const FrameState *st, size_t f, int use_rect, int h, int w)
{
    Rect r;
    if (use_rect == 1) { r.x0 = st[f].crect[0]; r.y0 = st[f].crect[1]; r.x1 = st[f].crect[2]; r.y1 = st[f].crect[3]; }
    else if (use_rect == 2) {   // region rectangle (boundingRect of the hull) + 2 px: holds every mask derived from mask_contour
        const int *q = st[f].rect;
        r.x0 = max(q[0] - 2, 0); r.y0 = max(q[1] - 2, 0); r.x1 = min(q[0] + q[2] + 1, w - 1); r.y1 = min(q[1] + q[3] + 1, h - 1);
    } else if (use_rect == 3) { r.x0 = st[f].srect[0]; r.y0 = st[f].srect[1]; r.x1 = st[f].srect[2]; r.y1 = st[f].srect[3]; }   // spot window (may be empty)
    else { r.x0 = 0; r.y0 = 0; r.x1 = w - 1; r.y1 = h - 1; }
    return r;
}

constexpr int CCL_INIT_ROWS = 16;
__global__ __launch_bounds__(256) void k_ccl_init(const uint8_t *__restrict__ img, int rows_total, int h, int w,
                                                  int thr, int invert, const FrameState *__restrict__ st, int use_rect,
                                                  int *__restrict__ L, int *__restrict__ cnt, int sparse)
{
    const int lane = threadIdx.x & 63;
    for (int rr = 0; rr < CCL_INIT_ROWS / 4; rr++) {   // CCL_INIT_ROWS rows per workgroup, one wavefront per row and turn
    const int row = (blockIdx.x * (CCL_INIT_ROWS / 4) + rr) * 4 + (threadIdx.x >> 6);
    if (row >= rows_total) return;
    const int f = row / h, y = row - f * h;
    const Rect r = get_rect(st, f, use_rect, h, w);
    if (y < r.y0 || y > r.y1 || r.x1 < r.x0) continue;
    const size_t base = (size_t)row * w;  // == frame * h*w + y*w
    int carry_in = 0, carry_start = 0;
    for (int x0 = r.x0; x0 <= r.x1; x0 += 64) {
        int x = x0 + lane;
        bool valid = x <= r.x1;
        bool in = valid && pred(img, base + x, thr, invert);
        unsigned long long b = __ballot(in);
        unsigned long long prev = (b << 1) | (unsigned long long)carry_in;
        unsigned long long starts = b & ~prev;
        if (cnt && valid) cnt[base + x] = 0;
        // sparse 3: first node of a pixel outside the set = the first pixel of its run of one grey-level bucket inside this
        // 64-pixel chunk (the run joins the dark forest of the blob sweep as one component; depth 1, like the run labels above)
        int pre = -1;
        if (sparse == 3) {
            const int lv = (valid && !in) ? sweep_level(img[base + x]) : 0;
            const int lvl_left = __shfl_up(lv, 1, 64);
            const bool same = lv > 0 && lane > 0 && lvl_left == lv;
            const unsigned long long st3 = __ballot(lv > 0 && !same);
            if (lv > 0) pre = y * w + x - (lane - (63 - __clzll((long long)(st3 & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull))))));
        }
        if (in) {
            unsigned long long m = starts & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
            int sx = m ? (x0 + 63 - __clzll(m)) : carry_start;
            L[base + x] = y * w + sx;
        } else if (valid && sparse != 1) {
            // sparse 1: labels outside the set are never read; 2: singletons (a growing set will absorb them later);
            // 3 (the dark forest of the blob sweep): singletons, except that the pixels of a run that joins the set at one
            // threshold (same grey-level bucket) start as children of the run's first pixel -- the run is one component the
            // moment it joins, and these stores are coalesced, which the per-bucket lists' are not
            L[base + x] = pre >= 0 ? pre : (sparse ? y * w + x : -1);
        }
        bool last_in = (b >> 63) & 1ull;
        if (last_in) {
            carry_start = starts ? (x0 + 63 - __clzll(starts)) : carry_start;
            carry_in = 1;
        } else {
            carry_in = 0;
        }
    }
    }
}

// grid = (ceil(N / CCL_BLK_PX), n): a workgroup walks CCL_BLK_PX pixels, a thread 4 consecutive ones per step (one dword
// of the image when rows allow), so the common case -- none of them in the set -- costs one load and no 64-bit index
// arithmetic, and the grid stays small enough that workgroup dispatch is not what the pass waits for
constexpr int CCL_BLK_PX = 8192;
__global__ __launch_bounds__(256) void k_ccl_merge(const uint8_t *__restrict__ img, int h, int w, int thr,
                                                   int invert, int conn8, const FrameState *__restrict__ st, int use_rect,
                                                   int *__restrict__ L)
{
    const int N = h * w;
    const size_t f = blockIdx.y;
    const Rect r = get_rect(st, f, use_rect, h, w);
    const uint8_t *im = img + f * (size_t)N;
    int *Lf = L + f * (size_t)N;
    const bool aligned = ((((size_t)im) | (size_t)N) & 3) == 0;
    for (int it = 0; it < CCL_BLK_PX / 1024; it++) {
        const int i0 = blockIdx.x * CCL_BLK_PX + it * 1024 + threadIdx.x * 4;
        if (i0 >= N) break;
        int y = i0 / w, x = i0 - y * w;
        if (y > r.y1 || y + 1 < r.y0) continue;
        if (aligned) {
            const uint32_t v4 = *reinterpret_cast<const uint32_t *>(im + i0);
            bool any = false;
#pragma unroll
            for (int k = 0; k < 4; k++) any |= ((((int)((v4 >> (8 * k)) & 255u) > thr) ? 1 : 0) != invert);
            if (!any) continue;
        }
        for (int k = 0; k < 4; k++, x++) {
            const int i = i0 + k;
            if (i >= N) break;
            if (x == w) { x = 0; y++; }
            if (y <= r.y0 || y > r.y1 || x < r.x0 || x > r.x1) continue;
            if (!pred(im, i, thr, invert)) continue;
            const bool up = pred(im, i - w, thr, invert);
            const bool left = x > r.x0 && pred(im, i - 1, thr, invert);
            if (up) {
                bool upleft = x > r.x0 && pred(im, i - w - 1, thr, invert);
                if (!(left && upleft)) uf_unite(Lf, i, i - w);
            } else if (conn8) {
                if (x < r.x1 && pred(im, i - w + 1, thr, invert)) {
                    bool right = pred(im, i + 1, thr, invert);
                    if (!right) uf_unite(Lf, i, i - w + 1);
                }
                if (x > r.x0 && !left && pred(im, i - w - 1, thr, invert)) uf_unite(Lf, i, i - w - 1);
            }
        }
    }
}

// ---- word-level walks (rows 16-byte aligned): a thread owns one 64-pixel word column over CCL_STRIP rows, packs each
// row's membership into a 64-bit mask (SWAR byte compare + multiply gather) and finds the pixels that have work to do --
// run starts, run-overlap starts -- with shifts and ANDs; only those few touch the label plane.
constexpr int CCL_STRIP = 8;
__device__ __forceinline__ unsigned pack8_gt(unsigned long long v, int thr, int invert)   // bit k = ((byte k > thr) != invert)
{
    const unsigned long long H = 0x8080808080808080ull, O = 0x0101010101010101ull;
    const unsigned long long g = ((v & ~H) + (unsigned long long)(0x7f - (thr & 0x7f)) * O) & H;   // low 7 bits > low 7 bits of thr
    unsigned long long res = (thr < 128) ? ((v & H) | g) : ((v & H) & g);
    if (invert) res ^= H;
    return (unsigned)(((res >> 7) * 0x0102040810204080ull) >> 56);
}
__device__ __forceinline__ unsigned long long pack_row64(const uint8_t *row, int x0, int w, int thr, int invert)
{
    unsigned long long bits = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int x = x0 + 16 * c;
        if (x < w) {   // w % 16 == 0: the 16 pixels are all inside
            const uint4 v = *reinterpret_cast<const uint4 *>(row + x);
            const unsigned long long lo = v.x | ((unsigned long long)v.y << 32), hi = v.z | ((unsigned long long)v.w << 32);
            bits |= (unsigned long long)(pack8_gt(lo, thr, invert) | (pack8_gt(hi, thr, invert) << 8)) << (16 * c);
        }
    }
    return bits;
}
__device__ __forceinline__ unsigned long long col_mask64(int x0, int lo, int hi)   // pixels x0 + b with lo <= x0 + b <= hi
{
    const int a = max(lo - x0, 0), b = min(hi - x0, 63);
    if (b < a) return 0ull;
    const unsigned long long m = (b == 63) ? ~0ull : ((1ull << (b + 1)) - 1ull);
    return m & ~((1ull << a) - 1ull);
}

// where the word-level walks read the set from: the u8 image (compare with a threshold) or a one-bit plane of it that
// already exists (cpe_dev.h BitWin layout: pixel x is bit x + 32 of its row; zero beyond the image) -- an eighth of the bytes
struct ByteSrc {
    const uint8_t *img; int w, thr, invert;
    __device__ __forceinline__ ByteSrc frame(size_t f, int h) const { return ByteSrc{img + f * (size_t)h * w, w, thr, invert}; }
    __device__ __forceinline__ unsigned long long pack(int y, int x0) const { return pack_row64(img + (size_t)y * w, x0, w, thr, invert); }
    __device__ __forceinline__ bool px(int y, int x) const { return pred(img + (size_t)y * w, x, thr, invert); }
};
struct BitSrc {
    const uint32_t *plane; int ws;
    __device__ __forceinline__ BitSrc frame(size_t f, int h) const { return BitSrc{plane + f * (size_t)h * ws, ws}; }
    __device__ __forceinline__ unsigned long long pack(int y, int x0) const
    {
        typedef unsigned long long u64a4 __attribute__((aligned(4)));
        return *reinterpret_cast<const u64a4 *>(plane + (size_t)y * ws + (x0 >> 5) + 1);
    }
    __device__ __forceinline__ bool px(int y, int x) const { return (plane[(size_t)y * ws + ((x + 32) >> 5)] >> ((x + 32) & 31)) & 1u; }
};

// labels of a sparse pass: every pixel of the set points at the first pixel of its run inside the word; a run that
// continues from the word to the left points at that word's last pixel instead (a chain of at most one link per word,
// parents always smaller: the same forest the row-wise k_ccl_init builds, a few links deeper)
template <class SRC>
__global__ __launch_bounds__(256) void k_ccl_init64(SRC src0, int h, int w,
                                                    const FrameState *__restrict__ st, int use_rect, int *__restrict__ L)
{
    const int WW = (w + 63) >> 6, strips = (h + CCL_STRIP - 1) / CCL_STRIP;
    const size_t f = blockIdx.y;
    const int gi = blockIdx.x * 256 + threadIdx.x;
    if (gi >= WW * strips) return;
    const int sy = gi / WW, j = gi - sy * WW;
    const Rect r = get_rect(st, f, use_rect, h, w);
    const int x0 = j * 64;
    if (r.x1 < r.x0 || x0 > r.x1 || x0 + 63 < r.x0) return;
    const int ya = max(sy * CCL_STRIP, r.y0), yb = min(sy * CCL_STRIP + CCL_STRIP - 1, r.y1);
    const size_t N = (size_t)h * w;
    const SRC src = src0.frame(f, h);
    int *Lf = L + f * N;
    const unsigned long long cmask = col_mask64(x0, r.x0, r.x1);
    const bool hasL = x0 - 1 >= r.x0;
    for (int y = ya; y <= yb; y++) {
        unsigned long long m = src.pack(y, x0) & cmask;
        if (!m) continue;
        const int base = y * w + x0;
        const bool cL = (m & 1ull) && hasL && src.px(y, x0 - 1);
        while (m) {
            const unsigned long long low = m & (0ull - m), run = m & ~(m + low);
            const int a = __ffsll((long long)low) - 1, e = a + __popcll(run);
            const int label = (a == 0 && cL) ? base - 1 : base + a;
            int b = a;
            while (b < e) {
                if ((b & 3) == 0 && b + 4 <= e) { *reinterpret_cast<int4 *>(Lf + base + b) = make_int4(label, label, label, label); b += 4; }
                else { Lf[base + b] = label; b++; }
            }
            m &= ~run;
        }
    }
}

// same unions as k_ccl_merge
template <class SRC>
__global__ __launch_bounds__(256) void k_ccl_merge64(SRC src0, int h, int w,
                                                     int conn8, const FrameState *__restrict__ st, int use_rect,
                                                     int *__restrict__ L)
{
    const int WW = (w + 63) >> 6, strips = (h + CCL_STRIP - 1) / CCL_STRIP;
    const size_t f = blockIdx.y;
    const int gi = blockIdx.x * 256 + threadIdx.x;
    if (gi >= WW * strips) return;
    const int sy = gi / WW, j = gi - sy * WW;
    const Rect r = get_rect(st, f, use_rect, h, w);
    const int x0 = j * 64;
    if (r.x1 < r.x0 || x0 > r.x1 || x0 + 63 < r.x0) return;
    const int ya = max(sy * CCL_STRIP, r.y0 + 1), yb = min(sy * CCL_STRIP + CCL_STRIP - 1, r.y1);
    if (ya > yb) return;
    const size_t N = (size_t)h * w;
    const SRC src = src0.frame(f, h);
    int *Lf = L + f * N;
    const unsigned long long cmask = col_mask64(x0, r.x0, r.x1);
    const bool hasL = x0 - 1 >= r.x0, hasR = x0 + 64 <= r.x1;
    unsigned long long A = src.pack(ya - 1, x0) & cmask;
    unsigned long long aL = (hasL && src.px(ya - 1, x0 - 1)) ? 1ull : 0ull, aR = (hasR && src.px(ya - 1, x0 + 64)) ? 1ull : 0ull;
    for (int y = ya; y <= yb; y++) {
        const unsigned long long C = src.pack(y, x0) & cmask;
        const unsigned long long cL = (hasL && src.px(y, x0 - 1)) ? 1ull : 0ull, cR = (hasR && src.px(y, x0 + 64)) ? 1ull : 0ull;
        if (C) {
            const unsigned long long Cs = (C << 1) | cL, As = (A << 1) | aL;            // left, up-left
            const int base = y * w + x0;
            unsigned long long m = C & A & ~(Cs & As);
            while (m) { const int b = __ffsll((long long)m) - 1; m &= m - 1; uf_unite(Lf, base + b, base + b - w); }
            if (conn8) {
                const unsigned long long Cr = (C >> 1) | (cR << 63), Ar = (A >> 1) | (aR << 63);   // right, up-right
                const unsigned long long nu = C & ~A;
                m = nu & Ar & ~Cr;
                while (m) { const int b = __ffsll((long long)m) - 1; m &= m - 1; uf_unite(Lf, base + b, base + b - w + 1); }
                m = nu & As & ~Cs;
                while (m) { const int b = __ffsll((long long)m) - 1; m &= m - 1; uf_unite(Lf, base + b, base + b - w - 1); }
            }
        }
        A = C; aL = cL; aR = cR;
    }
}

// component list of a roots-only pass: only the first pixel of a horizontal run can carry its own index
template <class SRC>
__global__ __launch_bounds__(256) void k_ccl_roots64(SRC src0, int h, int w,
                                                     FrameState *__restrict__ st, int use_rect, const int *__restrict__ L,
                                                     int *__restrict__ roots, int cnt_sel)
{
    const int WW = (w + 63) >> 6, strips = (h + CCL_STRIP - 1) / CCL_STRIP;
    const size_t f = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int gi = blockIdx.x * 256 + threadIdx.x;
    const Rect r = get_rect(st, f, use_rect, h, w);
    const size_t N = (size_t)h * w;
    const SRC src = src0.frame(f, h);
    const int *Lf = L + f * N;
    const int sy = gi / WW, j = gi - sy * WW, x0 = j * 64;
    const bool live = gi < WW * strips && !(r.x1 < r.x0 || x0 > r.x1 || x0 + 63 < r.x0);
    // (an empty rectangle may be INT_MAX .. -1: nothing of it enters the row arithmetic unless `live`)
    const int ya = live ? max(sy * CCL_STRIP, r.y0) : 0, yb = live ? min(sy * CCL_STRIP + CCL_STRIP - 1, r.y1) : -1;
    const unsigned long long cmask = live ? col_mask64(x0, r.x0, r.x1) : 0ull;
    const bool hasL = live && x0 - 1 >= r.x0;
    for (int k = 0; k < CCL_STRIP; k++) {     // wave-uniform trip count: the appends below are wavefront collectives
        const int y = ya + k;
        unsigned long long starts = 0;
        if (y <= yb) {
            const unsigned long long C = src.pack(y, x0) & cmask;
            const unsigned long long cL = (C & 1ull) && hasL && src.px(y, x0 - 1) ? 1ull : 0ull;
            starts = C & ~((C << 1) | cL);
        }
        while (__ballot(starts != 0)) {
            int i = -1;
            if (starts) { const int b = __ffsll((long long)starts) - 1; starts &= starts - 1; i = y * w + x0 + b; }
            const bool cand = i >= 0 && Lf[i] == i;
            const unsigned long long rb = __ballot(cand);
            if (!rb) continue;
            int base = 0;
            const int leader = __ffsll((long long)rb) - 1;
            if (lane == leader) base = atomicAdd(root_counter(st[f], cnt_sel), __popcll(rb));
            base = __shfl(base, leader, 64);
            if (cand) {
                const int q = base + __popcll(rb & ((1ull << lane) - 1ull));
                if (q < MAXROOTS) roots[f * MAXROOTS + q] = i;
                else set_overflow(st[f], OVF_ROOTS);
            }
        }
    }
}

// components of the set that reach the border of the working rectangle: touch[root] = 1
__global__ __launch_bounds__(256) void k_ccl_touch(const int *__restrict__ L, int n, int h, int w,
                                                   const FrameState *__restrict__ st, int use_rect,
                                                   uint8_t *__restrict__ touch)
{
    const int per = 2 * w + 2 * h;
    int gi = blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= n * per) return;
    int f = gi / per, k = gi - f * per;
    const Rect r = get_rect(st, f, use_rect, h, w);
    if (r.x1 < r.x0) return;
    const int rw = r.x1 - r.x0 + 1, rh = r.y1 - r.y0 + 1;
    int x, y;
    if (k < w) { if (k >= rw) return; x = r.x0 + k; y = r.y0; }
    else if (k < 2 * w) { if (k - w >= rw) return; x = r.x0 + k - w; y = r.y1; }
    else if (k < 2 * w + h) { if (k - 2 * w >= rh) return; x = r.x0; y = r.y0 + k - 2 * w; }
    else { if (k - 2 * w - h >= rh) return; x = r.x1; y = r.y0 + k - 2 * w - h; }
    size_t N = (size_t)h * w;
    const int *Lf = L + f * N;
    int v = Lf[(size_t)y * w + x];
    if (v >= 0) touch[f * N + uf_find(Lf, v)] = 1;
}

// component list only (roots-only passes): a root is a pixel of the set whose label is its own index.  Same walk as
// k_ccl_merge (CCL_BLK_PX pixels per workgroup, 4 per thread and step, dword reject).
__global__ __launch_bounds__(256) void k_ccl_roots4(const uint8_t *__restrict__ img, int h, int w, int thr, int invert,
                                                    FrameState *__restrict__ st, int use_rect, const int *__restrict__ L,
                                                    int *__restrict__ roots, int cnt_sel)
{
    const int N = h * w;
    const size_t f = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const uint8_t *im = img + f * (size_t)N;
    const int *Lf = L + f * (size_t)N;
    const Rect r = get_rect(st, f, use_rect, h, w);
    const bool aligned = ((((size_t)im) | (size_t)N) & 3) == 0;
    for (int it = 0; it < CCL_BLK_PX / 1024; it++) {
        const int i0 = blockIdx.x * CCL_BLK_PX + it * 1024 + threadIdx.x * 4;
        bool cand[4] = {false, false, false, false};
        if (i0 < N) {
            int y = i0 / w, x = i0 - y * w;
            if (!(y > r.y1 || y + 1 < r.y0)) {
                bool any = true;
                if (aligned) {
                    const uint32_t v4 = *reinterpret_cast<const uint32_t *>(im + i0);
                    any = false;
#pragma unroll
                    for (int k = 0; k < 4; k++) any |= ((((int)((v4 >> (8 * k)) & 255u) > thr) ? 1 : 0) != invert);
                }
                if (any) {
                    for (int k = 0; k < 4; k++, x++) {
                        const int i = i0 + k;
                        if (i >= N) break;
                        if (x == w) { x = 0; y++; }
                        if (y < r.y0 || y > r.y1 || x < r.x0 || x > r.x1) continue;
                        cand[k] = pred(im, i, thr, invert) && Lf[i] == i;
                    }
                }
            }
        }
        if (!__ballot(cand[0] || cand[1] || cand[2] || cand[3])) continue;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const unsigned long long rb = __ballot(cand[k]);
            if (!rb) continue;
            int base = 0;
            const int leader = __ffsll((long long)rb) - 1;
            if (lane == leader) base = atomicAdd(root_counter(st[f], cnt_sel), __popcll(rb));
            base = __shfl(base, leader, 64);
            if (cand[k]) {
                const int q = base + __popcll(rb & ((1ull << lane) - 1ull));
                if (q < MAXROOTS) roots[f * MAXROOTS + q] = i0 + k;
                else set_overflow(st[f], OVF_ROOTS);
            }
        }
    }
}

// flatten + count + collect roots + bounding box, one read of the label plane.
//   count_mode 0: none, 1: all pixels of the component, 2: interior pixels only (8 neighbours in the set, inside
//   the image).  Exact prune bounds of the blob detector: a hole of >= 5000 pixels, or a bright component with
//   >= 5000 interior pixels, has border-polygon area >= 5000 (polygon edges only cross the unit squares of their
//   own end-point pixels).
constexpr int CCL_FIN_PX = 8192;
__global__ __launch_bounds__(256) void k_ccl_finish(const uint8_t *__restrict__ img, int h, int w, int thr,
                                                    int invert, FrameState *__restrict__ st, int use_rect,
                                                    int *__restrict__ L, const uint8_t *__restrict__ touch, int count_mode,
                                                    int *__restrict__ cnt, int *__restrict__ roots, int *__restrict__ nrect, int sparse,
                                                    int noflatten, int cnt_sel)
{
    // grid = (ceil(N / CCL_FIN_PX), n): a workgroup never straddles two frames, so every wave-level aggregate below is
    // per frame; many steps of 256 pixels per workgroup keep the grid (and its dispatch time) small
    const size_t N = (size_t)h * w;
    const size_t f = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const Rect r = get_rect(st, f, use_rect, h, w);
    {   // the workgroup's pixels lie in rows ya .. yb: outside the working rectangle there is nothing to flatten, count or list
        const size_t p0 = (size_t)blockIdx.x * CCL_FIN_PX;
        const int ya = (int)(p0 / w), yb = (int)((min(p0 + CCL_FIN_PX, N) - 1) / w);
        if (r.x1 < r.x0 || yb < r.y0 || ya > r.y1) return;
    }
    for (int it = 0; it < CCL_FIN_PX / 256; it++) {
    const int i = blockIdx.x * CCL_FIN_PX + it * 256 + threadIdx.x;
    if ((size_t)(blockIdx.x * CCL_FIN_PX + it * 256) >= N) break;
    const size_t gi = f * N + (size_t)i;
    int root = -1, x = 0, y = 0;
    if ((size_t)i < N) {
        y = i / w; x = i - y * w;
        if (!(y < r.y0 || y > r.y1 || x < r.x0 || x > r.x1) && (!sparse || pred(img + f * N, i, thr, invert))) {
            int v = L[gi];
            if (noflatten) {
                root = (v == i) ? i : -1;   // only the component list is wanted: a root is a root, flattened or not
            } else if (v >= 0) {
                // read-only walk: the pointer jumping of uf_find_c re-points nodes at *an* ancestor, and such a store from
                // another thread's walk through this pixel may land after the store below and leave it one hop short
                root = uf_find(L + f * N, v);
                L[gi] = root;
            }
        }
    }
    const bool in = root >= 0;
    if (nrect && __ballot(in)) {
        int mnx = in ? x : INT_MAX, mxx = in ? x : INT_MIN, mny = in ? y : INT_MAX, mxy = in ? y : INT_MIN;
        for (int off = 32; off >= 1; off >>= 1) {
            mnx = min(mnx, __shfl_xor(mnx, off, 64)); mxx = max(mxx, __shfl_xor(mxx, off, 64));
            mny = min(mny, __shfl_xor(mny, off, 64)); mxy = max(mxy, __shfl_xor(mxy, off, 64));
        }
        if (lane == 0) {
            // most wavefronts already lie inside the accumulated box: test first, keep the atomics rare
            int *nr = nrect + 16 * f;   // 64 B per frame: its own cache line
            if (mnx < __hip_atomic_load(nr + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(nr + 0, mnx);
            if (mny < __hip_atomic_load(nr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(nr + 1, mny);
            if (mxx > __hip_atomic_load(nr + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(nr + 2, mxx);
            if (mxy > __hip_atomic_load(nr + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(nr + 3, mxy);
        }
    }
    const bool touched = in && touch && touch[f * N + root];
    if (roots) {
        // root list: one atomic per wavefront
        const bool is_root = in && !touched && root == i;
        unsigned long long rb = __ballot(is_root);
        if (rb) {
            int base = 0;
            const int leader = __ffsll((long long)rb) - 1;
            if (lane == leader) base = atomicAdd(root_counter(st[f], cnt_sel), __popcll(rb));
            base = __shfl(base, leader, 64);
            if (is_root) {
                int k = base + __popcll(rb & ((1ull << lane) - 1ull));
                if (k < MAXROOTS) roots[f * MAXROOTS + k] = i;
                else set_overflow(st[f], OVF_ROOTS);
            }
        }
    }
    if (count_mode == 1 || count_mode == 2) {
        bool c = in && !touched;
        if (c && count_mode == 2) {
            const uint8_t *im = img + f * N;
            bool inter = x > 0 && x < w - 1 && y > 0 && y < h - 1;
            if (inter) {
                inter = pred(im, i - w - 1, thr, invert) && pred(im, i - w, thr, invert) && pred(im, i - w + 1, thr, invert) &&
                        pred(im, i - 1, thr, invert) && pred(im, i + 1, thr, invert) && pred(im, i + w - 1, thr, invert) &&
                        pred(im, i + w, thr, invert) && pred(im, i + w + 1, thr, invert);
            }
            c = inter;
        }
        int key = c ? root : -1;
        unsigned long long active = __ballot(key >= 0);
        while (active) {
            int leader = __ffsll((long long)active) - 1;
            int lk = __shfl(key, leader, 64);
            unsigned long long same = __ballot(key == lk) & active;
            if (lane == leader) atomicAdd(&cnt[f * N + lk], __popcll(same));
            active &= ~same;
        }
    }
    }
}

__global__ void k_ccl_ctl(FrameState *st, int *nrect, int n, int h, int w, int op, int cnt_sel)
{
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    FrameState &S = st[f];
    int *nr = nrect ? nrect + 16 * f : nullptr;
    if (op == 0) {          // reset the root list
        *root_counter(S, cnt_sel) = 0;
    } else if (op == 1) {   // working rectangle = whole frame, accumulator empty
        S.crect[0] = 0; S.crect[1] = 0; S.crect[2] = w - 1; S.crect[3] = h - 1;
        nr[0] = INT_MAX; nr[1] = INT_MAX; nr[2] = -1; nr[3] = -1;
    } else if (op == 2) {   // working rectangle = accumulated bounding box; accumulator emptied
        for (int k = 0; k < 4; k++) { S.crect[k] = nr[k]; S.nrect[k] = nr[k]; }
        nr[0] = INT_MAX; nr[1] = INT_MAX; nr[2] = -1; nr[3] = -1;
    }
    // op 3: keep crect (a superset of the next, smaller set), accumulator already empty
}

// one wavefront per 512 pixels of a row (8 groups of 64): a ballot per threshold is the 64-bit group of that plane; lane t
// collects plane t's eight groups and writes them as 64 contiguous bytes
__global__ __launch_bounds__(256) void k_bitplanes(const uint8_t *__restrict__ img, int rows_total, int h, int w, int thr0, int step,
                                                   int nplanes, uint32_t *__restrict__ planes)
{
    const int lane = threadIdx.x & 63;
    const int chunks = (w + 63) >> 6, groups = (chunks + 7) >> 3;
    const long long gw = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gw >= (long long)rows_total * groups) return;
    const int row = (int)(gw / groups), g = (int)(gw - (long long)row * groups);
    const int f = row / h, y = row - f * h;
    const int ws = bit_row_words(w);
    const size_t plane_words = (size_t)h * ws;
    uint32_t *out = planes + (size_t)f * nplanes * plane_words + (size_t)y * ws;
    unsigned long long mine[8];
    const int c0 = g * 8, nc = min(8, chunks - c0);
#pragma unroll
    for (int q = 0; q < 8; q++) {
        mine[q] = 0;
        if (q < nc) {
            const int x = (c0 + q) * 64 + lane;
            const int v = x < w ? (int)img[(size_t)row * w + x] : -1;
            for (int t = 0; t < nplanes; t++) {
                unsigned long long b = __ballot(v > thr0 + t * step);
                if (lane == t) mine[q] = b;
            }
        }
    }
    if (lane < nplanes) {
        uint32_t *o = out + (size_t)lane * plane_words;
#pragma unroll
        for (int q = 0; q < 8; q++)
            if (q < nc) { o[1 + 2 * (c0 + q)] = (uint32_t)mine[q]; o[2 + 2 * (c0 + q)] = (uint32_t)(mine[q] >> 32); }
        if (g == 0) o[0] = 0;
        if (g == groups - 1) for (int k = 1 + 2 * chunks; k < ws; k++) o[k] = 0;
    }
}

// the same planes when rows start on 16-byte boundaries: a thread loads 64 pixels once (4 x 16 bytes) and makes their word of
// every plane with SWAR compares -- no ballots, 16 times fewer load instructions
__global__ __launch_bounds__(256) void k_bitplanes64(const uint8_t *__restrict__ img, int rows_total, int h, int w, int thr0, int step,
                                                     int nplanes, uint32_t *__restrict__ planes)
{
    const int chunks = (w + 63) >> 6;
    const long long gi = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gi >= (long long)rows_total * chunks) return;
    const int row = (int)(gi / chunks), j = (int)(gi - (long long)row * chunks);
    const int f = row / h, y = row - f * h;
    const int ws = bit_row_words(w);
    const size_t plane_words = (size_t)h * ws;
    uint32_t *out = planes + (size_t)f * nplanes * plane_words + (size_t)y * ws;
    const uint8_t *p = img + (size_t)row * w + j * 64;
    unsigned long long v[8];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        uint4 q = make_uint4(0, 0, 0, 0);
        if (j * 64 + 16 * c < w) q = *reinterpret_cast<const uint4 *>(p + 16 * c);   // w % 16 == 0: all 16 inside
        v[2 * c] = q.x | ((unsigned long long)q.y << 32);
        v[2 * c + 1] = q.z | ((unsigned long long)q.w << 32);
    }
    // pixels past the end of the row were loaded as 0 and every threshold is >= 0: their bits stay clear
    for (int t = 0; t < nplanes; t++) {
        const int thr = thr0 + t * step;
        unsigned long long bits = 0;
#pragma unroll
        for (int c = 0; c < 8; c++) bits |= (unsigned long long)pack8_gt(v[c], thr, 0) << (8 * c);
        uint32_t *o = out + (size_t)t * plane_words;
        o[1 + 2 * j] = (uint32_t)bits;
        o[2 + 2 * j] = (uint32_t)(bits >> 32);
        if (j == 0) o[0] = 0;
        if (j == chunks - 1) for (int k = 1 + 2 * chunks; k < ws; k++) o[k] = 0;
    }
}

// single plane (mask != 0 / image > thr): one thread per 8 pixels = one byte of the plane (pixel x is bit (x + 32) of its
// row, so byte 4 + x / 8 holds pixels 8 (x / 8) .. + 7), 64 consecutive bytes per wavefront
__global__ __launch_bounds__(256) void k_bitplane1(const uint8_t *__restrict__ img, int rows_total, int w, int thr,
                                                   uint32_t *__restrict__ plane)
{
    const int ws = bit_row_words(w), wb = ws * 4;             // bytes per plane row
    const long long gi = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gi >= (long long)rows_total * wb) return;
    const int row = (int)(gi / wb), j = (int)(gi - (long long)row * wb);
    const int x0 = (j - 4) * 8;
    unsigned b = 0;
    if (j >= 4 && x0 < w) {
        const uint8_t *p = img + (size_t)row * w + x0;
        if (x0 + 8 <= w && ((((size_t)p) & 3) == 0)) {
            const uint32_t a = *reinterpret_cast<const uint32_t *>(p), c = *reinterpret_cast<const uint32_t *>(p + 4);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                b |= ((int)((a >> (8 * k)) & 255u) > thr ? 1u : 0u) << k;
                b |= ((int)((c >> (8 * k)) & 255u) > thr ? 1u : 0u) << (4 + k);
            }
        } else {
            for (int k = 0; k < 8 && x0 + k < w; k++) b |= ((int)p[k] > thr ? 1u : 0u) << k;
        }
    }
    reinterpret_cast<uint8_t *>(plane)[(size_t)row * wb + j] = (uint8_t)b;
}

}  // namespace

int build_bitplanes(const uint8_t *img, int n, int h, int w, int thr0, int step, int nplanes, uint32_t *planes, hipStream_t s)
{
    CPE_CHECK_ARG(nplanes >= 1 && nplanes <= 64, "build_bitplanes: 1..64 planes");
    CPE_LAUNCH_BEGIN();
    if (nplanes == 1) {
        const long long bytes = (long long)n * h * bit_row_words(w) * 4;
        CPE_KLAUNCH(k_bitplane1, dim3((unsigned)((bytes + 255) / 256)), dim3(256), 0, s, img, n * h, w, thr0, planes);
        CPE_CHECK_LAUNCH("k_bitplane1");
        return CPE_OK;
    }
    if (w % 16 == 0 && (((size_t)img) & 15) == 0 && thr0 >= 0 && step >= 0 && thr0 + (nplanes - 1) * step <= 255) {
        const long long words = (long long)n * h * ((w + 63) >> 6);
        CPE_KLAUNCH(k_bitplanes64, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, img, n * h, h, w, thr0, step, nplanes, planes);
        CPE_CHECK_LAUNCH("k_bitplanes64");
        return CPE_OK;
    }
    const long long waves = (long long)n * h * ((((w + 63) >> 6) + 7) >> 3);
    CPE_KLAUNCH(k_bitplanes, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, img, n * h, h, w, thr0, step, nplanes, planes);
    CPE_CHECK_LAUNCH("k_bitplanes");
    return CPE_OK;
}

int ccl_ctl(FrameState *st, int *nrect, int n, int h, int w, int op, hipStream_t s)
{
    CPE_LAUNCH_BEGIN();
    CPE_KLAUNCH(k_ccl_ctl, dim3((n + 63) / 64), dim3(64), 0, s, st, nrect, n, h, w, op, 0);
    CPE_CHECK_LAUNCH("k_ccl_ctl");
    return CPE_OK;
}


// ---- RETR_EXTERNAL: which components lie inside a hole of another one ------------------------------------------------
// cv2.findContours(RETR_EXTERNAL) skips an outer border whose start pixel lies inside the outer border of a component
// found earlier (icvFindNextContour: the last border mark passed on the row is positive), i.e. every component that sits
// in a hole of another one, at any depth.  Equivalent, scan-free form: a component is external iff the background pixel
// west of its raster-first pixel belongs to the OUTER background, the 4-connected background region that reaches the
// border of the window.  This kernel computes that region as a bit mask, one workgroup per frame:
//   out[y][j] bit b = 1  <=>  pixel (64 j + b, y) is background and 4-connected to the window border through background
// by alternating downward / upward sweeps over the rows (lane j owns word j of a row; a row takes what the previous row
// of the sweep has, fills it sideways through runs of background -- carry-ripple fill inside a word, lane exchange across
// words) until a sweep changes nothing.  Typical masks (small blobs, open line fragments) converge in two sweeps; the
// third one only confirms.  Pixels outside the window count as outer background.  w <= 4096 (64 words).
__device__ __forceinline__ unsigned long long pack_nonzero64(const uint8_t *row, int x0, int w)
{
    unsigned long long m = 0;
    if (x0 + 64 <= w && ((((size_t)row) + x0) & 7) == 0) {
        const unsigned long long *p = reinterpret_cast<const unsigned long long *>(row + x0);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            unsigned long long v = p[k];
            v |= v >> 4; v |= v >> 2; v |= v >> 1;
            v &= 0x0101010101010101ull;
            m |= ((v * 0x0102040810204080ull) >> 56) << (8 * k);
        }
    } else {
        for (int b = 0; b < 64 && x0 + b < w; b++) m |= (unsigned long long)(row[x0 + b] != 0) << b;
    }
    return m;
}

// sideways fill of `seed` through the runs of bg, across the words held by the lanes 0 .. WW-1.
// Inside a word: adding the seeds to the run mask ripples a carry from every seed to the end of its run (fill towards the
// high bits); the same on the bit-reversed word fills towards the low bits.  Between words: a filled bit 63 / bit 0 seeds
// the neighbouring word's bit 0 / bit 63; repeat while any lane received a new seed.
__device__ __forceinline__ unsigned long long flood_row(unsigned long long bg, unsigned long long seed, int lane, int WW)
{
    unsigned long long f = seed & bg;
    const unsigned long long rb = __brevll(bg);
    for (;;) {
        f = (((bg + f) ^ bg) & bg) | f;
        const unsigned long long rf = __brevll(f);
        f = __brevll((((rb + rf) ^ rb) & rb) | rf);
        const unsigned long long from_lo = __shfl_up(f, 1, 64), from_hi = __shfl_down(f, 1, 64);
        unsigned long long add = 0;
        if (lane > 0 && lane < WW && (from_lo >> 63) && (bg & 1ull) && !(f & 1ull)) add |= 1ull;
        if (lane + 1 < WW && (from_hi & 1ull) && (bg >> 63) && !(f >> 63)) add |= 1ull << 63;
        if (!__ballot(add != 0ull)) break;
        f |= add;
    }
    return f;
}

// One workgroup per frame, FLOOD_BANDS wavefronts: the window's rows are cut into bands, one per wavefront.  A round is a
// downward and an upward sweep of every band at the same time, each starting from what its neighbour band's boundary row held
// at the last barrier (rows outside the window: all outer background); rounds repeat until one changes nothing anywhere.
// The result is the least fixed point of "background next to outer background (or to the window border) is outer
// background", which does not depend on the order of the updates: the same mask as one wavefront sweeping the whole window
// (the round-2 form: 1.16 ms per launch at 1920x1200, a serial walk of ~1000 rows three times).  The left / right window
// columns seed every row directly, so a sparse mask is nearly done after the first sweep of each band whatever the band above
// knows; what is shadowed from both sides on its own row gets filled from the rows above / below in the next rounds.
constexpr int FLOOD_BANDS = 16;
__global__ __launch_bounds__(64 * FLOOD_BANDS) void k_outside_flood(const uint8_t *__restrict__ mask, int h, int w, FrameState *__restrict__ st,
                                                                   int use_rect, unsigned long long *__restrict__ bgw_all,
                                                                   unsigned long long *__restrict__ out_all, size_t plane_words,
                                                                   const uint32_t *__restrict__ bits)
{
    // bits (optional): the mask's one-bit plane (build_bitplanes, 1 plane per frame) -- an eighth of the bytes of the first sweep
    __shared__ int s_any[2];
    const size_t f = blockIdx.x;
    const int lane = threadIdx.x & 63, band = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), WW = (w + 63) >> 6;
    const Rect r = get_rect(st, f, use_rect, h, w);
    if (r.x1 < r.x0 || r.y1 < r.y0) return;
    const uint8_t *im = mask + f * (size_t)h * w;
    const int bws = bit_row_words(w);
    const uint32_t *bp = bits ? bits + f * (size_t)h * bws : nullptr;
    unsigned long long *bgw = bgw_all + f * plane_words, *out = out_all + f * plane_words;
    const int x0 = lane * 64;
    const unsigned long long cmask = lane < WW ? col_mask64(x0, r.x0, r.x1) : 0ull;
    // window columns r.x0 / r.x1 touch the outside on their left / right
    unsigned long long edge = 0;
    if (lane < WW) {
        if (r.x0 >= x0 && r.x0 < x0 + 64) edge |= 1ull << (r.x0 - x0);
        if (r.x1 >= x0 && r.x1 < x0 + 64) edge |= 1ull << (r.x1 - x0);
    }
    const int nrows = r.y1 - r.y0 + 1;
    const int per = (nrows + FLOOD_BANDS - 1) / FLOOD_BANDS;
    const int ya = r.y0 + band * per, yb = min(r.y1, ya + per - 1);      // this wavefront's rows (none: yb < ya)
    const int mine = max(0, yb - ya + 1);
    if (threadIdx.x < 2) s_any[threadIdx.x] = 0;
    // Rows are loaded ahead of their use: the flood of a row only needs the previous row's result (a register), so the
    // next chunk of rows is requested before the current one is worked on.
    constexpr int CHUNK = 8;
    // boundary row of the neighbour band as of the last barrier (L1-bypassing load: another wavefront wrote it)
    auto nb_row = [&](int y) -> unsigned long long {
        if (y < r.y0 || y > r.y1) return ~0ull;                         // outside the window: all outer background
        return lane < WW ? __hip_atomic_load(&out[(size_t)y * WW + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    };
    // ---- first sweep (down): pack the background once, seed from the window border; the band above is not known yet
    {
        unsigned long long prev = band == 0 ? ~0ull : 0ull;
        unsigned long long nxt[CHUNK];
        auto load0 = [&](int k0, unsigned long long *dst) {
#pragma unroll
            for (int k = 0; k < CHUNK; k++) {
                const int y = ya + k0 + k;
                unsigned long long m = 0;
                if (lane < WW && k0 + k < mine)    // pixel x is bit x + 32 of its plane row: pixels 64 j .. 64 j + 63 are words 1 + 2 j, 2 + 2 j
                    m = bp ? *reinterpret_cast<const u64_a4 *>(bp + (size_t)y * bws + 1 + 2 * lane) : pack_nonzero64(im + (size_t)y * w, x0, w);
                dst[k] = ~m & cmask;
            }
        };
        load0(0, nxt);
        for (int k0 = 0; k0 < mine; k0 += CHUNK) {
            unsigned long long bgc[CHUNK];
#pragma unroll
            for (int k = 0; k < CHUNK; k++) bgc[k] = nxt[k];
            if (k0 + CHUNK < mine) load0(k0 + CHUNK, nxt);
#pragma unroll
            for (int k = 0; k < CHUNK; k++) {
                if (k0 + k >= mine) break;
                const int y = ya + k0 + k;
                const unsigned long long bg = bgc[k];
                unsigned long long seed = (prev | edge) & bg;
                if (y == r.y1) seed = bg;      // the row below the window is all outside
                // most rows of a sparse mask: every background pixel has outer background right above it
                const unsigned long long o = __ballot(seed != bg) ? flood_row(bg, seed, lane, WW) : bg;
                if (lane < WW) { bgw[(size_t)y * WW + lane] = bg; out[(size_t)y * WW + lane] = o; }
                prev = o;
            }
        }
    }
    __syncthreads();
    // ---- further sweeps (up, down, up, ...) until a whole round changes nothing in any band.  A row's entry is only
    // rewritten by the wavefront that owns it, in program order, so loading it a chunk early is safe.
    bool converged = false;
    for (int pass = 1; pass < FLOOD_MAX_PASSES; pass++) {
        const bool upw = pass & 1;
        bool any = false;
        unsigned long long prev = mine > 0 ? nb_row(upw ? yb + 1 : ya - 1) : 0ull;
        unsigned long long nb[CHUNK], nc[CHUNK];
        auto load1 = [&](int k0, unsigned long long *db, unsigned long long *dc) {
#pragma unroll
            for (int k = 0; k < CHUNK; k++) {
                const int kk = k0 + k;
                const int y = upw ? yb - kk : ya + kk;
                const bool ok = lane < WW && kk < mine;
                db[k] = ok ? bgw[(size_t)y * WW + lane] : 0ull;
                dc[k] = ok ? out[(size_t)y * WW + lane] : 0ull;
            }
        };
        load1(0, nb, nc);
        for (int k0 = 0; k0 < mine; k0 += CHUNK) {
            unsigned long long bgc[CHUNK], curc[CHUNK];
#pragma unroll
            for (int k = 0; k < CHUNK; k++) { bgc[k] = nb[k]; curc[k] = nc[k]; }
            if (k0 + CHUNK < mine) load1(k0 + CHUNK, nb, nc);
#pragma unroll
            for (int k = 0; k < CHUNK; k++) {
                const int kk = k0 + k;
                if (kk >= mine) break;
                const int y = upw ? yb - kk : ya + kk;
                const unsigned long long bg = bgc[k], cur = curc[k];
                const unsigned long long seed = cur | (prev & bg);
                unsigned long long o = cur;
                if (__ballot(seed != cur)) {
                    o = flood_row(bg, seed, lane, WW);
                    if (lane < WW && o != cur) { out[(size_t)y * WW + lane] = o; any = true; }
                }
                prev = o;
            }
        }
        // a round = an upward and a downward sweep (passes 2q - 1 and 2q); s_any[q & 1] collects its changes
        const int slot = ((pass + 1) >> 1) & 1;
        if (__ballot(any) && lane == 0) atomicOr(&s_any[slot], 1);
        __syncthreads();
        if (!upw) {                                  // end of a round
            const int ch = s_any[slot];
            __syncthreads();
            if (threadIdx.x == 0) s_any[slot] = 0;   // next used two rounds on: the barriers in between order the reset
            if (!ch) { converged = true; break; }
        }
    }
    // a background that still grows after FLOOD_MAX_PASSES alternating sweeps (a spiral thousands of turns deep) would
    // leave components wrongly classified as nested: report the frame instead of altering it silently
    if (!converged && threadIdx.x == 0) set_overflow(st[f], OVF_TRACE);
}

int outside_flood(const uint8_t *mask, int n, int h, int w, FrameState *st, int use_rect, unsigned long long *bgw,
                  unsigned long long *out, size_t plane_words, hipStream_t s, const uint32_t *bits)
{
    CPE_CHECK_ARG(w <= 4096 && plane_words >= (size_t)h * ((w + 63) >> 6), "outside_flood: frame too wide or scratch too small");
    CPE_LAUNCH_BEGIN();
    CPE_KLAUNCH(k_outside_flood, dim3(n), dim3(64 * FLOOD_BANDS), 0, s, mask, h, w, st, use_rect, bgw, out, plane_words, bits);
    CPE_CHECK_LAUNCH("k_outside_flood");
    return CPE_OK;
}

// One labelling pass.  roots (optional): component list in st[].n_roots / roots; holes_only drops components
// that reach the border of the working rectangle (needs `touch`); count_mode/cnt as in k_ccl_finish;
// count_mode 3 only zeroes cnt inside the set's rectangle; sparse 1: labels of pixels outside the set are left untouched,
// sparse 2: they are written as singletons (own raster index).
// use_rect: restrict to st[].crect; nrect (optional, int[n][16]): accumulate the set's bounding box there.
int ccl_run(const uint8_t *img, int n, int h, int w, int thr, int invert, int conn8, int *L, int *roots, bool holes_only,
            uint8_t *touch, int count_mode, int *cnt, int use_rect, int *nrect, FrameState *st, hipStream_t s, int sparse, int flags, int cnt_sel)
{
    // flags: CCL_ROOTS_ONLY (1) = component list without flattening the label plane (needs roots, no counts / touch / bbox);
    //        CCL_LINKS_ONLY (2) = stop after the unions: the consumer resolves the few labels it needs with uf_find
    if (flags & 1) { CPE_CHECK_ARG(roots && !holes_only && !count_mode && !nrect, "ccl_run: roots-only pass with extra outputs"); }
    const size_t N = (size_t)h * w, total = N * n;
    const int rows = n * h;
    CPE_LAUNCH_BEGIN();
    if (roots) CPE_KLAUNCH(k_ccl_ctl, dim3((n + 63) / 64), dim3(64), 0, s, st, (int *)nullptr, n, h, w, 0, cnt_sel);
    // rows that start on 16-byte boundaries: word-level walks
    const bool words = (w % 16 == 0) && (((size_t)img & 15) == 0) && (((size_t)L & 15) == 0);
    const dim3 gwords((unsigned)((((w + 63) / 64) * ((h + CCL_STRIP - 1) / CCL_STRIP) + 255) / 256), n);
    if (words && sparse == 1 && !count_mode)
        CPE_KLAUNCH(k_ccl_init64<ByteSrc>, gwords, dim3(256), 0, s, (ByteSrc{img, w, thr, invert}), h, w, (const FrameState *)st, use_rect, L);
    else
        CPE_KLAUNCH(k_ccl_init, dim3((rows + CCL_INIT_ROWS - 1) / CCL_INIT_ROWS), dim3(256), 0, s, img, rows, h, w, thr, invert, (const FrameState *)st, use_rect, L,
                    count_mode ? cnt : (int *)nullptr, sparse);
    if (words)
        CPE_KLAUNCH(k_ccl_merge64<ByteSrc>, gwords, dim3(256), 0, s, (ByteSrc{img, w, thr, invert}), h, w, conn8, (const FrameState *)st, use_rect, L);
    else
        CPE_KLAUNCH(k_ccl_merge, dim3((unsigned)((N + CCL_BLK_PX - 1) / CCL_BLK_PX), n), dim3(256), 0, s, img, h, w, thr, invert, conn8,
                    (const FrameState *)st, use_rect, L);
    if (holes_only) {
        (void)hipMemsetAsync(touch, 0, total, s);
        int per = 2 * w + 2 * h;
        CPE_KLAUNCH(k_ccl_touch, dim3((n * per + 255) / 256), dim3(256), 0, s, (const int *)L, n, h, w, (const FrameState *)st,
                    use_rect, touch);
    }
    if ((flags & 1) && words)
        CPE_KLAUNCH(k_ccl_roots64<ByteSrc>, gwords, dim3(256), 0, s, (ByteSrc{img, w, thr, invert}), h, w, st, use_rect, (const int *)L, roots, cnt_sel);
    else if (flags & 1)
        CPE_KLAUNCH(k_ccl_roots4, dim3((unsigned)((N + CCL_BLK_PX - 1) / CCL_BLK_PX), n), dim3(256), 0, s, img, h, w, thr, invert, st, use_rect, (const int *)L,
                    roots, cnt_sel);
    else if (!(flags & 2))
        CPE_KLAUNCH(k_ccl_finish, dim3((unsigned)((N + CCL_FIN_PX - 1) / CCL_FIN_PX), n), dim3(256), 0, s, img, h, w, thr, invert, st, use_rect, L,
                    holes_only ? (const uint8_t *)touch : (const uint8_t *)nullptr, count_mode, cnt, roots, nrect, sparse, flags & 1, cnt_sel);
    CPE_CHECK_LAUNCH("ccl_run");
    return CPE_OK;
}

// ccl_run(mask, ..., thr 0, 8-connected, sparse 1, CCL_ROOTS_ONLY) for a mask whose one-bit plane (build_bitplanes, 1 plane)
// already exists: the three word-level walks read the plane.  Needs w % 16 == 0 and a 16-byte aligned label plane; the caller
// falls back to ccl_run otherwise (returns CPE_ERR_ARG without launching).
int ccl_roots_bits(const uint32_t *bits, int n, int h, int w, int *L, int *roots, int use_rect, FrameState *st, hipStream_t s, int cnt_sel)
{
    if (!((w % 16 == 0) && (((size_t)L & 15) == 0) && bits && roots)) return CPE_ERR_ARG;
    CPE_LAUNCH_BEGIN();
    CPE_KLAUNCH(k_ccl_ctl, dim3((n + 63) / 64), dim3(64), 0, s, st, (int *)nullptr, n, h, w, 0, cnt_sel);
    const dim3 gwords((unsigned)((((w + 63) / 64) * ((h + CCL_STRIP - 1) / CCL_STRIP) + 255) / 256), n);
    const BitSrc src{bits, bit_row_words(w)};
    CPE_KLAUNCH(k_ccl_init64<BitSrc>, gwords, dim3(256), 0, s, src, h, w, (const FrameState *)st, use_rect, L);
    CPE_KLAUNCH(k_ccl_merge64<BitSrc>, gwords, dim3(256), 0, s, src, h, w, 1, (const FrameState *)st, use_rect, L);
    CPE_KLAUNCH(k_ccl_roots64<BitSrc>, gwords, dim3(256), 0, s, src, h, w, st, use_rect, (const int *)L, roots, cnt_sel);
    CPE_CHECK_LAUNCH("ccl_roots_bits");
    return CPE_OK;
}

}  // namespace cpe
