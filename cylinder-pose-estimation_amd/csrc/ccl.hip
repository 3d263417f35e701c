// Connected-component labelling on the GPU (replaces cv2.connectedComponents, util_cylinder.py:28, and
// is the front end of every cv2.findContours call site: components -> one border trace each).
//
// Set membership: in(p) = (img[p] > thr) != invert.  Foreground sets use 8-connectivity, background
// sets 4-connectivity (the pairing under which Suzuki-Abe outer / hole borders are defined).
// Three passes over an int32 label plane (label = raster index of the component's first pixel):
//   init   : one wavefront per image row; every pixel points at the first pixel of its horizontal run
//            (ballot + bit scan, carried across 64-pixel chunks)
//   merge  : runs are united with the row above through atomicMin union-find (only where a run
//            starts on either side, so a long run costs O(1) unions per neighbour run)
//   flatten: every pixel reads its root
// HBM bytes per pixel: 1 (image) + 4 written + 4 read/written + 4 read/written.
#include "cpe_dev.h"

namespace cpe {

namespace {

__device__ __forceinline__ bool pred(const uint8_t *img, size_t i, int thr, int invert)
{
    return (((int)img[i] > thr) ? 1 : 0) != invert;
}

__global__ __launch_bounds__(256) void k_ccl_init(const uint8_t *__restrict__ img, int rows_total, int h, int w,
                                                  int thr, int invert, int *__restrict__ L, int *__restrict__ cnt)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows_total) return;
    const int y = row % h;
    const size_t base = (size_t)row * w;  // == frame * h*w + y*w
    int carry_in = 0, carry_start = 0;
    for (int x0 = 0; x0 < w; x0 += 64) {
        int x = x0 + lane;
        bool in = x < w && pred(img, base + x, thr, invert);
        unsigned long long b = __ballot(in);
        unsigned long long prev = (b << 1) | (unsigned long long)carry_in;
        unsigned long long starts = b & ~prev;
        if (cnt && x < w) cnt[base + x] = 0;
        if (in) {
            unsigned long long m = starts & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
            int sx = m ? (x0 + 63 - __clzll(m)) : carry_start;
            L[base + x] = y * w + sx;
        } else if (x < w) {
            L[base + x] = -1;
        }
        // carry for the next chunk
        bool last_in = (b >> 63) & 1ull;
        if (last_in) {
            unsigned long long m = starts;
            carry_start = m ? (x0 + 63 - __clzll(m)) : carry_start;
            carry_in = 1;
        } else {
            carry_in = 0;
        }
    }
}

__global__ __launch_bounds__(256) void k_ccl_merge(const uint8_t *__restrict__ img, size_t total, int h, int w, int thr,
                                                   int invert, int conn8, int *__restrict__ L)
{
    size_t gi = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= total) return;
    const size_t N = (size_t)h * w;
    const size_t f = gi / N;
    const int i = (int)(gi - f * N);
    const int y = i / w, x = i - y * w;
    if (y == 0) return;
    const uint8_t *im = img + f * N;
    if (!pred(im, i, thr, invert)) return;
    int *Lf = L + f * N;
    const bool up = pred(im, i - w, thr, invert);
    const bool left = x > 0 && pred(im, i - 1, thr, invert);
    if (up) {
        bool upleft = x > 0 && pred(im, i - w - 1, thr, invert);
        if (!(left && upleft)) uf_unite(Lf, i, i - w);
    } else if (conn8) {
        if (x + 1 < w && pred(im, i - w + 1, thr, invert)) {
            bool right = pred(im, i + 1, thr, invert);
            if (!right) uf_unite(Lf, i, i - w + 1);
        }
        if (x > 0 && !left && pred(im, i - w - 1, thr, invert)) uf_unite(Lf, i, i - w - 1);
    }
}

__global__ __launch_bounds__(256) void k_ccl_flatten(size_t total, size_t N, int *__restrict__ L)
{
    size_t gi = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= total) return;
    int v = L[gi];
    if (v < 0) return;
    const size_t f = gi / N;
    L[gi] = uf_find(L + f * N, v);
}

// frame-connected background components: touch[root] = 1
__global__ __launch_bounds__(256) void k_ccl_touch(const int *__restrict__ L, int n, int h, int w, uint8_t *__restrict__ touch)
{
    const int per = 2 * w + 2 * h;
    int gi = blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= n * per) return;
    int f = gi / per, k = gi - f * per;
    int x, y;
    if (k < w) { x = k; y = 0; }
    else if (k < 2 * w) { x = k - w; y = h - 1; }
    else if (k < 2 * w + h) { x = 0; y = k - 2 * w; }
    else { x = w - 1; y = k - 2 * w - h; }
    size_t N = (size_t)h * w;
    int v = L[f * N + (size_t)y * w + x];
    if (v >= 0) touch[f * N + v] = 1;
}

// per-component pixel counts, aggregated per wavefront before the atomic (one add per distinct root
// in a wave).  interior_only: count only pixels whose 8 neighbours are all in the set and inside the image.
// Used as exact prune bounds for the blob detector: a hole of >= 5000 pixels has polygon area >= 5000, and
// a bright component with >= 5000 interior pixels has outer-polygon area >= 5000 (border-polygon edges
// only cross the unit squares of their own end-point pixels).
__global__ __launch_bounds__(256) void k_ccl_count(const uint8_t *__restrict__ img, const int *__restrict__ L,
                                                   size_t total, int h, int w, int thr, int invert, int interior_only,
                                                   const uint8_t *__restrict__ touch, int *__restrict__ cnt)
{
    size_t gi = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t N = (size_t)h * w;
    int root = -1;
    size_t f = 0;
    if (gi < total) {
        f = gi / N;
        const int i = (int)(gi - f * N);
        root = L[gi];
        if (root >= 0 && touch && touch[f * N + root]) root = -1;   // frame-connected background: never a hole
        if (root >= 0 && interior_only) {
            const int y = i / w, x = i - y * w;
            const uint8_t *im = img + f * N;
            bool inter = x > 0 && x < w - 1 && y > 0 && y < h - 1;
            if (inter) {
                inter = pred(im, i - w - 1, thr, invert) && pred(im, i - w, thr, invert) && pred(im, i - w + 1, thr, invert) &&
                        pred(im, i - 1, thr, invert) && pred(im, i + 1, thr, invert) && pred(im, i + w - 1, thr, invert) &&
                        pred(im, i + w, thr, invert) && pred(im, i + w + 1, thr, invert);
            }
            if (!inter) root = -1;
        }
    }
    // key = frame-local root; lanes of one wave may straddle two frames: include the frame in the key
    long long key = root >= 0 ? (long long)(f * N) + root : -1;
    unsigned long long active = __ballot(key >= 0);
    const int lane = threadIdx.x & 63;
    while (active) {
        int leader = __ffsll((long long)active) - 1;
        long long lk = __shfl(key, leader, 64);
        unsigned long long same = __ballot(key == lk) & active;
        if (lane == leader) atomicAdd(&cnt[lk], __popcll(same));
        active &= ~same;
    }
}

// roots -> per-frame list (order arbitrary; consumers sort or are order-independent)
__global__ __launch_bounds__(256) void k_collect_roots(const int *__restrict__ L, const uint8_t *__restrict__ touch,
                                                       size_t total, size_t N, int *__restrict__ roots,
                                                       FrameState *__restrict__ st)
{
    size_t gi = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= total) return;
    const size_t f = gi / N;
    const int i = (int)(gi - f * N);
    if (L[gi] != i) return;
    if (touch && touch[gi]) return;
    int k = atomicAdd(&st[f].n_roots, 1);
    if (k < MAXROOTS) roots[f * MAXROOTS + k] = i;
    else st[f].overflow = 1;
}

__global__ void k_reset_roots(FrameState *st, int n)
{
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < n) st[f].n_roots = 0;
}

}  // namespace

// labels for the set {(img > thr) != invert}; conn8 selects 8- vs 4-connectivity
int ccl_label_count(const uint8_t *img, int n, int h, int w, int thr, int invert, int conn8, int *L, int *cnt,
                    int interior_only, uint8_t *touch, hipStream_t s)
{
    const size_t N = (size_t)h * w, total = N * n;
    const int rows = n * h;
    CPE_LAUNCH_BEGIN();
    CPE_KLAUNCH(k_ccl_init, dim3((rows + 3) / 4), dim3(256), 0, s, img, rows, h, w, thr, invert, L, cnt);
    CPE_KLAUNCH(k_ccl_merge, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, img, total, h, w, thr, invert,
                       conn8, L);
    CPE_KLAUNCH(k_ccl_flatten, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, total, N, L);
    if (touch) {
        (void)hipMemsetAsync(touch, 0, total, s);
        int per = 2 * w + 2 * h;
        CPE_KLAUNCH(k_ccl_touch, dim3((n * per + 255) / 256), dim3(256), 0, s, L, n, h, w, touch);
    }
    CPE_KLAUNCH(k_ccl_count, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, img, L, total, h, w, thr, invert,
                       interior_only, (const uint8_t *)touch, cnt);
    CPE_CHECK_LAUNCH("ccl_label_count");
    return CPE_OK;
}

int ccl_label(const uint8_t *img, int n, int h, int w, int thr, int invert, int conn8, int *L, hipStream_t s)
{
    const size_t N = (size_t)h * w, total = N * n;
    const int rows = n * h;
    CPE_LAUNCH_BEGIN();
    CPE_KLAUNCH(k_ccl_init, dim3((rows + 3) / 4), dim3(256), 0, s, img, rows, h, w, thr, invert, L, (int *)nullptr);
    CPE_KLAUNCH(k_ccl_merge, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, img, total, h, w, thr, invert,
                       conn8, L);
    CPE_KLAUNCH(k_ccl_flatten, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, total, N, L);
    CPE_CHECK_LAUNCH("ccl_label");
    return CPE_OK;
}

// collect component roots; holes_only: skip components that touch the image frame (needs a zeroed touch plane)
int ccl_collect(const int *L, int n, int h, int w, bool holes_only, uint8_t *touch, int *roots, FrameState *st,
                hipStream_t s, bool touch_ready)
{
    const size_t N = (size_t)h * w, total = N * n;
    CPE_LAUNCH_BEGIN();
    CPE_KLAUNCH(k_reset_roots, dim3((n + 63) / 64), dim3(64), 0, s, st, n);
    if (holes_only && !touch_ready) {
        (void)hipMemsetAsync(touch, 0, total, s);
        int per = 2 * w + 2 * h;
        CPE_KLAUNCH(k_ccl_touch, dim3((n * per + 255) / 256), dim3(256), 0, s, L, n, h, w, touch);
    }
    CPE_KLAUNCH(k_collect_roots, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, L,
                       holes_only ? touch : (const uint8_t *)nullptr, total, N, roots, st);
    CPE_CHECK_LAUNCH("ccl_collect");
    return CPE_OK;
}

}  // namespace cpe
