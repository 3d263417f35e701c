// Device-side building blocks shared by the image-stage kernels (gfx950).
#pragma once
#include "cpe_internal.h"

namespace cpe {

// ---------------------------------------------------------------- per-image state kept in the workspace
constexpr int MAXROOTS = 262144;  // components per labelling pass and image (4K frames: ~48k noise specks in the joints mask)
// component lists of the blob sweep: one pool of (first pixel, count) entries per frame and list, the 17 thresholds one
// after the other in the order they are filled (the first entry of a threshold = the sum of the counters of the ones before
// it, region.hip sw_slot).  CLAHE turns sensor noise into specks: 1920x1200 frames with +-7..11 DN of noise (and an intensity
// ramp, tools/stress_parity.py) were seen with 770 000 dark components away from the border, 560 000 bright ones and 115 000
// followed hole borders over all thresholds; clean frames stay below 30 000.
enum { SWL_DARK = 0, SWL_BRIGHT = 1, SWL_TRACE = 2 };
__host__ __device__ inline int sweep_pool(int h, int w, int which)
{
    const long long N = (long long)h * w;
    long long v = which == SWL_TRACE ? N / 8 : N / 2;   // largest totals seen: 770 000 dark (noise + intensity ramp), 560 000 bright, 115 000
    const long long lo = which == SWL_TRACE ? (1 << 18) : (1 << 19);   // small frames: what 17 lists of 32768 entries held
    v = v < lo ? lo : (v > (1 << 23) ? (1 << 23) : v);
    return (int)((v + 255) / 256 * 256);
}
constexpr int MAXJ = CPE_MAXJ;   // joints kept inside the region rectangle (include/cpe.h)
constexpr int MAXB = 32768;      // blobs per threshold
constexpr int MAXG = 32768;      // blob groups (one per unmatched blob: a noisy intensity ramp makes thousands)
constexpr int MAXG_LDS = 1024;   // ... whose middle centres sit in k_blob_merge's LDS (24 KB; a clean frame has ~500); the rest are read from HBM
constexpr int GCAP = 48;         // centres per group (17 thresholds + same-threshold neighbours that fall into the same group)
constexpr int MAXV = 131072;     // contour-vertex scratch (int2) per image
constexpr int MAXL = CPE_MAXL;   // grid lines per direction (label groups of the joints: noise joints make extra ones)
constexpr int MAXLP = CPE_MAXLP; // joints per label group: a limit, not a slot size (the groups share one pool of MAXJ points)
constexpr int MAXSEG = 2048;     // line fragments per mask in the expansion stage

struct CompRec {        // one traced border
    int root;           // start pixel (raster index) = discovery key
    int is_hole;
    long long a00, a10, a01;  // Green sums (exact integers)
    int nverts;         // CHAIN_APPROX_SIMPLE vertex count
    int npts;           // CHAIN_APPROX_NONE point count
    int minx, maxx, miny, maxy;
    int voff;           // offset of stored vertices (or -1)
    int pad;
};

struct BlobRec { double x, y, r; int key; int pad; };
struct Group { int n; int pad; double c[GCAP][3]; };

struct FrameState {
    int status;
    int rect[4];        // boundingRect(max_contour)
    int r0;             // circle_radius0
    int spot[4];        // ellipse cx, cy, a, b
    int n_roots;        // scratch counters (reset by the stage that uses them)
    int n_comps;
    int n_joints_all;   // all joints
    int n_joints;       // joints inside rect (sorted in OpenCV contour order)
    int n_blobs;
    int n_groups;
    int n_groups_prev;
    int n_kp;
    int n_verts;        // bump pointer of the vertex scratch
    int n_dists;        // bump pointer of the distance scratch
    int best_comp;      // index of the selected contour
    int n_seg[2];       // valid fragments per mask (h, v)
    float gang[2], glen[2];
    int n_rows, n_cols;
    int overflow;
    int hull_n;
    int crect[4];       // working rectangle of a restricted labelling pass (x0, y0, x1, y1)
    int nrect[4];       // bounding-box accumulator
    int n_roots_p;      // component count of the joints labelling (runs on its own stream, beside the region stage)
    int n_roots_s;      // component count of the spot labelling (third stream)
    int spot_fail;      // the spot chain found no saturated spot: folded into `status` when the chains join
    int srect[4];       // window of the spot labelling (x0, y0, x1, y1): the tiles that hold a pixel > 240, + 16 px
};

struct SegRec { float p1x, p1y, p2x, p2y, angle, len; int valid; int pad; };

// stage buffer bundles (all device pointers into the caller's workspace, plane-major [n][...])
struct RegionBuffers {
    uint8_t *cl, *ext, *mc, *touch;
    int *lab, *cnt, *lab2, *cnt2, *roots, *sw, *nrect, *bk;
    double *gmid;              // middle centres (x, y, r) of the blob groups beyond MAXG_LDS
    int *hpar; uint8_t *htime;   // merge history of the bright forest: (absorbing root, step) per absorbed entry
    uint32_t *pool;   // border points of the hole traces (chunked)
    unsigned short *blob_ch;   // first 16 chunk ids of every blob's border
    int maxch, maxdf;          // capacities per frame and threshold: border-point chunks, distance scratch (doubles)
    uint32_t *bits;   // 17 one-bit planes per frame (threshold images), reused for single mask planes later
    int2 *hl, *bl, *tl;   // per-threshold component lists of the blob sweep (dark / bright)
    unsigned int *hist;
    uint8_t *lut;
    BlobRec *blobs;
    int *blob_d, *order;
    double *dists;
    Group *groups;
    unsigned long long *best;
    int *lohi, *hull;
};
// grid.x of the kernels that walk a per-frame list (components, blobs, fragments) in turns: enough workgroups per frame
// to fill the chip when the batch is small, few when it is large (a grid sized for the list capacity would be mostly
// empty workgroups: hundreds of thousands of them cost more than the work)
inline unsigned frame_waves(int lists, int lo, int hi)
{
    const int v = 16384 / (lists > 0 ? lists : 1);
    return (unsigned)(v < lo ? lo : (v > hi ? hi : v));
}
// optional helper stream of the region stage: the hole borders are followed while the bright sweep runs
struct RegionSide { hipStream_t s; hipEvent_t clahe_done, dark_done, traced, medians;
                    hipEvent_t joints_done = nullptr, spot_done = nullptr; };   // ends of the joints / spot chains (their label planes hold the tracers' tables afterwards)
struct MaskBuffers {
    uint8_t *binary, *hmask, *vmask, *joints_mask, *tmpA, *tmpB, *g19, *cm, *mc, *roi_h, *roi_v, *base_h, *base_v, *exp_h,
        *exp_v, *touch;
    int *lab, *roots, *jtmp, *joints, *verts;
    int *lab_p, *roots_p, *lab_s, *roots_s;
    int *lab_h, *lab_v;                       // union-find planes of the two expanded masks (read by k_lines)   // label planes / component lists of the joints and spot chains
    uint32_t *bits;
    unsigned long long *best, *best_s;
    unsigned long long *fl_j;                 // joints chain: [n][h * ceil(w/64)] background masks, then the same of outer background
    uint32_t *jbits;                          // one-bit plane of the joints mask (build_bitplanes layout, one plane per frame)
    SegRec *segs;
};

// FrameState::overflow is a bit mask of the fixed capacity that was exceeded (any bit => CPE_ST_OVERFLOW)
enum { OVF_ROOTS = 1, OVF_LINES = 2, OVF_TRACE = 4, OVF_JOINTS = 8, OVF_VERTS = 16, OVF_SEGS = 32, OVF_KERNEL = 64,
       OVF_EXPAND = 128, OVF_BLOBS = 256, OVF_DISTS = 512, OVF_GROUPS = 1024, OVF_SWEEP = 2048 };
__device__ __forceinline__ void set_overflow(FrameState &S, int bit) { atomicOr(&S.overflow, bit); }

// which per-frame component counter a labelling pass fills: 0 the main chain, 1 joints chain, 2 spot chain
__device__ __forceinline__ int *root_counter(FrameState &S, int sel)
{
    return sel == 0 ? &S.n_roots : (sel == 1 ? &S.n_roots_p : &S.n_roots_s);
}

// ---------------------------------------------------------------- RETR_EXTERNAL (ccl.hip: k_outside_flood)
// outer-background mask of every frame's window (use_rect 0: the frame, 2: region rectangle + 2 px): out[f][y][j] bit b = pixel
// (64 j + b, y) is background and 4-connected to the window border.  bgw / out: n * plane_words u64 of scratch each,
// plane_words >= h * ceil(w / 64).
// Frames up to 4096 columns wide (one wavefront holds a row as 64 words of 64 pixels; wider frames are refused with
// CPE_ERR_ARG by the callers' argument check).  A flood that has not converged after FLOOD_MAX_PASSES sweeps sets OVF_TRACE
// (a sweep walks one band of rows, a sixteenth of the window: 65536 of them are the 4096 whole-window sweeps of round 2).
constexpr int FLOOD_MAX_PASSES = 65536;
// bits (optional): the mask's one-bit plane (build_bitplanes with one plane per frame); the first sweep reads it instead of the bytes
int outside_flood(const uint8_t *mask, int n, int h, int w, FrameState *st, int use_rect, unsigned long long *bgw,
                  unsigned long long *out, size_t plane_words, hipStream_t s, const uint32_t *bits = nullptr);
// a component (raster-first pixel `root`) is external iff the pixel west of that pixel is outer background
// (cv2.findContours(RETR_EXTERNAL) drops the components that lie in a hole of another one)
__device__ __forceinline__ bool comp_is_external(const unsigned long long *out_f, int w, int root, int win_x0)
{
    const int y = root / w, x = root - y * w;
    if (x - 1 < win_x0) return true;          // the window border / the image border is outer background
    return (out_f[(size_t)y * ((w + 63) >> 6) + ((x - 1) >> 6)] >> ((x - 1) & 63)) & 1ull;
}
__device__ __forceinline__ int window_x0(const FrameState &S, int use_rect)
{
    return use_rect == 2 ? max(S.rect[0] - 2, 0) : (use_rect == 1 ? S.crect[0] : 0);
}

// grey-level bucket of a CLAHE value for the blob sweep: 0: v <= 50 (dark at every threshold), b: 50 + 10 (b - 1) < v <= 50 + 10 b,
// 17: v > 210.  Bucket b joins the dark forest at threshold slot b and the bright forest at slot b - 1.
__host__ __device__ inline int sweep_level(int v) { return v <= 50 ? 0 : (((v - 41) / 10) < 17 ? ((v - 41) / 10) : 17); }

// ---------------------------------------------------------------- union-find on an int label plane
// S: ints per node (1: a plain label plane; 2: the bright forest of the blob sweep, whose node is {parent, merge-history word})
template <int S = 1> __device__ __forceinline__ int uf_load(const int *L, int i)
{
    return __hip_atomic_load(L + (size_t)i * S, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int S = 1> __device__ __forceinline__ int uf_find(const int *L, int x)
{
    int p;
    while ((p = uf_load<S>(L, x)) != x) x = p;
    return x;
}
// find with intermediate pointer jumping (ECL-CC): every node on the walked path is re-pointed at its
// grandparent.  Parents always have smaller indices and roots are never written, so concurrent use with
// uf_unite is safe; stale writes can only re-point a node at another of its ancestors.
template <int S = 1> __device__ __forceinline__ int uf_find_c(int *L, int x)
{
    int curr = uf_load<S>(L, x);
    if (curr != x) {
        int prev = x, next;
        while (curr > (next = uf_load<S>(L, curr))) {
            __hip_atomic_store(L + (size_t)prev * S, next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            prev = curr;
            curr = next;
        }
    }
    return curr;
}
template <int S = 1> __device__ __forceinline__ void uf_unite(int *L, int a, int b)
{
    for (;;) {
        a = uf_find_c<S>(L, a);
        b = uf_find_c<S>(L, b);
        if (a == b) return;
        if (a < b) { int t = a; a = b; b = t; }
        int old = atomicMin(&L[(size_t)a * S], b);
        if (old == a) return;
        a = old;
    }
}

// ---------------------------------------------------------------- Suzuki border following (OpenCV icvFetchContour)
// Pred(x, y) -> bool : pixel is non-zero (false outside the image).
// Visitor.point(x, y, is_vertex): every border pixel in order (CHAIN_APPROX_NONE); is_vertex marks the
// subset CHAIN_APPROX_SIMPLE keeps.  Returns false if the step bound was hit.
// The 8 neighbours of the current pixel are fetched together (independent loads, one memory latency per
// border step instead of one per examined neighbour); the direction search then runs on the bit mask.
template <class Pred>
__device__ __forceinline__ unsigned nbr_mask(Pred &nz, int x, int y)
{
    const bool b0 = nz(x + 1, y), b1 = nz(x + 1, y - 1), b2 = nz(x, y - 1), b3 = nz(x - 1, y - 1);
    const bool b4 = nz(x - 1, y), b5 = nz(x - 1, y + 1), b6 = nz(x, y + 1), b7 = nz(x + 1, y + 1);
    return (b0 ? 1u : 0u) | (b1 ? 2u : 0u) | (b2 ? 4u : 0u) | (b3 ? 8u : 0u) | (b4 ? 16u : 0u) | (b5 ? 32u : 0u) |
           (b6 ? 64u : 0u) | (b7 ? 128u : 0u);
}

// 1-bit planes for border following.  A row is `bit_row_words(w)` u32 words; pixel x is bit (x + 32) of its row: one
// zero word on the left and at least two on the right, so every 64-column window that starts on a word boundary is
// one 8-byte load and never wraps into the next row.  BitWin keeps a 16-row x 64-column window around the current
// border pixel in LDS (one column of `win` per lane): a border step costs three LDS reads, and global memory is
// touched only when the border leaves the window (every ~10 steps) instead of eight dependent loads per step.
constexpr int BW_ROWS = 16;
__host__ __device__ inline int bit_row_words(int w) { return ((w + 31) >> 5) + 4; }
typedef unsigned long long u64_a4 __attribute__((aligned(4)));
struct BitWin {
    const uint32_t *plane;       // this frame's plane
    int ws, h;
    unsigned long long *win;     // LDS, row r of this lane at win[r * 64]
    int wx0 = 0, wy0 = INT_MIN / 2;   // padded bit index of window column 0, image row of window row 0
    // (re)fill the window around (x, y).  The window is put ahead of the border's direction of travel: leaving through
    // the top / bottom row puts the pixel on the bottom / top side of the new window, leaving through the left / right
    // columns puts it on the right / left side -- borders keep their heading for a while, so a refill lasts about
    // twice as many steps as a centred window.
    __device__ __forceinline__ void load(int x, int y, int row_in_win, int col_lo)
    {
        const int k = (x + 32 - col_lo) >> 5;   // the pixel lands in columns col_lo .. col_lo + 31 of the window
        wx0 = 32 * k;
        wy0 = y - row_in_win;
#pragma unroll
        for (int r = 0; r < BW_ROWS; r++) {
            const int yy = wy0 + r;
            const int cy = min(max(yy, 0), h - 1);
            unsigned long long v = *(const u64_a4 *)(plane + (size_t)cy * ws + k);
            win[r * 64] = ((unsigned)yy < (unsigned)h) ? v : 0ull;
        }
    }
    __device__ __forceinline__ unsigned nbrs(int x, int y)
    {
        int p = x + 32 - wx0, r = y - wy0;
        if (p < 1 || p > 62 || r < 1 || r > BW_ROWS - 2) {
            const bool fresh = wy0 == INT_MIN / 2;
            int row = BW_ROWS / 2, col = 16;
            if (!fresh) {
                if (r < 1) row = BW_ROWS - 3;          // heading up
                else if (r > BW_ROWS - 2) row = 2;     // heading down
                if (p < 1) col = 29;                   // heading left: columns 29..60 (x + 32 >= 32 keeps k >= 0)
                else if (p > 62) col = 3;              // heading right: columns 3..34
            }
            load(x, y, row, col);
            p = x + 32 - wx0; r = y - wy0;
        }
        const unsigned ta = (unsigned)(win[(r - 1) * 64] >> (p - 1)) & 7u;   // bit 0: x - 1, bit 1: x, bit 2: x + 1
        const unsigned tb = (unsigned)(win[r * 64] >> (p - 1)) & 7u;
        const unsigned tc = (unsigned)(win[(r + 1) * 64] >> (p - 1)) & 7u;
        return ((tb >> 2) & 1u) | (((ta >> 2) & 1u) << 1) | (((ta >> 1) & 1u) << 2) | ((ta & 1u) << 3) | ((tb & 1u) << 4) |
               ((tc & 1u) << 5) | (((tc >> 1) & 1u) << 6) | (((tc >> 2) & 1u) << 7);
    }
    __device__ __forceinline__ bool operator()(int x, int y) const   // single pixel, straight from the plane
    {
        if ((unsigned)x >= (unsigned)(32 * (ws - 3)) || (unsigned)y >= (unsigned)h) return false;
        return (plane[(size_t)y * ws + ((x + 32) >> 5)] >> ((x + 32) & 31)) & 1u;
    }
};
__device__ __forceinline__ unsigned nbr_mask(BitWin &bw, int x, int y) { return bw.nbrs(x, y); }

// planes[t] = (img > thr0 + t * step), t < nplanes; plane stride per frame = nplanes_alloc * h * bit_row_words(w)
int build_bitplanes(const uint8_t *img, int n, int h, int w, int thr0, int step, int nplanes, uint32_t *planes, hipStream_t s);
int ccl_roots_bits(const uint32_t *bits, int n, int h, int w, int *L, int *roots, int use_rect, FrameState *st, hipStream_t s, int cnt_sel);

// direction s = 0..7 counter-clockwise from east (x right, y down): DX = {1,1,0,-1,-1,-1,0,1}, DY = {0,-1,-1,-1,0,1,1,1},
// stored as 2-bit fields (value + 1) so a step needs no table in memory
__device__ __forceinline__ int trace_dx(int s) { return (int)((0x901Au >> (2 * s)) & 3u) - 1; }
__device__ __forceinline__ int trace_dy(int s) { return (int)((0xA901u >> (2 * s)) & 3u) - 1; }

template <class Pred, class Visitor>
__device__ bool trace_border(Pred &nz, int x0, int y0, bool is_hole, Visitor &vis, int max_steps)
{
    int s, s_end;
    s_end = s = is_hole ? 0 : 4;
    unsigned bits = nbr_mask(nz, x0, y0);
    do {
        s = (s - 1) & 7;
    } while (!((bits >> s) & 1u) && s != s_end);
    if (s == s_end) {
        vis.point(x0, y0, true);
        return true;
    }
    const int x1 = x0 + trace_dx(s), y1 = y0 + trace_dy(s);
    int x3 = x0, y3 = y0, prev_s = s ^ 4;
    for (int step = 0; step < max_steps; step++) {
        // first set neighbour counter-clockwise after s (the pixel we came from is always set, so one exists)
        const unsigned rot = ((bits | (bits << 8)) >> ((s + 1) & 7)) & 0xFFu;
        if (rot) s = (s + __ffs(rot)) & 7;
        const int x4 = x3 + trace_dx(s), y4 = y3 + trace_dy(s);
        bool vertex = (s != prev_s);
        vis.point(x3, y3, vertex);
        if (vertex) prev_s = s;
        if (x4 == x0 && y4 == y0 && x3 == x1 && y3 == y1) return true;
        if (vis.stop()) return true;   // the visitor has seen enough (e.g. more vertices than its consumer accepts)
        x3 = x4;
        y3 = y4;
        s = (s + 4) & 7;
        bits = nbr_mask(nz, x3, y3);
    }
    return false;
}

struct MaskPred {
    const uint8_t *m;
    int w, h;
    __device__ __forceinline__ bool operator()(int x, int y) const
    {
        // unconditional load from the clamped address: lets the 8 neighbour loads of one border step issue together
        const bool inb = (unsigned)x < (unsigned)w && (unsigned)y < (unsigned)h;
        const int cx = min(max(x, 0), w - 1), cy = min(max(y, 0), h - 1);
        const uint8_t v = m[(size_t)cy * w + cx];
        return inb & (v != 0);
    }
};
struct ThreshPred {  // binarised = img > t
    const uint8_t *m;
    int w, h, t;
    __device__ __forceinline__ bool operator()(int x, int y) const
    {
        const bool inb = (unsigned)x < (unsigned)w && (unsigned)y < (unsigned)h;
        const int cx = min(max(x, 0), w - 1), cy = min(max(y, 0), h - 1);
        const int v = m[(size_t)cy * w + cx];
        return inb & (v > t);
    }
};

// Green sums + counts + bbox; optional vertex store
struct StatVisitor {
    long long a00 = 0, a10 = 0, a01 = 0;
    int npts = 0, nverts = 0;
    int minx = INT_MAX, maxx = INT_MIN, miny = INT_MAX, maxy = INT_MIN;
    bool have_prev = false;
    int fx = 0, fy = 0, px = 0, py = 0;  // first and previous written point
    __device__ __forceinline__ void edge(int x0, int y0, int x1, int y1)
    {
        long long dxy = (long long)x0 * y1 - (long long)x1 * y0;
        a00 += dxy;
        a10 += dxy * (x0 + x1);
        a01 += dxy * (y0 + y1);
    }
    __device__ __forceinline__ void point(int x, int y, bool vertex)
    {
        npts++;
        if (vertex) nverts++;
        minx = min(minx, x); maxx = max(maxx, x); miny = min(miny, y); maxy = max(maxy, y);
        if (have_prev) edge(px, py, x, y);
        else { fx = x; fy = y; have_prev = true; }
        px = x; py = y;
    }
    __device__ __forceinline__ void finish() { if (have_prev) edge(px, py, fx, fy); }
    __device__ __forceinline__ bool stop() const { return false; }
};

// cv2.moments(contour): m00, m10, m01 from the Green sums (contourMoments)
__device__ __forceinline__ void moments_from_sums(long long a00, long long a10, long long a01, double &m00,
                                                  double &m10, double &m01)
{
    m00 = m10 = m01 = 0;
    double d00 = (double)a00;
    if (fabs(d00) > 1.1920928955078125e-07) {
        double db1_2, db1_6;
        if (d00 > 0) { db1_2 = 0.5; db1_6 = 0.16666666666666666666666666666667; }
        else { db1_2 = -0.5; db1_6 = -0.16666666666666666666666666666667; }
        m00 = d00 * db1_2;
        m10 = (double)a10 * db1_6;
        m01 = (double)a01 * db1_6;
    }
}

}  // namespace cpe
