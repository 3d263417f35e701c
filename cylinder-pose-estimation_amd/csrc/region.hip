// Stage a-3 on the GPU: detect_largest_blob (util_cylinder.py:1830-1899)
//   L = LAB-L LUT(gray) -> CLAHE(4.5, 4x4) -> SimpleBlobDetector -> filled discs -> largest external
//   contour -> convex hull -> filled hull mask (mask_contour) + boundingRect(max_contour).
// [ext] OpenCV 4.5.5 semantics restated exactly as in oracle/src/orc_blob.c (parity unpinned vs cv2).
//
// SimpleBlobDetector = 17 binarisations (50..210 step 10).  Per threshold: foreground (8-conn) and
// background (4-conn) labelling, one thread per component follows its Suzuki border (outer border of a
// bright component / hole border of an enclosed dark one), Green-theorem moments, area + colour
// filters, median border distance (one wavefront per blob), then the order-dependent cross-threshold
// grouping runs as one wavefront per frame.  Suzuki-Abe starts every hole border at the pixel west of
// the hole's raster-first pixel and every outer border at the component's raster-first pixel, so the
// parallel formulation visits exactly the borders the sequential raster scan does.
#include "cpe_dev.h"
#include <stdlib.h>

namespace cpe {

int ccl_run(const uint8_t *img, int n, int h, int w, int thr, int invert, int conn8, int *L, int *roots, bool holes_only,
            uint8_t *touch, int count_mode, int *cnt, int use_rect, int *nrect, FrameState *st, hipStream_t s, int sparse = 0, int flags = 0, int cnt_sel = 0);
int ccl_ctl(FrameState *st, int *nrect, int n, int h, int w, int op, hipStream_t s);

namespace {

__constant__ uint8_t c_lab_l[256] = { 0, 1, 1, 2, 2, 3, 5, 5, 6, 7, 7, 8, 9, 9, 10, 11, 12, 12, 14, 15, 16, 17, 18, 19, 21, 23, 24, 25, 27, 27, 28, 30, 31, 33, 34, 35, 36, 38, 39, 40, 41, 42, 43, 45, 46, 47, 48, 50, 51, 52, 53, 54, 55, 57, 58, 59, 60, 61, 62, 63, 65, 66, 67, 68, 69, 70, 71, 73, 74, 75, 76, 77, 78, 79, 80, 82, 82, 83, 85, 86, 87, 88, 89, 90, 91, 92, 93, 94, 95, 97, 98, 99, 100, 101, 102, 103, 104, 105, 106, 107, 108, 109, 110, 111, 112, 113, 114, 115, 116, 117, 119, 119, 121, 122, 123, 124, 125, 126, 127, 128, 129, 130, 131, 132, 133, 134, 135, 136, 137, 138, 139, 140, 141, 142, 143, 144, 145, 146, 147, 148, 149, 150, 151, 152, 153, 154, 155, 156, 156, 157, 158, 159, 160, 161, 162, 163, 164, 165, 166, 167, 168, 169, 170, 171, 172, 173, 174, 175, 176, 177, 178, 179, 180, 180, 181, 182, 183, 184, 185, 186, 187, 188, 189, 190, 191, 192, 193, 194, 195, 196, 196, 197, 198, 199, 200, 201, 202, 203, 204, 205, 206, 207, 208, 208, 209, 210, 211, 212, 213, 214, 215, 216, 217, 218, 219, 219, 220, 221, 222, 223, 224, 225, 226, 227, 228, 228, 229, 230, 231, 232, 233, 234, 235, 236, 237, 237, 238, 239, 240, 241, 242, 243, 244, 245, 245, 246, 247, 248, 249, 250, 251, 252, 253, 253, 254, 255 };

__device__ __forceinline__ int sat_u8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// ---- CLAHE ----------------------------------------------------------------------------------------
struct ClaheGeom { int tilesX, tilesY, tw, th, eh, ew, clipLimit; float lutScale; };

// lab_lut != 0: `gray` is a grey frame and L = c_lab_l[grey]; 0: `gray` already is the L plane of a colour frame (k_bgr2labl)
__global__ __launch_bounds__(256) void k_clahe_hist(const uint8_t *__restrict__ gray, int n, int h, int w, ClaheGeom g,
                                                    int strips, unsigned int *__restrict__ hist, int lab_lut)
{
    // one 256-bin histogram per wavefront (the background of a frame sits in a dozen bins: with one histogram for the
    // workgroup every LDS atomic waited for the other wavefronts' hits on the same few addresses), four pixels per load
    __shared__ unsigned int sh[4][256];
    int b = blockIdx.x;
    int strip = b % strips; b /= strips;
    int tile = b % (g.tilesX * g.tilesY);
    int f = b / (g.tilesX * g.tilesY);
    int ty = tile / g.tilesX, tx = tile - ty * g.tilesX;
    for (int k = 0; k < 4; k++) sh[k][threadIdx.x] = 0;
    __syncthreads();
    unsigned int *mine = sh[threadIdx.x >> 6];
    const uint8_t *im = gray + (size_t)f * h * w;
    int rows_per = (g.th + strips - 1) / strips;
    int y0 = strip * rows_per, y1 = min(g.th, y0 + rows_per);
    const int gx0 = tx * g.tw;
    // the tile lies inside the frame and its rows start on dwords (frames whose size is a multiple of 4: no padding)
    const bool fast = gx0 + g.tw <= w && ty * g.th + g.th <= h && (g.tw & 3) == 0 && (w & 3) == 0 && (((size_t)im + gx0) & 3) == 0;
    if (fast) {
        const int qw = g.tw >> 2;
        for (int i = threadIdx.x; i < (y1 - y0) * qw; i += 256) {
            const int yy = i / qw, q = i - yy * qw;
            const uint32_t v4 = *reinterpret_cast<const uint32_t *>(im + (size_t)(ty * g.th + y0 + yy) * w + gx0 + 4 * q);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int v = (v4 >> (8 * k)) & 255u;
                atomicAdd(&mine[lab_lut ? (int)c_lab_l[v] : v], 1u);
            }
        }
    } else {
        for (int yy = y0; yy < y1; yy++) {
            int gy = reflect101(ty * g.th + yy, h);
            for (int xx = threadIdx.x; xx < g.tw; xx += 256) {
                int gx = reflect101(gx0 + xx, w);
                const int v = im[(size_t)gy * w + gx];
                atomicAdd(&mine[lab_lut ? (int)c_lab_l[v] : v], 1u);
            }
        }
    }
    __syncthreads();
    unsigned int v = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
    if (v) atomicAdd(&hist[((size_t)f * g.tilesX * g.tilesY + tile) * 256 + threadIdx.x], v);
}

__global__ __launch_bounds__(256) void k_clahe_lut(unsigned int *__restrict__ hist, ClaheGeom g, uint8_t *__restrict__ lut)
{
    __shared__ int sh[256];
    __shared__ int s_clipped;
    unsigned int *hh = hist + (size_t)blockIdx.x * 256;
    int t = threadIdx.x;
    int v = (int)hh[t];
    if (t == 0) s_clipped = 0;
    __syncthreads();
    if (g.clipLimit > 0 && v > g.clipLimit) {
        atomicAdd(&s_clipped, v - g.clipLimit);
        v = g.clipLimit;
    }
    __syncthreads();
    int clipped = s_clipped;
    int batch = clipped / 256, residual = clipped - batch * 256;
    v += batch;
    if (residual != 0) {
        int stepr = max(256 / residual, 1);
        // hist[i]++ for i = 0, stepr, 2*stepr, ... while i < 256 and fewer than residual increments
        if (t % stepr == 0 && t / stepr < residual) v++;
    }
    sh[t] = v;
    __syncthreads();
    if (t == 0) {
        int sum = 0;
        for (int i = 0; i < 256; i++) { sum += sh[i]; sh[i] = sum; }
    }
    __syncthreads();
    lut[(size_t)blockIdx.x * 256 + t] = (uint8_t)sat_u8((int)rintf((float)sh[t] * g.lutScale));
}

// grid = (ceil(N / CLAHE_BLK_PX), n).  The frame's 16 tile LUTs (4 KB) sit in LDS; a thread takes 4 neighbouring pixels
// per step (one dword in, one dword out when rows allow).  Also accumulates the bounding box of the result's pixels > 50
// (the lowest threshold of the blob detector) in nrect[f]: the working rectangle of the whole sweep comes with the pass
// that writes the image.  (Many pixels per workgroup keep the same-address atomics on the box rare.)
constexpr int CLAHE_BLK_PX = 8192;
// bucket_cnt (optional): the sweep's grey-level bucket sizes (sw + SW_BS, SW_STRIDE ints per frame) are counted here as well:
// every pixel above the lowest threshold lies inside the working rectangle by the rectangle's definition, so the counting
// pass of the bucket sort (one more read of the image) is not needed.
__global__ __launch_bounds__(256) void k_clahe_apply(const uint8_t *__restrict__ gray, int h, int w,
                                                     ClaheGeom g, const uint8_t *__restrict__ lut,
                                                     uint8_t *__restrict__ dst, int *__restrict__ nrect, int lab_lut,
                                                     int *__restrict__ bucket_cnt, int bucket_stride)
{
    __shared__ int s_b[4];
    __shared__ int s_lv[32];
    if (threadIdx.x < 32) s_lv[threadIdx.x] = 0;
    __shared__ uint8_t s_lut[16 * 256];
    __shared__ uint8_t s_lab[256];
    const int N = h * w;
    const size_t f = blockIdx.y;
    if (threadIdx.x == 0) { s_b[0] = INT_MAX; s_b[1] = INT_MAX; s_b[2] = -1; s_b[3] = -1; }
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(lut + f * 16 * 256);   // tilesX = tilesY = 4
        uint32_t *d32 = reinterpret_cast<uint32_t *>(s_lut);
        for (int i = threadIdx.x; i < 16 * 256 / 4; i += 256) d32[i] = src[i];
        s_lab[threadIdx.x] = lab_lut ? c_lab_l[threadIdx.x] : (uint8_t)threadIdx.x;
    }
    __syncthreads();
    int mnx = INT_MAX, mny = INT_MAX, mxx = -1, mxy = -1;
    const float inv_tw = 1.0f / g.tw, inv_th = 1.0f / g.th;
    const uint8_t *gf = gray + f * (size_t)N;
    uint8_t *df = dst + f * (size_t)N;
    const bool al4 = ((((size_t)gf | (size_t)df) & 3) == 0);
    auto one = [&](int i, int v0) -> int {
        const int y = i / w, x = i - y * w;
        float tyf = y * inv_th - 0.5f;
        int ty1 = (int)floorf(tyf), ty2 = ty1 + 1;
        float ya = tyf - ty1, ya1 = 1.0f - ya;
        ty1 = max(ty1, 0);
        ty2 = min(ty2, g.tilesY - 1);
        float txf = x * inv_tw - 0.5f;
        int tx1 = (int)floorf(txf), tx2 = tx1 + 1;
        float xa = txf - tx1, xa1 = 1.0f - xa;
        tx1 = max(tx1, 0);
        tx2 = min(tx2, g.tilesX - 1);
        const int v = s_lab[v0];
        const uint8_t *p1 = s_lut + (ty1 * g.tilesX) * 256, *p2 = s_lut + (ty2 * g.tilesX) * 256;
        float a = (float)p1[tx1 * 256 + v] * xa1, b = (float)p1[tx2 * 256 + v] * xa;
        float c = (float)p2[tx1 * 256 + v] * xa1, d = (float)p2[tx2 * 256 + v] * xa;
        float res = (a + b) * ya1 + (c + d) * ya;
        const int out = sat_u8((int)rintf(res));
        if (out > 50) {
            mnx = min(mnx, x); mxx = max(mxx, x); mny = min(mny, y); mxy = max(mxy, y);
            if (bucket_cnt) atomicAdd(&s_lv[sweep_level(out)], 1);
        }
        return out;
    };
    // the four pixels of a dword lie in one row when the rows are a multiple of 4 long: row index, row weights and the two
    // rows of tile tables once per dword (the per-pixel form spent a third of its instructions on i / w and the row terms)
    auto four = [&](int i0, uint32_t v4) -> uint32_t {
        const int y = i0 / w, x0 = i0 - y * w;
        float tyf = y * inv_th - 0.5f;
        int ty1 = (int)floorf(tyf), ty2 = ty1 + 1;
        const float ya = tyf - ty1, ya1 = 1.0f - ya;
        ty1 = max(ty1, 0);
        ty2 = min(ty2, g.tilesY - 1);
        const uint8_t *p1 = s_lut + (ty1 * g.tilesX) * 256, *p2 = s_lut + (ty2 * g.tilesX) * 256;
        uint32_t o4 = 0;
        bool any = false;
        int xlo = INT_MAX, xhi = -1;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int x = x0 + q;
            float txf = x * inv_tw - 0.5f;
            int tx1 = (int)floorf(txf), tx2 = tx1 + 1;
            const float xa = txf - tx1, xa1 = 1.0f - xa;
            tx1 = max(tx1, 0);
            tx2 = min(tx2, g.tilesX - 1);
            const int v = s_lab[(v4 >> (8 * q)) & 255u];
            const float a = (float)p1[tx1 * 256 + v] * xa1, b = (float)p1[tx2 * 256 + v] * xa;
            const float c = (float)p2[tx1 * 256 + v] * xa1, d = (float)p2[tx2 * 256 + v] * xa;
            const float res = (a + b) * ya1 + (c + d) * ya;
            const int out = sat_u8((int)rintf(res));
            if (out > 50) {
                any = true; xlo = min(xlo, x); xhi = max(xhi, x);
                if (bucket_cnt) atomicAdd(&s_lv[sweep_level(out)], 1);
            }
            o4 |= (uint32_t)out << (8 * q);
        }
        if (any) { mnx = min(mnx, xlo); mxx = max(mxx, xhi); mny = min(mny, y); mxy = max(mxy, y); }
        return o4;
    };
    const bool row4 = al4 && (w & 3) == 0;
    for (int k = 0; k < CLAHE_BLK_PX / 1024; k++) {
        const int i0 = blockIdx.x * CLAHE_BLK_PX + k * 1024 + threadIdx.x * 4;
        if (i0 >= N) break;
        if (row4) {
            *reinterpret_cast<uint32_t *>(df + i0) = four(i0, *reinterpret_cast<const uint32_t *>(gf + i0));
        } else if (al4 && i0 + 4 <= N) {
            const uint32_t v4 = *reinterpret_cast<const uint32_t *>(gf + i0);
            uint32_t o4 = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) o4 |= (uint32_t)one(i0 + q, (v4 >> (8 * q)) & 255u) << (8 * q);
            *reinterpret_cast<uint32_t *>(df + i0) = o4;
        } else {
            for (int q = 0; q < 4 && i0 + q < N; q++) df[i0 + q] = (uint8_t)one(i0 + q, gf[i0 + q]);
        }
    }
    if (__ballot(mxx >= 0)) {
        for (int off = 32; off >= 1; off >>= 1) {
            mnx = min(mnx, __shfl_xor(mnx, off, 64)); mxx = max(mxx, __shfl_xor(mxx, off, 64));
            mny = min(mny, __shfl_xor(mny, off, 64)); mxy = max(mxy, __shfl_xor(mxy, off, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&s_b[0], mnx); atomicMin(&s_b[1], mny); atomicMax(&s_b[2], mxx); atomicMax(&s_b[3], mxy);
        }
    }
    __syncthreads();
    if (bucket_cnt && threadIdx.x > 0 && threadIdx.x < 32 && s_lv[threadIdx.x])
        atomicAdd(&bucket_cnt[f * bucket_stride + threadIdx.x], s_lv[threadIdx.x]);
    if (threadIdx.x == 0 && s_b[2] >= 0) {
        int *nr = nrect + 16 * f;   // test first: most workgroups lie inside the box already
        if (s_b[0] < __hip_atomic_load(nr + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(nr + 0, s_b[0]);
        if (s_b[1] < __hip_atomic_load(nr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(nr + 1, s_b[1]);
        if (s_b[2] > __hip_atomic_load(nr + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(nr + 2, s_b[2]);
        if (s_b[3] > __hip_atomic_load(nr + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(nr + 3, s_b[3]);
    }
}

// ---- blobs per threshold ----------------------------------------------------------------------------
struct DistVisitor {
    double cx, cy;
    double *out;
    int k = 0;
    __device__ __forceinline__ void point(int x, int y, bool)
    {
        double dx = cx - (double)x, dy = cy - (double)y;
        out[k++] = sqrt(dx * dx + dy * dy);
    }
    __device__ __forceinline__ bool stop() const { return false; }
};

// ---- threshold sweep bookkeeping: per-frame int counters (SW_STRIDE ints per frame)
constexpr int NTHR = 17;          // thresholds 50, 60, ..., 210 (SimpleBlobDetector defaults, util_cylinder.py:1836)
constexpr int SW_STRIDE = 192;
constexpr int NBK = NTHR + 1;     // grey-level buckets: 0: v <= 50, b: 50 + 10 (b - 1) < v <= 50 + 10 b, 17: v > 210
enum {
    SW_NH = 8,                    // + k: dark components away from the rectangle border at threshold k (length of hl[k])
    SW_NL = SW_NH + NTHR,         // + k: bright components at threshold k (length of bl[k])
    SW_NB = SW_NL + NTHR,         // + k: blobs of threshold k
    SW_ND = SW_NB + NTHR,         // + k: border distances stored for threshold k
    SW_BS = SW_ND + NTHR,         // + b: pixels of bucket b inside the rectangle
    SW_BO = SW_BS + NBK,          // + b: first entry of bucket b in the bucket plane
    SW_BC = SW_BO + NBK,          // + b: fill cursor
    SW_NT = SW_BC + NBK,          // + k: holes of threshold k whose border is followed (length of tl[k])
    SW_NC = SW_NT + NTHR,         // + k: border-point chunks in use
    SW_NA = SW_NC + NTHR          // + k: blobs of threshold k that came from hole borders (they are listed first)
};
static_assert(SW_NA + NTHR <= SW_STRIDE, "sweep counters");

// a threshold's share of a pooled list (cpe_dev.h sweep_pool): first entry and number of entries that fit.  `desc`: the
// list is filled from the highest threshold down (bright sweep).  The counters of the thresholds not yet reached are 0.
struct SwSlot { int off, cnt; };
__device__ __forceinline__ SwSlot sw_slot(const int *S, int cnt_base, int slot, bool desc, int pool)
{
    int o = 0;
    if (desc) { for (int j = NTHR - 1; j > slot; j--) o += S[cnt_base + j]; }
    else { for (int j = 0; j < slot; j++) o += S[cnt_base + j]; }
    o = min(o, pool);
    return SwSlot{o, min(S[cnt_base + slot], pool - o)};
}

// one thread per component and threshold: outer border (is_hole = 0) or hole border (is_hole = 1).
// lists[f][slot][k] = (raster-first pixel, pixel count of the hole | pixels of the holes the bright component encloses)
// Border points of the hole traces are kept while the border is followed the first time, in 128-byte chunks of a pool
// per frame and threshold (words 0..30: x | y << 16, word 31: previous chunk of the same border), so the accepted blobs
// need no second pass along their border; k_blob_median turns the points into distances.
constexpr int CH_PTS = 31;
// chunks per frame and threshold: RegionBuffers::maxch = clamp(h*w/256, 8192, 65535) (ids are stored as u16)
constexpr int MAXCHAIN = 512;          // chunks of one border the median kernel can index (15 872 points)
constexpr int PTS_STORED = 0x40000000; // blob_d[2 bi] = last chunk | PTS_STORED, else offset into the distance scratch
// distance scratch (double) per frame, shared by the 17 thresholds (bright blobs, fall-backs; a noisy frame needs nearly all
// of it at the lowest threshold): NTHR * RegionBuffers::maxdf entries, one counter (SW_ND + 0)
constexpr int CH_DIRECT = 16;          // chunk ids kept with the blob record: borders up to 496 points need no chain walk

struct StoreVisitor {
    StatVisitor sv;
    uint32_t *pool;
    int *counter;
    int maxch;
    unsigned short *ids;   // LDS, entry j of this lane at ids[j * 64]: the first CH_DIRECT chunks of the border
    int cur = -1, link = -1, fill = CH_PTS, nch = 0;
    int res_next = 0, res_end = 0;   // chunks reserved up front (one atomic per border in the common case)
    uint32_t b0 = 0, b1 = 0, b2 = 0;   // points wait in registers until four of them (16 aligned bytes) can go out together
    bool ok = true;
    __device__ __forceinline__ void reserve(int expected_points)
    {
        const int want = min((expected_points + CH_PTS - 1) / CH_PTS, CH_DIRECT);
        res_next = atomicAdd(counter, want);
        res_end = min(res_next + want, maxch);
    }
    __device__ __forceinline__ void point(int x, int y, bool vertex)
    {
        sv.point(x, y, vertex);
        if (!ok) return;
        if (fill == CH_PTS) {
            int c;
            if (res_next < res_end) c = res_next++;
            else c = (sv.npts <= MAXCHAIN * CH_PTS) ? atomicAdd(counter, 1) : maxch;
            if (c >= maxch) { ok = false; return; }
            link = cur;
            cur = c;
            fill = 0;
            if (nch < CH_DIRECT) ids[nch * 64] = (unsigned short)c;
            nch++;
        }
        const uint32_t p = (uint32_t)x | ((uint32_t)y << 16);
        const int slot = fill & 3;
        fill++;
        if (slot == 3) *reinterpret_cast<uint4 *>(pool + (size_t)cur * 32 + fill - 4) = make_uint4(b0, b1, b2, p);
        else if (fill == CH_PTS) *reinterpret_cast<uint4 *>(pool + (size_t)cur * 32 + 28) = make_uint4(b0, b1, p, (uint32_t)link);
        else if (slot == 0) b0 = p;
        else if (slot == 1) b1 = p;
        else b2 = p;
    }
    // points still in registers and the link word of the last chunk
    __device__ __forceinline__ void flush()
    {
        if (!ok || cur < 0 || fill == CH_PTS) return;
        uint32_t *c = pool + (size_t)cur * 32;
        const int r = fill & 3, base = fill - r;
        if (r > 0) c[base] = b0;
        if (r > 1) c[base + 1] = b1;
        if (r > 2) c[base + 2] = b2;
        c[31] = (uint32_t)link;
    }
    __device__ __forceinline__ bool stop() const { return false; }
};

// one border per lane; a few wavefronts per (frame, threshold) take the list entries in turns (a grid sized for the list
// capacity would be millions of empty workgroups; see frame_waves)
template <int is_hole>
__device__ __forceinline__ void blob_trace_one(int f, int slot, int k, int h, int w, const int2 *__restrict__ list,
                                               FrameState *__restrict__ st, int *__restrict__ S, BlobRec *__restrict__ blobs_all,
                                               int *__restrict__ blob_d_all, double *__restrict__ dists_all,
                                               const uint32_t *__restrict__ bits, uint32_t *__restrict__ pool_all,
                                               unsigned short *__restrict__ blob_ch_all, int maxch, int maxdf,
                                               unsigned long long *s_win, unsigned short *s_ids)
{
    const int2 e = list[k];
    const int root = e.x;
    // exact prunes: a hole's polygon area is >= its pixel count; a bright component's outer polygon contains the
    // unit squares of every pixel of every hole it encloses, so its area is >= their total pixel count
    if (e.y >= 5000) return;
    int *blob_d = blob_d_all + ((size_t)f * NTHR + slot) * MAXB * 2;
    double *dists = dists_all + (size_t)f * NTHR * maxdf;
    BlobRec *blobs = blobs_all + ((size_t)f * NTHR + slot) * MAXB;
    int y0 = root / w, x0 = root - y0 * w;
    if (is_hole) x0 -= 1;
    const int ws = bit_row_words(w);
    BitWin nz{bits + ((size_t)f * NTHR + slot) * h * ws, ws, h, s_win + threadIdx.x};   // binarised = cl > 50 + 10 slot
    const int max_steps = 4 * (w + h) + 65536;
    StoreVisitor tv;
    tv.pool = pool_all + ((size_t)f * NTHR + slot) * maxch * 32;
    tv.maxch = maxch;
    tv.counter = &S[SW_NC + slot];
    tv.ids = s_ids + threadIdx.x;
    tv.ok = is_hole != 0;   // bright components: few are accepted, their borders are followed again instead
    // a compact hole of n pixels has a border of about 4 sqrt(n) + 4 pixels; reserve for twice that
    if (is_hole) tv.reserve(8 * (int)sqrtf((float)e.y) + 16);
    StatVisitor &sv = tv.sv;
    bool ok = trace_border(nz, x0, y0, is_hole != 0, tv, max_steps);
    if (!ok) { set_overflow(st[f], OVF_TRACE); return; }
    tv.flush();
    sv.finish();
    double m00, m10, m01;
    moments_from_sums(sv.a00, sv.a10, sv.a01, m00, m10, m01);
    if (m00 < 10.0 || m00 >= 5000.0) return;
    if (m00 == 0.0) return;
    double cx = m10 / m00, cy = m01 / m00;
    int ix = (int)rint(cx), iy = (int)rint(cy);
    if (nz(ix, iy)) return;  // blobColor = 0: centre pixel must be dark (out-of-image cannot happen for a valid centroid)
    int bi = atomicAdd(&S[SW_NB + slot], 1);
    if (bi >= MAXB) { set_overflow(st[f], OVF_BLOBS); return; }
    if (tv.ok) {
        blob_d[bi * 2] = tv.cur | PTS_STORED;
        blob_d[bi * 2 + 1] = sv.npts;
        unsigned short *ch = blob_ch_all + (((size_t)f * NTHR + slot) * MAXB + bi) * CH_DIRECT;
        for (int j = 0; j < min(tv.nch, CH_DIRECT); j++) ch[j] = s_ids[j * 64 + threadIdx.x];
    } else {
        int doff = atomicAdd(&S[SW_ND], sv.npts);
        if ((long long)doff + sv.npts > (long long)NTHR * maxdf) { set_overflow(st[f], OVF_DISTS); blob_d[bi * 2] = -1; blob_d[bi * 2 + 1] = 0; }
        else {
            DistVisitor dv{cx, cy, dists + doff};
            trace_border(nz, x0, y0, is_hole != 0, dv, max_steps);
            blob_d[bi * 2] = doff;
            blob_d[bi * 2 + 1] = sv.npts;
        }
    }
    BlobRec &b = blobs[bi];
    b.x = cx; b.y = cy; b.r = 0;
    b.key = root;   // discovery position of the border in the raster scan
}

template <int is_hole>
__global__ __launch_bounds__(64) void k_blob_trace(const uint8_t *__restrict__ cl, int h, int w,
                                                   const int2 *__restrict__ lists, int cnt_base,
                                                   FrameState *__restrict__ st, int *__restrict__ sw, BlobRec *__restrict__ blobs_all,
                                                   int *__restrict__ blob_d_all, double *__restrict__ dists_all,
                                                   const uint32_t *__restrict__ bits, uint32_t *__restrict__ pool_all,
                                                   unsigned short *__restrict__ blob_ch_all, int maxch, int maxdf)
{
    __shared__ unsigned long long s_win[BW_ROWS * 64];
    __shared__ unsigned short s_ids[CH_DIRECT * 64];
    const int f = blockIdx.y, slot = blockIdx.z;
    int *S = sw + (size_t)f * SW_STRIDE;
    const int pool = sweep_pool(h, w, is_hole ? SWL_TRACE : SWL_BRIGHT);
    const SwSlot sl = sw_slot(S, cnt_base, slot, !is_hole, pool);
    const int cnt = sl.cnt;
    const int2 *list = lists + (size_t)f * pool + sl.off;
    for (int k = blockIdx.x * 64 + threadIdx.x; k < cnt; k += gridDim.x * 64)
        blob_trace_one<is_hole>(f, slot, k, h, w, list, st, S, blobs_all, blob_d_all, dists_all, bits, pool_all, blob_ch_all, maxch, maxdf,
                                s_win, s_ids);
}

// k-th smallest (0-based) of n values read through get(i) by one wavefront, for any n: the candidate set is narrowed by
// 64-bucket histograms (monotone bucket map over [min, max] of the candidates; the members of one bucket are exactly
// the candidates between that bucket's smallest and largest member) until at most SEL_CAP candidates are left, which
// are then ranked against each other in LDS.  O(n) reads per level, usually two levels.
constexpr int SEL_CAP = 256;
template <class Get>
__device__ double wave_select(Get get, int n, int k, int lane, double *s_buf /* SEL_CAP */, int *s_hist /* 64 */, int *s_cnt)
{
    double lo = -1e300, hi = 1e300;   // candidates: lo <= v <= hi
    for (;;) {
        double mn = 1e300, mx = -1e300;
        int cnt = 0, below = 0;
        for (int i = lane; i < n; i += 64) {
            const double v = get(i);
            if (v < lo) below++;
            else if (v <= hi) { cnt++; mn = fmin(mn, v); mx = fmax(mx, v); }
        }
        for (int off = 32; off >= 1; off >>= 1) {
            mn = fmin(mn, __shfl_xor(mn, off, 64)); mx = fmax(mx, __shfl_xor(mx, off, 64));
            cnt += __shfl_xor(cnt, off, 64); below += __shfl_xor(below, off, 64);
        }
        const int kk = k - below;   // rank among the candidates
        if (mx == mn) return mn;
        __syncthreads();
        if (cnt <= SEL_CAP) {
            if (lane == 0) *s_cnt = 0;
            __syncthreads();
            for (int i = lane; i < n; i += 64) {
                const double v = get(i);
                if (v >= lo && v <= hi) s_buf[atomicAdd(s_cnt, 1)] = v;
            }
            __syncthreads();
            double found = -1e300;
            for (int j = lane; j < cnt; j += 64) {
                const double x = s_buf[j];
                int less = 0, leq = 0;
                for (int q = 0; q < cnt; q++) { const double y = s_buf[q]; less += (y < x) ? 1 : 0; leq += (y <= x) ? 1 : 0; }
                if (less <= kk && kk < leq) found = x;
            }
            for (int off = 32; off >= 1; off >>= 1) found = fmax(found, __shfl_xor(found, off, 64));
            return found;
        }
        const double scale = 64.0 / (mx - mn);
        s_hist[lane] = 0;
        __syncthreads();
        for (int i = lane; i < n; i += 64) {
            const double v = get(i);
            if (v >= lo && v <= hi) atomicAdd(&s_hist[min(63, (int)((v - mn) * scale))], 1);
        }
        __syncthreads();
        const int hc = s_hist[lane];
        int incl = hc;
        for (int off = 1; off < 64; off <<= 1) { int tt = __shfl_up(incl, off, 64); if (lane >= off) incl += tt; }
        const int bstar = __popcll(__ballot(incl <= kk));
        // the bucket's members: candidates whose bucket index is bstar = the candidates between its extreme members
        double nlo = 1e300, nhi = -1e300;
        for (int i = lane; i < n; i += 64) {
            const double v = get(i);
            if (v >= lo && v <= hi && min(63, (int)((v - mn) * scale)) == bstar) { nlo = fmin(nlo, v); nhi = fmax(nhi, v); }
        }
        for (int off = 32; off >= 1; off >>= 1) { nlo = fmin(nlo, __shfl_xor(nlo, off, 64)); nhi = fmax(nhi, __shfl_xor(nhi, off, 64)); }
        lo = nlo; hi = nhi;
    }
}

// (d[(n-1)/2] + d[n/2]) / 2 of the sorted values
template <class Get>
__device__ double wave_median(Get get, int n, int lane, double *s_buf, int *s_hist, int *s_cnt)
{
    const int k1 = (n - 1) / 2, k2 = n / 2;
    const double v1 = wave_select(get, n, k1, lane, s_buf, s_hist, s_cnt);
    double v2 = v1;
    if (k2 != k1) {
        int le = 0;
        double nxt = 1e300;
        for (int i = lane; i < n; i += 64) {
            const double v = get(i);
            if (v <= v1) le++;
            else nxt = fmin(nxt, v);
        }
        for (int off = 32; off >= 1; off >>= 1) { le += __shfl_xor(le, off, 64); nxt = fmin(nxt, __shfl_xor(nxt, off, 64)); }
        if (le <= k2) v2 = nxt;   // the (k2)-th value is the next larger one
    }
    return (v1 + v2) / 2.;
}

// radius = median distance of the border points from the centre: one wavefront per blob.
// The square root is monotone, so the two middle distances are the roots of the two middle squared distances.
// Common case (border of up to 496 points, ids of its chunks in the blob record): the squared distances stay in
// registers, a 64-bucket histogram over [min, max] (monotone bucket map) finds the bucket that holds rank k, and only
// that bucket's few members are compared with each other -- O(n) instead of the O(n^2) rank count of the general path.
constexpr int MED_FAST = CH_DIRECT * CH_PTS;

// part 0: the blobs of the hole borders [0, NA), part 1: those of the bright components [NA, NB)
__global__ __launch_bounds__(64) void k_blob_median(int part, int *__restrict__ sw, BlobRec *__restrict__ blobs_all,
                                                    const int *__restrict__ blob_d_all, double *__restrict__ dists_all,
                                                    const uint32_t *__restrict__ pool_all,
                                                    const unsigned short *__restrict__ blob_ch_all, FrameState *__restrict__ st,
                                                    int maxch, int maxdf)
{
    __shared__ double s_d[MED_FAST > SEL_CAP ? MED_FAST : SEL_CAP];
    __shared__ int s_ch[MAXCHAIN];
    __shared__ int s_hist[64];
    __shared__ int s_cnt;
    const int f = blockIdx.y, slot = blockIdx.z, lane = threadIdx.x;
    int *S = sw + (size_t)f * SW_STRIDE;
    // part 0 runs before any bright component's blob is appended: the blobs listed so far are the hole borders' (SW_NA)
    const int nb = min(S[SW_NB + slot], MAXB);
    if (!part && blockIdx.x == 0 && lane == 0) S[SW_NA + slot] = nb;
    BlobRec *blobs = blobs_all + ((size_t)f * NTHR + slot) * MAXB;
    const int *blob_d = blob_d_all + ((size_t)f * NTHR + slot) * MAXB * 2;
    double *dists = dists_all + (size_t)f * NTHR * maxdf;
    const uint32_t *pool = pool_all + ((size_t)f * NTHR + slot) * maxch * 32;
    const unsigned short *blob_ch = blob_ch_all + ((size_t)f * NTHR + slot) * MAXB * CH_DIRECT;
    for (int bi = (part ? S[SW_NA + slot] : 0) + blockIdx.x; bi < nb; bi += gridDim.x) {
        const int code = blob_d[bi * 2], n = blob_d[bi * 2 + 1];
        if (code < 0 || n <= 0) continue;
        double r;
        if ((code & PTS_STORED) && n <= MED_FAST) {
            const double cx = blobs[bi].x, cy = blobs[bi].y;
            const int myid = lane < CH_DIRECT ? (int)blob_ch[bi * CH_DIRECT + lane] : 0;
            double v[MED_FAST / 64 + 1];
            double mn = 1e300, mx = -1.0;
#pragma unroll
            for (int m = 0; m < MED_FAST / 64 + 1; m++) {
                const int i = lane + 64 * m;
                const bool valid = i < n;
                const int c = __shfl(myid, valid ? i / CH_PTS : 0, 64);
                v[m] = -1.0;
                if (valid) {
                    const uint32_t p = pool[(size_t)c * 32 + i % CH_PTS];
                    const double dx = cx - (double)(int)(p & 0xFFFFu), dy = cy - (double)(int)(p >> 16);
                    v[m] = dx * dx + dy * dy;
                    mn = fmin(mn, v[m]); mx = fmax(mx, v[m]);
                }
            }
            for (int off = 32; off >= 1; off >>= 1) { mn = fmin(mn, __shfl_xor(mn, off, 64)); mx = fmax(mx, __shfl_xor(mx, off, 64)); }
            double res[2] = {mn, mn};
            if (mx > mn) {
                const double scale = 64.0 / (mx - mn);
                int bkt[MED_FAST / 64 + 1];
                __syncthreads();
                s_hist[lane] = 0;
                __syncthreads();
#pragma unroll
                for (int m = 0; m < MED_FAST / 64 + 1; m++) {
                    bkt[m] = -1;
                    if (v[m] >= 0.0) { bkt[m] = min(63, (int)((v[m] - mn) * scale)); atomicAdd(&s_hist[bkt[m]], 1); }
                }
                __syncthreads();
                const int hc = s_hist[lane];
                int incl = hc;
                for (int off = 1; off < 64; off <<= 1) { int tt = __shfl_up(incl, off, 64); if (lane >= off) incl += tt; }
                const int ks[2] = {(n - 1) / 2, n / 2};
                for (int which = 0; which < 2; which++) {
                    if (which == 1 && ks[1] == ks[0]) { res[1] = res[0]; break; }
                    const int k = ks[which];
                    const int bstar = __popcll(__ballot(incl <= k));          // first bucket whose running count exceeds k
                    const int kk = k - __shfl(incl - hc, bstar, 64);          // rank inside that bucket
                    const int sz = __shfl(hc, bstar, 64);
                    __syncthreads();
                    if (lane == 0) s_cnt = 0;
                    __syncthreads();
#pragma unroll
                    for (int m = 0; m < MED_FAST / 64 + 1; m++)
                        if (bkt[m] == bstar) s_d[atomicAdd(&s_cnt, 1)] = v[m];
                    __syncthreads();
                    double found = -1.0;
                    for (int j = lane; j < sz; j += 64) {
                        const double x = s_d[j];
                        int less = 0, leq = 0;
                        for (int q = 0; q < sz; q++) { const double y = s_d[q]; less += (y < x) ? 1 : 0; leq += (y <= x) ? 1 : 0; }
                        if (less <= kk && kk < leq) found = x;
                    }
                    for (int off = 32; off >= 1; off >>= 1) found = fmax(found, __shfl_xor(found, off, 64));
                    res[which] = found;
                }
            }
            r = (sqrt(res[0]) + sqrt(res[1])) / 2.;
        } else if (code & PTS_STORED) {
            // long border (> 496 points): chunk ids by walking the chain, distances recomputed from the points on every read
            const int nch = (n + CH_PTS - 1) / CH_PTS;   // <= MAXCHAIN (StoreVisitor stops storing beyond that)
            __syncthreads();
            if (lane == 0) {
                int c = code & (PTS_STORED - 1);
                for (int j = nch - 1; j >= 0; j--) { s_ch[j] = c; c = (int)pool[(size_t)c * 32 + 31]; }
            }
            __syncthreads();
            const double cx = blobs[bi].x, cy = blobs[bi].y;
            auto get = [&](int i) {
                const uint32_t p = pool[(size_t)s_ch[i / CH_PTS] * 32 + i % CH_PTS];
                const double dx = cx - (double)(int)(p & 0xFFFFu), dy = cy - (double)(int)(p >> 16);
                return sqrt(dx * dx + dy * dy);
            };
            r = wave_median(get, n, lane, s_d, s_hist, &s_cnt);
        } else {
            const double *dd = dists + code;   // border followed a second time (bright components, pool overflow)
            r = wave_median([&](int i) { return dd[i]; }, n, lane, s_d, s_hist, &s_cnt);
        }
        if (lane == 0) blobs[bi].r = r;
    }
}

// for every threshold in ascending order: order its blobs like cv2.findContours returns contours (latest discovery
// first) and merge them into the groups (SimpleBlobDetector::detect inner loops); one workgroup per frame.
//
// The merge is sequential by definition (a centre joins the FIRST group whose middle centre is close, and joining
// moves that middle centre), so it is run in batches of 64 blobs:
//   1. all threads: first matching group of every blob of the batch against the middle centres as they are at the
//      start of the batch (LDS), no barrier between blobs;
//   2. the centre lists of the groups found are fetched from HBM together (one round trip per batch);
//   3. one wavefront replays the batch in order.  Only groups changed earlier in the batch can alter a blob's answer:
//      if its group is unchanged, the changed groups with a smaller index are re-tested; if its group was changed, the
//      blob is searched again.  Insertions work on the LDS copies (one lane per list element);
//   4. changed lists go back to HBM.
// Result and order of operations per blob are those of the sequential loop.
__device__ __forceinline__ bool blob_joins(double gx, double gy, double gr, double cx, double cy, double cr)
{
    const double dx = gx - cx, dy = gy - cy;
    const double d2 = dx * dx + dy * dy;
    // sqrt(d2) >= a certainly holds when d2 >= a * a * (1 + 2^-40): only the others need the root
    const double a = fmax(fmax(10.0, gr), cr);
    if (d2 >= a * a * 1.0000000000009095) return false;
    const double dist = sqrt(d2);
    const bool isNew = dist >= 10.0 && dist >= gr && dist >= cr;
    return !isNew;
}

__device__ __forceinline__ int wave_min_int(int v)
{
    for (int off = 32; off >= 1; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
    return v;
}

// The first-match search does not walk the groups: their middle centres are kept in a uniform grid over the frame (cells of
// 64 x 64 pixels, one singly linked list of group indices per cell; a group moves when an insertion moves its middle centre
// into another cell).  A blob joins a group only if dist < max(10, r_group, r_blob), so its candidates lie in the cells within
// R = max(10, r_blob, largest middle-centre radius so far) + 1 of it -- 3 x 3 cells on a clean frame -- and the first match is
// the smallest index among the candidates that pass the same test as before.  A frame with 20 000 blob groups (heavy sensor
// noise) used to cost 0.2 - 1.5 s of all-pairs tests here; the search is now proportional to the local density.
// LDS: 78 KB per workgroup, so two frames share a CU (a call of 400 - 450 images is one round of workgroups instead of two).
// Only the centre lists of groups with at most MG_STAGE centres are staged in LDS for a batch (a clean frame's groups end
// with 17, one per threshold); longer ones are updated in place in HBM by the in-order path.
constexpr int MG_NT = 256;     // threads of k_blob_merge (4 wavefronts: cell ranges, pair tests and list traffic are split over them)
constexpr int MG_CELLS = 1024; // grid cells (the cell edge doubles until the frame fits: 64 px at 1920 x 1200, 128 px at 3840 x 2160)
constexpr int MG_LCAP = 24;    // centres per staged list
constexpr int MG_STAGE = MG_LCAP - 4;
constexpr int MG_RANK_DIRECT = 4096, MG_RANK_BUCKETS = 4096;   // blobs of a threshold ranked all-pairs in LDS up to this many; key buckets beyond
__global__ __launch_bounds__(MG_NT) void k_blob_merge(FrameState *__restrict__ st, const int *__restrict__ sw,
                                                    const BlobRec *__restrict__ blobs_all, int *__restrict__ order,
                                                    Group *__restrict__ groups, int always_replay, double *__restrict__ gmid_all,
                                                    int h, int w)
{
    // location + radius of each group's middle centre (what the tests read)
    __shared__ double sX[MAXG_LDS], sY[MAXG_LDS], sR[MAXG_LDS];   // groups >= MAXG_LDS: gmid (HBM, this workgroup only)
    __shared__ int s_next[MAXG_LDS];                               // next group in the same grid cell (-1: none)
    __shared__ unsigned char s_gn[MAXG_LDS];                       // centres in the group's list (G[].n, saves a round trip per batch)
    __shared__ int s_head[MG_CELLS];
    __shared__ double s_maxr;                                      // largest radius a middle centre ever had (search range bound)
    __shared__ double bX[64], bY[64], bR[64];
    __shared__ double pl[MG_LCAP * 3][64];   // centre lists of the groups the batch touches: [element][slot], lanes = slots
    __shared__ double cX[64], cY[64], cR[64];   // middle centre each blob's group gets if the blob is inserted
    __shared__ int cG[64], s_bad[64];
    __shared__ int pn[64], pg[64], pd[64], s_jm[64], s_mod[64];
    __shared__ int s_ng, s_serial, s_adv, s_maxn;
    const int f = blockIdx.x, t = threadIdx.x, lane = t & 63;
    FrameState &S = st[f];
    int *ord = order + (size_t)f * 2 * MAXB;   // [0, MAXB): blob indices in cv2.findContours order; [MAXB, 2 MAXB): scratch of the bucketed ranking
    Group *G = groups + (size_t)f * MAXG;
    double *gm = gmid_all + (size_t)f * (MAXG - MAXG_LDS) * 4 - (size_t)MAXG_LDS * 4;   // gm[4 j ..] = x, y, r, next for j >= MAXG_LDS
    // HBM entries are written by other wavefronts of this workgroup between barriers: read past the vector L1
    auto gm_load = [&](int j, int k) { return __hip_atomic_load(gm + 4 * (size_t)j + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto mid_joins = [&](int j, double cx, double cy, double cr) {
        if (j < MAXG_LDS) return blob_joins(sX[j], sY[j], sR[j], cx, cy, cr);
        return blob_joins(gm_load(j, 0), gm_load(j, 1), gm_load(j, 2), cx, cy, cr);
    };
    auto mid_set = [&](int j, double x, double y, double r) {
        if (j < MAXG_LDS) { sX[j] = x; sY[j] = y; sR[j] = r; }
        else { gm[4 * (size_t)j] = x; gm[4 * (size_t)j + 1] = y; gm[4 * (size_t)j + 2] = r; }
    };
    // ---- the grid over the middle centres
    int cs = 64;
    while (((w + cs - 1) / cs) * ((h + cs - 1) / cs) > MG_CELLS) cs *= 2;
    const int gw = (w + cs - 1) / cs, gh = (h + cs - 1) / cs;
    const double inv_cs = 1.0 / cs;                       // a power of two: the scaling is exact
    auto cell_x = [&](double x) { return min(max((int)floor(x * inv_cs), 0), gw - 1); };
    auto cell_y = [&](double y) { return min(max((int)floor(y * inv_cs), 0), gh - 1); };
    auto cell_of = [&](double x, double y) { return cell_y(y) * gw + cell_x(x); };
    auto mid_cell = [&](int j) { return j < MAXG_LDS ? cell_of(sX[j], sY[j]) : cell_of(gm_load(j, 0), gm_load(j, 1)); };
    auto nxt_get = [&](int j) { return j < MAXG_LDS ? s_next[j] : (int)gm_load(j, 3); };
    auto nxt_set = [&](int j, int v) { if (j < MAXG_LDS) s_next[j] = v; else gm[4 * (size_t)j + 3] = (double)v; };
    auto grid_link = [&](int j, int c) { nxt_set(j, atomicExch(&s_head[c], j)); };   // any number of lanes of ONE wavefront at a time
    auto grid_unlink = [&](int j, int c) {                                            // one lane at a time
        int p = s_head[c];
        if (p == j) { s_head[c] = nxt_get(j); return; }
        for (int it = 0; p >= 0 && it < MAXG; it++) {   // (a list never holds more than every group: the walk ends whatever it reads)
            const int q = nxt_get(p);
            if (q == j) { nxt_set(p, nxt_get(j)); return; }
            p = q;
        }
    };
    // smallest group index < limit among the candidates in this caller's share of the cells around (cx, cy) that the blob
    // joins (INT_MAX: none); the cells of the range are dealt out in turns: share `part` of `nsplit`
    auto grid_first = [&](double cx, double cy, double cr, int limit, int part, int nsplit) {
        const double R = fmax(fmax(10.0, cr), s_maxr) + 1.0;
        const int x0 = cell_x(cx - R), x1 = cell_x(cx + R), y0 = cell_y(cy - R), y1 = cell_y(cy + R);
        int first = INT_MAX, k = 0;
        for (int yy = y0; yy <= y1; yy++)
            for (int xx = x0; xx <= x1; xx++, k++) {
                if (k % nsplit != part) continue;
                int it = 0;
                for (int g = s_head[yy * gw + xx]; g >= 0 && it < MAXG; g = nxt_get(g), it++)
                    if (g < limit && g < first && mid_joins(g, cx, cy, cr)) first = g;
            }
        return first;
    };
    for (int c = t; c < MG_CELLS; c += MG_NT) s_head[c] = -1;
    if (t == 0) s_maxr = 0.0;
    __syncthreads();
    int ng = 0;
    for (int thr = 0; thr < NTHR; thr++) {
        const int nb = min(sw[(size_t)f * SW_STRIDE + SW_NB + thr], MAXB);
        const BlobRec *B = blobs_all + ((size_t)f * NTHR + thr) * MAXB;
        if (nb <= MG_RANK_DIRECT && !(always_replay & 2)) {
            // rank by key (keys are distinct); the keys are staged in the list buffer, which is idle between thresholds
            int *keys = reinterpret_cast<int *>(&pl[0][0]);
            for (int i = t; i < nb; i += MG_NT) keys[i] = B[i].key;
            __syncthreads();
            for (int i = t; i < nb; i += MG_NT) {
                const int ki = keys[i];
                int rank = 0;
                for (int j = 0; j < nb; j++) rank += (keys[j] > ki) ? 1 : 0;
                ord[rank] = i;
            }
        } else {
            // a threshold of a noisy frame (up to 32768 blobs: the all-pairs count was 40 ms of key reads per such threshold):
            // bucket the keys by value (they are pixel indices), count, prefix from the top bucket down, scatter the blob
            // indices bucket by bucket into the scratch half of `order`, and rank every key among its bucket's few members
            int *cntb = reinterpret_cast<int *>(&pl[0][0]), *curb = cntb + MG_RANK_BUCKETS;
            int *scr = ord + MAXB;
            const int bw = (int)(((size_t)h * w + MG_RANK_BUCKETS - 1) / MG_RANK_BUCKETS);
            for (int b = t; b < MG_RANK_BUCKETS; b += MG_NT) cntb[b] = 0;
            __syncthreads();
            for (int i = t; i < nb; i += MG_NT) atomicAdd(&cntb[min(B[i].key / bw, MG_RANK_BUCKETS - 1)], 1);
            __syncthreads();
            {   // curb[b] = number of keys in the buckets above b (the rank of a bucket's largest key): a thread sums its run
                // of buckets from the top down, the runs are chained through s_mod (64 ints, idle here) wave by wave
                constexpr int PER = MG_RANK_BUCKETS / MG_NT;
                const int b_hi = MG_RANK_BUCKETS - 1 - PER * t;          // this thread's buckets: b_hi, b_hi - 1, ..., b_hi - PER + 1
                int sum = 0;
                for (int k = 0; k < PER; k++) sum += cntb[b_hi - k];
                int incl = sum;
                for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off, 64); if (lane >= off) incl += o; }
                if (lane == 63) s_mod[t >> 6] = incl;
                __syncthreads();
                int base = incl - sum;
                for (int k = 0; k < (t >> 6); k++) base += s_mod[k];
                for (int k = 0; k < PER; k++) { curb[b_hi - k] = base; base += cntb[b_hi - k]; }
            }
            __syncthreads();
            for (int i = t; i < nb; i += MG_NT) scr[atomicAdd(&curb[min(B[i].key / bw, MG_RANK_BUCKETS - 1)], 1)] = i;
            __threadfence_block();
            __syncthreads();
            for (int i = t; i < nb; i += MG_NT) {
                const int ki = B[i].key, b = min(ki / bw, MG_RANK_BUCKETS - 1);
                const int end = curb[b], start = end - cntb[b];          // the cursor now stands behind the bucket's last entry
                int rank = start;
                for (int p = start; p < end; p++)
                    rank += (B[__hip_atomic_load(scr + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)].key > ki) ? 1 : 0;
                ord[rank] = i;
            }
        }
        __syncthreads();
        const int ng0 = ng;   // centres of this threshold are only compared with groups of the earlier ones
        for (int q0 = 0; q0 < nb;) {
            const int qn = min(64, nb - q0);
            if (t < 64) {
                if (t < qn) { const BlobRec &c = B[ord[q0 + t]]; bX[t] = c.x; bY[t] = c.y; bR[t] = c.r; }
                s_jm[t] = INT_MAX; pg[t] = -1; pd[t] = 0;
            }
            __syncthreads();
            // 1. first match against the state at the start of the batch: lane = blob, the wavefronts take the cells around
            //    the blob in turns
            if (ng0 > 0) {
                const bool act = lane < qn;
                if (act) {
                    const int first = grid_first(bX[lane], bY[lane], bR[lane], ng0, t >> 6, MG_NT / 64);
                    if (first != INT_MAX) atomicMin(&s_jm[lane], first);
                }
            }
            __syncthreads();
            // 2. one LDS slot per distinct group (slot = first blob of the batch that found it), filled together
            if (t < 64) {
                const int g = t < qn ? s_jm[t] : INT_MAX;
                bool first = g != INT_MAX;
                for (int dd = 0; dd < 64; dd++) {
                    const int o = __shfl(g, dd, 64);
                    if (dd < t && o == g) first = false;
                }
                if (first) pg[t] = g;
            }
            __syncthreads();
            // list lengths first (LDS; one round trip for the groups beyond it), then only the entries in use, every thread's
            // loads in flight together.  A list too long for the staging area is not staged: its blob is then not "clean",
            // and the in-order path updates the list where it lies
            if (t < 64) {
                int g = pg[t];
                int gn = g >= 0 ? (g < MAXG_LDS ? (int)s_gn[g] : G[g].n) : 0;
                if (gn > MG_STAGE) { pg[t] = -1; g = -1; gn = 0; }
                pn[t] = gn;
                int mx = gn;
                for (int off = 32; off >= 1; off >>= 1) mx = max(mx, __shfl_xor(mx, off, 64));
                if (t == 0) s_maxn = mx;
            }
            __syncthreads();
            {
                const int per = 3 * s_maxn;                      // doubles per slot to look at (<= MG_STAGE * 3)
                for (int idx0 = 0; idx0 < 64 * per; idx0 += 4 * MG_NT) {
                    double v[4];
                    int qq[4], ee[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int idx = idx0 + u * MG_NT + t;
                        const int q = idx / per, e = idx - q * per;
                        const bool on = idx < 64 * per && pg[q] >= 0 && e < 3 * pn[q];
                        qq[u] = on ? q : -1; ee[u] = e;
                        v[u] = on ? G[pg[q]].c[e / 3][e % 3] : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        if (qq[u] >= 0) pl[ee[u]][qq[u]] = v[u];
                }
            }
            __syncthreads();
            // 3. Blobs of one batch rarely interact.  A blob is "clean" if no earlier blob of the batch found the same
            //    group and no earlier blob's group, with the middle centre it gets, would take this blob first (tested
            //    pairwise, lane = blob).  All blobs before the first unclean one insert themselves at once (lane per
            //    blob) and the next batch starts AT that blob -- as lane 0 it sees the updated state and is clean by
            //    definition.  If the clean prefix is short the batch is replayed in order instead (code below).
            const bool act = lane < qn;
            const int jm = act ? s_jm[lane] : INT_MAX;
            const double cx = act ? bX[lane] : 0, cy = act ? bY[lane] : 0, cr = act ? bR[lane] : 0;
            int gn = 0, pos = 0;
            bool ins = false, full = false;
            double nx = 0, ny = 0, nr = 0;
            if (t < 64) {
                const bool own = jm != INT_MAX && pg[lane] == jm;   // first blob of the batch with this group: slot = lane
                if (own) {
                    gn = pn[lane];
                    full = gn >= GCAP;
                    ins = !full;
                    if (ins) {
                        for (int k = 0; k < gn; k++) pos += !(cr < pl[3 * k + 2][lane]) ? 1 : 0;
                        const int m = (gn + 1) / 2, src = m < pos ? m : m - 1;
                        if (m == pos) { nx = cx; ny = cy; nr = cr; }
                        else { nx = pl[3 * src][lane]; ny = pl[3 * src + 1][lane]; nr = pl[3 * src + 2][lane]; }
                    }
                }
                cG[lane] = ins ? jm : -1; cX[lane] = nx; cY[lane] = ny; cR[lane] = nr;
                s_bad[lane] = (act && jm != INT_MAX && !own) ? 1 : 0;
            }
            __syncthreads();
            {   // pair tests: wavefront v checks the blobs dd = 4v .. 4v+3 against every later blob (lane)
                bool hit = false;
                const int d0 = (t >> 6) * (64 / (MG_NT / 64));
                for (int dd = d0; dd < d0 + 64 / (MG_NT / 64); dd++) {
                    const int og = cG[dd];
                    if (dd < lane && act && og >= 0 && og < jm && blob_joins(cX[dd], cY[dd], cR[dd], cx, cy, cr)) hit = true;
                }
                if (hit) s_bad[lane] = 1;
            }
            __syncthreads();
            if (t < 64) {
                const unsigned long long bb = __ballot(s_bad[lane] != 0);
                const int clean = bb ? __ffsll((long long)bb) - 1 : qn;   // >= 1: lane 0 has no earlier blob
                const bool serial = (always_replay & 1) || (clean < qn && clean < 8);
                if (!serial) {
                    const bool mine = lane < clean;
                    int link_g = -1, link_c = 0, old_c = -1;   // grid: group to (re)link into cell link_c; old_c >= 0: it leaves that cell
                    double newr = 0.0;
                    if (mine && ins) {
                        for (int k = gn - 1; k >= pos; k--) {
                            pl[3 * k + 3][lane] = pl[3 * k][lane]; pl[3 * k + 4][lane] = pl[3 * k + 1][lane]; pl[3 * k + 5][lane] = pl[3 * k + 2][lane];
                        }
                        pl[3 * pos][lane] = cx; pl[3 * pos + 1][lane] = cy; pl[3 * pos + 2][lane] = cr;
                        pn[lane] = gn + 1; pd[lane] = 1;
                        if (jm < MAXG_LDS) s_gn[jm] = (unsigned char)(gn + 1);
                        const int c0 = mid_cell(jm), c1 = cell_of(nx, ny);
                        if (c0 != c1) { link_g = jm; link_c = c1; old_c = c0; }
                        mid_set(jm, nx, ny, nr);
                        newr = nr;
                    }
                    if (mine && full) set_overflow(S, OVF_GROUPS);
                    const bool fresh = mine && jm == INT_MAX;
                    const unsigned long long fb = __ballot(fresh);
                    const int gi = ng + __popcll(fb & ((1ull << lane) - 1ull));
                    if (fresh) {
                        if (gi < MAXG) {
                            Group &g = G[gi];
                            g.n = 1;
                            g.c[0][0] = cx; g.c[0][1] = cy; g.c[0][2] = cr;
                            mid_set(gi, cx, cy, cr);
                            if (gi < MAXG_LDS) s_gn[gi] = 1;
                            link_g = gi; link_c = cell_of(cx, cy);
                            newr = cr;
                        } else set_overflow(S, OVF_GROUPS);
                    }
                    // grid upkeep by this one wavefront: the groups that change cell leave their lists one after the other,
                    // then every lane with a moved or a new group links it in (atomic exchange on the cell head)
                    for (unsigned long long mb = __ballot(old_c >= 0); mb; mb &= mb - 1ull)
                        if (lane == __ffsll((long long)mb) - 1) grid_unlink(link_g, old_c);
                    if (link_g >= 0) grid_link(link_g, link_c);
                    for (int off = 32; off >= 1; off >>= 1) newr = fmax(newr, __shfl_xor(newr, off, 64));
                    if (t == 0 && newr > s_maxr) s_maxr = newr;
                    ng = min(ng + __popcll(fb), MAXG);
                    if (t == 0) { s_ng = ng; s_adv = clean; }
                } else if (t == 0) s_adv = qn;
                if (t == 0) s_serial = serial ? 1 : 0;
            }
            __syncthreads();
            if (t < 64 && s_serial) {
                int nm = 0;
                for (int q = 0; q < qn; q++) {
                    // (a staged list is full -- a fifth blob of this batch for one group: the batch ends in front of this blob,
                    //  the next one finds the list too long to stage and updates it in HBM)
                    const double cx = bX[q], cy = bY[q], cr = bR[q];
                    int jm = s_jm[q];
                    const int mj = lane < nm ? s_mod[lane] : -1;
                    const bool changed = jm != INT_MAX && __ballot(mj == jm) != 0ull;
                    if (changed) {
                        jm = wave_min_int(grid_first(cx, cy, cr, ng0, lane, 64));   // the cells around the blob, one per lane
                    } else {
                        int cand = INT_MAX;
                        if (mj >= 0 && mj < jm && mid_joins(mj, cx, cy, cr)) cand = mj;
                        jm = min(jm, wave_min_int(cand));
                    }
                    if (jm != INT_MAX) {
                        const unsigned long long sb = __ballot(pg[lane] == jm);
                        const int slot = sb ? __ffsll((long long)sb) - 1 : -1;
                        Group &g = G[jm];
                        const int gn = slot >= 0 ? pn[slot] : g.n;
                        if (slot >= 0 && gn >= MG_LCAP) {
                            if (lane == 0) s_adv = q;
                            break;
                        }
                        if (gn < GCAP) {
                            // insertion behind the last centre whose radius is not larger (the list is sorted by radius)
                            double ex = 0, ey = 0, er = 0;
                            if (slot >= 0) {
                                if (lane < MG_LCAP) { ex = pl[3 * lane][slot]; ey = pl[3 * lane + 1][slot]; er = pl[3 * lane + 2][slot]; }
                            } else if (lane < GCAP) { ex = g.c[lane][0]; ey = g.c[lane][1]; er = g.c[lane][2]; }
                            const int pos = __popcll(__ballot(lane < gn && !(cr < er)));
                            const int m = (gn + 1) / 2;
                            const int src = m < pos ? m : m - 1;   // element of the old list that becomes the middle one
                            double sx = __shfl(ex, src & 63, 64), sy = __shfl(ey, src & 63, 64), sr = __shfl(er, src & 63, 64);
                            if (m == pos) { sx = cx; sy = cy; sr = cr; }
                            __builtin_amdgcn_wave_barrier();   // all lanes hold their element: now overwrite (LDS ops of a wave are in order)
                            if (slot >= 0) {
                                if (lane >= pos && lane < gn) { pl[3 * lane + 3][slot] = ex; pl[3 * lane + 4][slot] = ey; pl[3 * lane + 5][slot] = er; }
                                if (lane == 0) {
                                    pl[3 * pos][slot] = cx; pl[3 * pos + 1][slot] = cy; pl[3 * pos + 2][slot] = cr;
                                    pn[slot] = gn + 1; pd[slot] = 1;
                                }
                            } else {
                                if (lane >= pos && lane < gn) { g.c[lane + 1][0] = ex; g.c[lane + 1][1] = ey; g.c[lane + 1][2] = er; }
                                if (lane == 0) { g.c[pos][0] = cx; g.c[pos][1] = cy; g.c[pos][2] = cr; g.n = gn + 1; }
                            }
                            if (lane == 0) {
                                if (jm < MAXG_LDS) s_gn[jm] = (unsigned char)(gn + 1);
                                const int c0 = mid_cell(jm), c1 = cell_of(sx, sy);
                                if (c0 != c1) { grid_unlink(jm, c0); grid_link(jm, c1); }
                                mid_set(jm, sx, sy, sr);
                                if (sr > s_maxr) s_maxr = sr;
                            }
                            if (__ballot(mj == jm) == 0ull) {   // first change of this group in the batch
                                if (lane == 0) s_mod[nm] = jm;
                                nm++;
                            }
                        } else if (lane == 0) set_overflow(S, OVF_GROUPS);
                    } else if (ng < MAXG) {
                        if (lane == 0) {
                            Group &g = G[ng];
                            g.n = 1;
                            g.c[0][0] = cx; g.c[0][1] = cy; g.c[0][2] = cr;
                            mid_set(ng, cx, cy, cr);
                            if (ng < MAXG_LDS) s_gn[ng] = 1;
                            grid_link(ng, cell_of(cx, cy));
                            if (cr > s_maxr) s_maxr = cr;
                        }
                        ng++;
                    } else if (lane == 0) set_overflow(S, OVF_GROUPS);
                    __builtin_amdgcn_wave_barrier();
                }
                if (t == 0) s_ng = ng;
            }
            if (t < 64) {   // longest changed list (same wavefront as the two insertion paths above)
                int mx = (pg[lane] >= 0 && pd[lane]) ? pn[lane] : 0;
                for (int off = 32; off >= 1; off >>= 1) mx = max(mx, __shfl_xor(mx, off, 64));
                if (t == 0) s_maxn = mx;
            }
            __syncthreads();
            ng = s_ng;
            // 4. changed lists back to HBM
            {
                const int per = 3 * s_maxn + 1;                  // + 1: the length
                for (int idx = t; idx < 64 * per && per > 1; idx += MG_NT) {
                    const int q = idx / per, e = idx - q * per;
                    const int g = pg[q];
                    if (g >= 0 && pd[q]) {
                        if (e == per - 1) G[g].n = pn[q];
                        else if (e < pn[q] * 3) G[g].c[e / 3][e % 3] = pl[e][q];
                    }
                }
            }
            q0 += s_adv;
            __syncthreads();
        }
        __syncthreads();
    }
    if (t == 0) S.n_groups = ng;
}

// Enclosed-hole pixel totals of the bright components of every threshold, after both sweeps:
//   encl[component (at that threshold) of the pixel west of the hole's first pixel] += |hole|
// The bright forest of step t is read back from its merge history: an entry that stopped being a root at step j
// carries (hpar = the root that absorbed it, htime = j); following the links with htime <= t from any pixel of the
// step-t bright set ends at its root of step t.  One workgroup per frame walks the 17 thresholds; the totals live in
// the accumulator plane only between the two barriers of a threshold (atomics both ways: no stale L1 lines).
#define ENC_NT 1024
__global__ __launch_bounds__(ENC_NT) void k_enclosed_all(const int2 *__restrict__ hl, int2 *__restrict__ bl, const int *__restrict__ sw,
                                                         const int *__restrict__ forest /* bright: {parent, history} per pixel */,
                                                         int h, int w, int *__restrict__ encl)
{
    const size_t N = (size_t)h * w, f = blockIdx.x;
    const int *S = sw + f * SW_STRIDE;
    const unsigned *hist = reinterpret_cast<const unsigned *>(forest) + f * N * 2 + 1;   // hist[2 c]: (absorbing root) | (step << 24)
    int *ef = encl + f * N;
    const int pool_h = sweep_pool(h, w, SWL_DARK), pool_l = sweep_pool(h, w, SWL_BRIGHT);
    for (int slot = 0; slot < NTHR; slot++) {
        const int t = NTHR - 1 - slot;                     // bright step of this threshold
        const SwSlot sh = sw_slot(S, SW_NH, slot, false, pool_h), sl = sw_slot(S, SW_NL, slot, true, pool_l);
        const int nh = sh.cnt, nl = sl.cnt;
        for (int k = threadIdx.x; k < nh; k += ENC_NT) {
            const int2 e = hl[f * pool_h + sh.off + k];
            int c = e.x - 1;                               // bright pixel west of the hole
            for (unsigned hv; (int)((hv = hist[2 * (size_t)c]) >> 24) <= t;) c = (int)(hv & 0xFFFFFFu);
            atomicAdd(&ef[c], min(e.y, 5000));
        }
        __syncthreads();
        for (int k = threadIdx.x; k < nl; k += ENC_NT) {
            int2 &g = bl[f * pool_l + sl.off + k];
            g.y = atomicExch(&ef[g.x], 0);
        }
        __syncthreads();
    }
}

// ---- the 17 binarisations as two growing union-finds ------------------------------------------------------
// dark set {v <= t} grows with t, bright set {v > t} grows as t falls: every pixel joins each forest once, so a
// threshold costs one 1-B/px read of the rectangle plus unions / counts for the pixels that are new at it.
// Roots are always the raster-first pixel of their component (larger root is linked under the smaller).
//   new at this step: lo < v <= hi;  member: DARK ? v <= hi : v > lo
struct SwRect { int x0, y0, x1, y1; };
__device__ __forceinline__ SwRect sw_rect(const FrameState *st, size_t f)
{
    return SwRect{st[f].crect[0], st[f].crect[1], st[f].crect[2], st[f].crect[3]};
}

// ---- pixels of the rectangle sorted by the step at which they join: bucket b is new for the dark set at threshold
// slot b and for the bright set at slot b - 1, so every later kernel of the sweep runs over a dense list
__device__ __forceinline__ int sw_level(int v) { return sweep_level(v); }
static_assert(NTHR == 17, "sweep_level (cpe_dev.h) holds the bucket count");
constexpr int BK_CHUNK = 8192;    // pixels per workgroup of the two bucket passes

// The scatter pass also gives every listed pixel its first node of the bright forest: {the first pixel of its run of one
// bucket inside the wavefront's 64 pixels, never absorbed}.  (Round 2 wrote these entries bucket by bucket, one step ahead
// of their use, through the lists: 2 % of the rectangle per pass, one 64-byte line per entry; here the stores of a
// wavefront fall into a few lines.)
template <bool SCATTER>
__global__ __launch_bounds__(256) void k_bk_pass(const uint8_t *__restrict__ img, int h, int w, const FrameState *__restrict__ st,
                                                 int *__restrict__ sw, int *__restrict__ bk, int *__restrict__ bright /* nodes {parent, history} */)
{
    __shared__ int s_cnt[NBK], s_base[NBK];
    const size_t N = (size_t)h * w, f = blockIdx.y;
    const int t = threadIdx.x;
    const SwRect r = sw_rect(st, f);
    int *S = sw + f * SW_STRIDE;
    // first entry of every bucket = the sizes before it (counted by k_clahe_apply): every workgroup sums them for itself, the
    // first one of a frame also leaves the table for the sweep kernels (k_bk_scan was a launch of its own for this)
    __shared__ int s_off[NBK];
    if (SCATTER && t < NBK) {
        int off = 0;
        for (int b = 1; b < t; b++) off += S[SW_BS + b];
        s_off[t] = off;
        if (blockIdx.x == 0 && t > 0) S[SW_BO + t] = off;
    }
    {   // the workgroup's pixels lie in rows ya .. yb: nothing to do outside the working rectangle (three quarters of a frame)
        const size_t p0 = (size_t)blockIdx.x * BK_CHUNK;
        const int ya = (int)(p0 / w), yb = (int)(min(p0 + BK_CHUNK, N) - 1) / w;
        if (r.x1 < r.x0 || yb < r.y0 || ya > r.y1) return;
    }
    if (t < NBK) s_cnt[t] = 0;
    __syncthreads();
    const uint8_t *im = img + f * N;
    int lev[BK_CHUNK / 256];
#pragma unroll
    for (int k = 0; k < BK_CHUNK / 256; k++) {
        const size_t i = (size_t)blockIdx.x * BK_CHUNK + k * 256 + t;
        int l = 0, x = 0;
        if (i < N) {
            const int y = (int)(i / w);
            x = (int)(i - (size_t)y * w);
            if (!(y < r.y0 || y > r.y1 || x < r.x0 || x > r.x1)) l = sw_level(im[i]);
        }
        if (SCATTER) {
            // first node of the bright forest: the first pixel of the pixel's run of one bucket among this wavefront's 64
            // consecutive pixels (depth 1; x > r.x0: the lane before holds the left neighbour in the same row)
            const int lane = t & 63, ll = __shfl_up(l, 1, 64);
            const bool same = l > 0 && lane > 0 && ll == l && x > r.x0;
            const unsigned long long st1 = __ballot(l > 0 && !same);
            if (l > 0) {
                const int first = (int)i - (lane - (63 - __clzll((long long)(st1 & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull))))));
                *reinterpret_cast<int2 *>(bright + (f * N + i) * 2) = make_int2(first, -1);
            }
        }
        lev[k] = l;
        if (l && !SCATTER) atomicAdd(&s_cnt[l], 1);
        if (l && SCATTER) lev[k] = l | (atomicAdd(&s_cnt[l], 1) << 8);   // rank inside this workgroup's share
    }
    __syncthreads();
    if (!SCATTER) {
        if (t > 0 && t < NBK && s_cnt[t]) atomicAdd(&S[SW_BS + t], s_cnt[t]);
        return;
    }
    if (t > 0 && t < NBK) s_base[t] = s_cnt[t] ? s_off[t] + atomicAdd(&S[SW_BC + t], s_cnt[t]) : 0;   // the cursors start at 0 (memset)
    __syncthreads();
#pragma unroll
    for (int k = 0; k < BK_CHUNK / 256; k++) {
        const int l = lev[k] & 255;
        if (l) bk[f * N + s_base[l] + (lev[k] >> 8)] = (int)((size_t)blockIdx.x * BK_CHUNK + k * 256 + t);
    }
}

constexpr int SW_GRID = 96;       // workgroups per frame of the kernels that walk one bucket (fewer in large batches: 24 at 256 images)

// list entry of this lane = pixel p (negative: none): is the previous lane's entry its left neighbour in the same row?
// (wavefront collective; k_sw_new's init pass and k_sw_unite walk the bucket lists with the same lane <-> entry mapping)
__device__ __forceinline__ bool sw_prelinked(int p, int w, int lane)
{
    const int prev = __shfl_up(p, 1, 64);
    return p >= 0 && lane > 0 && prev == p - 1 && (p % w) != 0;
}

// The sweep kernels are launched on a one-dimensional grid of n * (blocks per frame) workgroups, frame after frame.
// (Measured and dropped in round 3: giving every frame to the workgroups of ONE XCD -- ids = f mod 8, so that its parent
//  plane and bucket lists sit in one 4 MiB L2 -- made k_sw_unite 10 % slower, 2698 against 2725 frames/s for the path: the
//  unions and the agent-scope loads of the forest go past the L2 anyway, and eight frames at a time load the XCDs unevenly.)
struct SwBlock { int f, bx, gx; };
__device__ __forceinline__ SwBlock sw_block(int n, int per_frame)
{
    SwBlock r;
    r.f = blockIdx.x / per_frame; r.bx = blockIdx.x - r.f * per_frame; r.gx = per_frame;
    if (r.f >= n) r.f = -1;
    return r;
}
inline unsigned sw_grid(int n, int per_frame) { return (unsigned)(n * per_frame); }

// Membership of the neighbours comes from the one-bit planes of the thresholds (planes[t] = img > 50 + 10 t, built for the
// border tracers anyway): member plane pm, and plane po for "was a member before this step" (po < 0: nobody was).  A list
// holds ~2 % of the rectangle's pixels, so the eight neighbour bytes of an entry were three 64-byte lines of the image of
// its own; the same rows of a bit plane are shared by the ~10 entries of 512 pixels.
template <bool DARK>
__device__ __forceinline__ void sw_unite_body(const SwBlock vb, const uint32_t *__restrict__ bits, int pm, int po, int h, int w, int bucket,
                                              const FrameState *__restrict__ st, const int *__restrict__ sw,
                                              const int *__restrict__ bk, int *__restrict__ P)
{
    const size_t N = (size_t)h * w, f = vb.f;
    const int *S = sw + f * SW_STRIDE;
    const int nb = S[SW_BS + bucket];
    const int *list = bk + f * N + S[SW_BO + bucket];
    const SwRect r = sw_rect(st, f);
    const int ws = bit_row_words(w);
    const size_t plane_words = (size_t)h * ws;
    const uint32_t *bm = bits + (f * NTHR + pm) * plane_words;
    const uint32_t *bo = bits + (f * NTHR + (po < 0 ? pm : po)) * plane_words;
    constexpr int FS = DARK ? 1 : 2;              // ints per forest node (the bright forest: {parent, merge-history word})
    int *Pf = P + f * N * FS;
    // A pair of adjacent members is united by the newer pixel (the later one in raster order when both are new).
    // Horizontal pairs always; a vertical pair only if the pair one column to the left is not also a member pair
    // (that pair is connected by induction and joins through the two horizontal links); a diagonal pair only if
    // neither of the two pixels completing the 2x2 square is a member.
    // The lanes of a pre-linked run (consecutive lanes, consecutive pixels, all pointing at the run's first pixel since
    // k_sw_new's init pass) share their unions: each lane looks up the roots of its own partners, the run takes the
    // smallest of them, and only the run's first lane links the run to it.  Lane by lane the same unions were one
    // memory-side atomic per pixel -- the lanes of a wavefront all read "not linked yet" before any of them links.
    const int lane = threadIdx.x & 63;
    for (int e0 = vb.bx * 256; e0 < nb; e0 += vb.gx * 256) {   // wave-uniform: the shuffles below need every lane
        const int e = e0 + threadIdx.x;
        const int i = e < nb ? list[e] : -2;
        // already a child of its run's first pixel: the previous entry is its left neighbour AND the passes that wrote the first
        // nodes saw the two in one 64-pixel chunk (k_ccl_init: chunks from the rectangle's left edge; k_bk_pass: 64 consecutive
        // frame indices) -- runs are not linked across chunk borders, there the pixel unites with its left neighbour below
        const bool prel = sw_prelinked(i, w, lane) && (DARK ? (((i % w) - r.x0) & 63) != 0 : (i & 63) != 0);
        int pr[DARK ? 4 : 8], np = 0;
        if (i >= 0) {
            const int y = i / w, x = i - y * w;
            const bool Lb = x > r.x0, Rb = x < r.x1, Ub = y > r.y0, Db = y < r.y1;
            // three-column windows (bit 0: x - 1, bit 1: x, bit 2: x + 1; pixel x is bit x + 32 of its plane row) of the rows
            // y - 1 .. y + 1; columns / rows outside the rectangle are never members
            const int sh = (x + 31) & 31;
            const size_t wo = (size_t)y * ws + ((x + 31) >> 5);
            const unsigned cm = (Lb ? 1u : 0u) | 2u | (Rb ? 4u : 0u);
            auto win = [&](const uint32_t *pl, long long drow) -> unsigned {
                const unsigned long long v = *reinterpret_cast<const u64_a4 *>(pl + wo + drow * ws);
                const unsigned b = (unsigned)(v >> sh);
                return (DARK ? ~b : b) & cm;
            };
            const unsigned mu = Ub ? win(bm, -1) : 0u, mc = win(bm, 0), md = Db ? win(bm, 1) : 0u;
            const unsigned oc = po < 0 ? 0u : win(bo, 0), od = (po < 0 || !Db) ? 0u : win(bo, 1);
            const bool mL = mc & 1u, mU = mu & 2u, mD = md & 2u;
            if (mL && !prel) pr[np++] = i - 1;
            if (oc & 4u) pr[np++] = i + 1;
            if (mU && !(mL && (mu & 1u))) pr[np++] = i - w;
            if ((od & 2u) && !(mL && (md & 1u))) pr[np++] = i + w;
            if (!DARK) {
                const bool mR = mc & 4u;
                if (!mU) {
                    if ((mu & 4u) && !mR) pr[np++] = i - w + 1;
                    if ((mu & 1u) && !mL) pr[np++] = i - w - 1;
                }
                if (!mD) {
                    if ((od & 4u) && !mR) pr[np++] = i + w + 1;
                    if ((od & 1u) && !mL) pr[np++] = i + w - 1;
                }
            }
        }
        int mine = INT_MAX;
#pragma unroll
        for (int k = 0; k < (DARK ? 4 : 8); k++)
            if (k < np) { pr[k] = uf_find<FS>(Pf, pr[k]); mine = min(mine, pr[k]); }   // read-only walk: k_sw_new flattens every step, the paths are short, and pointer-jumping stores are fabric writes
        // smallest partner root of the run: segmented min over the lanes that share a first lane
        const unsigned long long starts = __ballot(i >= 0 && !prel);
        const int hl = i >= 0 ? 63 - __clzll((long long)(starts & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull)))) : -1 - lane;
        int m = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int om = __shfl_down(m, d, 64), oh = __shfl_down(hl, d, 64);
            if (lane + d < 64 && oh == hl) m = min(m, om);
        }
        const int runmin = __shfl(m, hl < 0 ? lane : hl, 64);
        if (i >= 0 && runmin != INT_MAX) {
            if (lane == hl) uf_unite<FS>(Pf, i, runmin);
#pragma unroll
            for (int k = 0; k < (DARK ? 4 : 8); k++)
                if (k < np && pr[k] != runmin) uf_unite<FS>(Pf, pr[k], runmin);   // the run joins several components: rare
        }
    }
}

// rectangle border pixels of the dark set: touch[root] = epoch (such a component is not a hole, now or later)
__global__ __launch_bounds__(256) void k_sw_touch(const uint8_t *__restrict__ img, int n, int h, int w, int hi,
                                                  const FrameState *__restrict__ st, const int *__restrict__ P,
                                                  uint8_t *__restrict__ touch, int epoch)
{
    // grid (a few workgroups, n): the border pixels of a frame's rectangle in turns (one thread per pixel of the largest
    // possible border was n * 24 workgroups per launch, nearly all of them empty)
    const int f = blockIdx.y;
    const SwRect r = sw_rect(st, f);
    if (r.x1 < r.x0) return;
    const int rw = r.x1 - r.x0 + 1, rh = r.y1 - r.y0 + 1;
    const size_t N = (size_t)h * w;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < 2 * rw + 2 * rh; k += gridDim.x * blockDim.x) {
        int x, y;
        if (k < rw) { x = r.x0 + k; y = r.y0; }
        else if (k < 2 * rw) { x = r.x0 + k - rw; y = r.y1; }
        else if (k < 2 * rw + rh) { x = r.x0; y = r.y0 + k - 2 * rw; }
        else { x = r.x1; y = r.y0 + k - 2 * rw - rh; }
        const int p = y * w + x;
        if ((int)img[f * N + p] <= hi) touch[f * N + uf_find(P + f * N, p)] = (uint8_t)epoch;
    }
}

__device__ __forceinline__ void sw_append(bool want, int value, int *counter, int2 *list, FrameState *S, int cap)
{
    const int lane = threadIdx.x & 63;
    unsigned long long b = __ballot(want);
    if (!b) return;
    const int leader = __ffsll((long long)b) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, __popcll(b));
    base = __shfl(base, leader, 64);
    if (want) {
        int k = base + __popcll(b & ((1ull << lane) - 1ull));
        if (k < cap) list[k].x = value;
        else set_overflow(*S, OVF_SWEEP);
    }
}

// pixels that joined at this step.  DARK: flatten, add them to their component's pixel count, list the ones that are
// roots of components away from the rectangle border.  BRIGHT: flatten, list the ones that are roots, zero their
// enclosed total.
template <bool DARK>
__device__ __forceinline__ void sw_new_body(const SwBlock vb, int h, int w, int bucket, int init_bucket, FrameState *__restrict__ st,
                                            const int *__restrict__ bk, int *__restrict__ P, int *__restrict__ acc,
                                            const uint8_t *__restrict__ touch, int epoch, int2 *__restrict__ lists,
                                            int *__restrict__ sw, int cnt_base, int slot)
{
    // The bright forest's node is {parent, history word}: history = (root that absorbed the entry) | (step << 24), 0xFFFFFFFF
    // while the entry has never been absorbed -- what k_enclosed_all follows to read the forest as it was at an earlier
    // step.  One 8-byte store per new pixel where round 2 wrote three planes (parent, absorbing root, step), and the
    // "never" value comes with the entry's first store instead of a memset of a plane.
    const size_t N = (size_t)h * w, f = vb.f;
    const int lane = threadIdx.x & 63;
    const int *S = sw + f * SW_STRIDE;
    const int nb = S[SW_BS + bucket];
    const int *list = bk + f * N + S[SW_BO + bucket];
    constexpr int FS = DARK ? 1 : 2;
    int *Pf = P + f * N * FS;
    // (the entries' first nodes -- singletons, runs of one bucket pre-linked along the row -- come from the passes that stream
    //  the image: k_ccl_init for the dark forest, k_bk_pass for the bright one)
    (void)init_bucket;
    if (bucket < 1) return;
    const int pool = sweep_pool(h, w, DARK ? SWL_DARK : SWL_BRIGHT);
    const int loff = sw_slot(S, cnt_base, slot, !DARK, pool).off;   // the counters of the other thresholds are at rest
    for (int e0 = vb.bx * 256; e0 < nb; e0 += vb.gx * 256) {
        const int e = e0 + threadIdx.x;
        const bool isnew = e < nb;
        int i = -1, root = -1;
        if (isnew) {
            i = list[e];
            root = uf_find<FS>(Pf, i);
            if (root != i) {
                if (DARK) Pf[i] = root;
                else *reinterpret_cast<int2 *>(Pf + 2 * (size_t)i) = make_int2(root, root | (epoch << 24));   // joined `root` at this step
            }
        }
        bool is_root = isnew && root == i;
        if (DARK) {
            is_root = is_root && touch[f * N + i] != (uint8_t)epoch;
            int key = root;
            unsigned long long active = __ballot(key >= 0);
            while (active) {
                int leader = __ffsll((long long)active) - 1;
                int lk = __shfl(key, leader, 64);
                unsigned long long same = __ballot(key == lk) & active;
                if (lane == leader) atomicAdd(&acc[f * N + lk], __popcll(same));
                active &= ~same;
            }
        } else if (is_root) acc[f * N + i] = 0;
        sw_append(is_root, i, &sw[f * SW_STRIDE + cnt_base + slot], lists + f * (size_t)pool + loff, &st[f], pool - loff);
    }
}

// components of the previous step.  Still a root: keep (DARK: unless it now reaches the rectangle border).
// Merged into another (DARK): hand its pixel count to the component that absorbed it.
template <bool DARK>
__device__ __forceinline__ void sw_old_body(const SwBlock vb, const int *__restrict__ src, size_t src_frame_stride, int src_cap,
                                            const int *__restrict__ src_cnt, int src_cnt_stride, int src_slot,
                                            int h, int w, FrameState *__restrict__ st, int *__restrict__ P,
                                            int *__restrict__ acc, const uint8_t *__restrict__ touch, int epoch,
                                            int2 *__restrict__ lists, int *__restrict__ sw, int cnt_base, int slot)
{
    // source: the list of threshold `src_slot` (the previous step) or, for the first dark step (src_slot < 0), the root list
    // of the run-based labelling (src: ints, src_frame_stride apart, src_cnt[f * src_cnt_stride] of them)
    const size_t N = (size_t)h * w, f = vb.f;
    const int *S = sw + f * SW_STRIDE;
    const int pool = sweep_pool(h, w, DARK ? SWL_DARK : SWL_BRIGHT);
    const int loff = sw_slot(S, cnt_base, slot, !DARK, pool).off;
    int ns, es = 1;
    const int *sp;
    if (src_slot >= 0) {
        const SwSlot ss = sw_slot(S, cnt_base, src_slot, !DARK, pool);
        ns = ss.cnt; es = 2; sp = reinterpret_cast<const int *>(lists + f * (size_t)pool + ss.off);
    } else { ns = min(src_cnt[f * src_cnt_stride], src_cap); sp = src + f * src_frame_stride; }
    for (int k0 = vb.bx * 256; k0 < ns; k0 += vb.gx * 256) {   // wave-uniform: sw_append is a wavefront collective
    const int k = k0 + threadIdx.x;
    bool keep = false;
    int r = 0;
    if (k < ns) {
        r = sp[(size_t)k * es];
        constexpr int FS = DARK ? 1 : 2;
        int *Pf = P + f * N * FS;
        if (DARK) {
            if (touch[f * N + r] != (uint8_t)epoch) {
                if (uf_load(Pf, r) == r) keep = true;
                else atomicAdd(&acc[f * N + uf_find_c(Pf, r)], acc[f * N + r]);
            }
        } else {
            keep = uf_load<FS>(Pf, r) == r;
            if (keep) acc[f * N + r] = 0;
            else Pf[2 * (size_t)r + 1] = uf_find_c<FS>(Pf, r) | (epoch << 24);   // absorbed at this step
        }
    }
    sw_append(keep, r, &sw[f * SW_STRIDE + cnt_base + slot], lists + f * (size_t)pool + loff, &st[f], pool - loff);
    }
}

// freeze the per-component totals of threshold slot `slot` next to the roots: the accumulator plane moves on
// trace (optional): the entries whose border has to be followed go to a dense list of their own, so the lanes of a
// wavefront of k_blob_trace walk borders of similar length.  A hole of n <= 3 pixels spans at most 1x3 or 2x2 pixels,
// its border polygon runs through pixels 8-adjacent to it, so its area is at most 2x4 or 3x3 < 10 = minArea;
// a hole of n >= 5000 pixels has a border polygon of area >= n >= maxArea.
__device__ __forceinline__ void sw_snap_body(const SwBlock vb, int2 *__restrict__ lists, int *__restrict__ sw, int cnt_base, int slot,
                                             int h, int w, const int *__restrict__ acc, int2 *__restrict__ trace, FrameState *__restrict__ st)
{
    const size_t N = (size_t)h * w, f = vb.f;
    const int *S = sw + f * SW_STRIDE;
    const int pool = sweep_pool(h, w, SWL_DARK), pool_t = sweep_pool(h, w, SWL_TRACE);
    const SwSlot sl = sw_slot(S, cnt_base, slot, false, pool);
    const int cnt = sl.cnt;
    const int toff = trace ? sw_slot(S, SW_NT, slot, false, pool_t).off : 0;
    const int lane = threadIdx.x & 63;
    for (int k0 = vb.bx * 256; k0 < cnt; k0 += vb.gx * 256) {
        const int k = k0 + threadIdx.x;
        const bool valid = k < cnt;
        int2 e = make_int2(0, 0);
        if (valid) {
            int2 &g = lists[f * (size_t)pool + sl.off + k];
            g.y = acc[f * N + g.x];
            e = g;
        }
        if (!trace) continue;
        const bool want = valid && e.y > 3 && e.y < 5000;
        unsigned long long b = __ballot(want);
        if (!b) continue;
        const int leader = __ffsll((long long)b) - 1;
        int base = 0;
        if (lane == leader) base = atomicAdd(&sw[f * SW_STRIDE + SW_NT + slot], __popcll(b));
        base = __shfl(base, leader, 64);
        if (want) {
            const int q = toff + base + __popcll(b & ((1ull << lane) - 1ull));
            if (q < pool_t) trace[f * (size_t)pool_t + q] = e;
            else set_overflow(st[f], OVF_SWEEP);
        }
    }
}

// The launches of a sweep step.  Per threshold the order is: unions of the pixels that join -> (dark) border marks ->
// new roots + surviving old roots -> freeze the totals.  Steps that do not depend on each other share a launch (they are
// independent workgroups of one grid): the totals of step t-1 are frozen beside the unions of step t (nothing changes a
// total between the end of step t-1 and k_sw_new_old of step t), and the new and the old roots of a step are listed
// together (both append to the same list through its atomic counter).  17 thresholds x 2 polarities: 87 launches
// instead of 135, and the short list walks (a few hundred entries per frame) no longer pay a launch of their own.
template <bool DARK>
__global__ __launch_bounds__(256) void k_sw_unite_snap(int n, int g_unite, int g_snap,
                                                       const uint32_t *__restrict__ bits, int pm, int po, int h, int w, int bucket,
                                                       const FrameState *__restrict__ st, const int *__restrict__ bk, int *__restrict__ P,
                                                       int2 *__restrict__ lists, int *__restrict__ sw, int cnt_base, int snap_slot,
                                                       const int *__restrict__ acc, int2 *__restrict__ trace, FrameState *__restrict__ stw)
{
    SwBlock vb = sw_block(n, g_unite + g_snap);
    if (vb.f < 0) return;
    if (vb.bx < g_unite) {
        vb.gx = g_unite;
        sw_unite_body<DARK>(vb, bits, pm, po, h, w, bucket, st, (const int *)sw, bk, P);
    } else {
        vb.bx -= g_unite; vb.gx = g_snap;
        sw_snap_body(vb, lists, sw, cnt_base, snap_slot, h, w, acc, trace, stw);
    }
}

template <bool DARK>
__global__ __launch_bounds__(256) void k_sw_new_old(int n, int g_new, int g_old, int h, int w, int bucket, int init_bucket,
                                                    FrameState *__restrict__ st, const int *__restrict__ bk, int *__restrict__ P,
                                                    int *__restrict__ acc, const uint8_t *__restrict__ touch, int epoch,
                                                    int2 *__restrict__ lists, int *__restrict__ sw, int cnt_base, int slot,
                                                    const int *__restrict__ src, size_t src_frame_stride, int src_cap,
                                                    const int *__restrict__ src_cnt, int src_cnt_stride, int src_slot)
{
    SwBlock vb = sw_block(n, g_new + g_old);
    if (vb.f < 0) return;
    if (vb.bx < g_new) {
        vb.gx = g_new;
        sw_new_body<DARK>(vb, h, w, bucket, init_bucket, st, bk, P, acc, touch, epoch, lists, sw, cnt_base, slot);
    } else {
        vb.bx -= g_new; vb.gx = g_old;
        sw_old_body<DARK>(vb, src, src_frame_stride, src_cap, src_cnt, src_cnt_stride, src_slot, h, w, st, P, acc, touch, epoch,
                          lists, sw, cnt_base, slot);
    }
}

__global__ __launch_bounds__(256) void k_sw_snap(int n, int g_snap, int2 *__restrict__ lists, int *__restrict__ sw, int cnt_base, int slot,
                                                 int h, int w, const int *__restrict__ acc, int2 *__restrict__ trace, FrameState *__restrict__ st)
{
    const SwBlock vb = sw_block(n, g_snap);
    if (vb.f < 0) return;
    sw_snap_body(vb, lists, sw, cnt_base, slot, h, w, acc, trace, st);
}

// groups with >= 2 centres -> key points -> filled discs (cv2.circle, Circle() midpoint spans)
__global__ __launch_bounds__(256) void k_discs(FrameState *__restrict__ st, const Group *__restrict__ groups, int h, int w,
                                               uint8_t *__restrict__ ext)
{
    // one wavefront per group, in turns: lane 0's arithmetic is cv2's, the lanes share the pixels of each span
    const int f = blockIdx.y, lane = threadIdx.x & 63;
    const int ng = min(st[f].n_groups, MAXG);
    uint8_t *im = ext + (size_t)f * h * w;
    for (int gi = blockIdx.x * 4 + (threadIdx.x >> 6); gi < ng; gi += gridDim.x * 4) {
        const Group &g = groups[(size_t)f * MAXG + gi];
        const int gn = g.n;
        if (gn < 2) continue;
        double sx = 0, sy = 0, nrm = 0;
        for (int j = 0; j < gn; j++) { sx += 1.0 * g.c[j][0]; sy += 1.0 * g.c[j][1]; nrm += 1.0; }
        sx *= (1. / nrm);
        sy *= (1. / nrm);
        float kx = (float)sx, ky = (float)sy, ksize = (float)(g.c[gn / 2][2]) * 2.0f;
        float radius = ksize / 2;
        int er = (int)((double)radius + 4);
        int cx = (int)kx, cy = (int)ky;
        if (lane == 0) atomicAdd(&st[f].n_kp, 1);
        int err = 0, dx = er, dy = 0, plus = 1, minus = (er << 1) - 1;
        while (dx >= dy) {
            int ys[4] = {cy - dy, cy + dy, cy - dx, cy + dx};
            int xa[4] = {cx - dx, cx - dx, cx - dy, cx - dy};
            int xb[4] = {cx + dx, cx + dx, cx + dy, cx + dy};
            for (int q = 0; q < 4; q++) {
                if (ys[q] < 0 || ys[q] >= h) continue;
                int x1 = max(xa[q], 0), x2 = min(xb[q], w - 1);
                for (int x = x1 + lane; x <= x2; x += 64) im[(size_t)ys[q] * w + x] = 255;
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
}

// contourArea of every external contour of the disc union; keep the largest (first in OpenCV order on ties)
__global__ __launch_bounds__(64) void k_region_area(const uint32_t *__restrict__ ext_bits, int h, int w,
                                                    const int *__restrict__ roots, FrameState *__restrict__ st,
                                                    unsigned long long *__restrict__ best, int single_is_positive)
{
    __shared__ unsigned long long s_win[BW_ROWS * 64];
    const int f = blockIdx.y;
    const int ncomp = min(st[f].n_roots, MAXROOTS);
    const int ws = bit_row_words(w);
    if (single_is_positive && ncomp == 1) {
        // the usual case of the disc union: one component, so there is nothing to compare and its 3000-step border (one
        // lane, ~1.7 us per step: 5 ms on the critical path of a call) need not be walked.  Its area is positive: a
        // union of filled discs of radius >= 4.
        if (blockIdx.x == 0 && threadIdx.x == 0) best[f] = (1ull << 24) | (unsigned long long)(roots[(size_t)f * MAXROOTS] & 0xFFFFFF);
        return;
    }
    for (int k = blockIdx.x * 64 + threadIdx.x; k < ncomp; k += gridDim.x * 64) {   // components in turns, one per lane
        const int root = roots[(size_t)f * MAXROOTS + k];
        BitWin nz{ext_bits + (size_t)f * h * ws, ws, h, s_win + threadIdx.x};
        StatVisitor sv;
        if (!trace_border(nz, root % w, root / w, false, sv, 8 * (w + h) + (1 << 20))) { set_overflow(st[f], OVF_TRACE); continue; }
        sv.finish();
        long long a2 = sv.a00 < 0 ? -sv.a00 : sv.a00;  // 2 * area, exact
        if (a2 <= 0) continue;
        // maximise (area, root): the later-discovered contour comes first in OpenCV's list
        unsigned long long key = ((unsigned long long)a2 << 24) | (unsigned long long)(root & 0xFFFFFF);
        atomicMax(&best[f], key);
    }
}

struct HullVisitor {
    int *lo, *hi;
    int minx = INT_MAX, maxx = INT_MIN, miny = INT_MAX, maxy = INT_MIN;
    __device__ __forceinline__ void point(int x, int y, bool)
    {
        lo[x] = min(lo[x], y);
        hi[x] = max(hi[x], y);
        minx = min(minx, x); maxx = max(maxx, x); miny = min(miny, y); maxy = max(maxy, y);
    }
    __device__ __forceinline__ bool stop() const { return false; }
};

__device__ __forceinline__ long long cross3(int ox, int oy, int ax, int ay, int bx, int by)
{
    return (long long)(ax - ox) * (by - oy) - (long long)(ay - oy) * (bx - ox);
}

__device__ void dev_line(uint8_t *img, int h, int w, int x1, int y1, int x2, int y2)
{
    int dx = x2 - x1, dy = y2 - y1;
    if (dx < 0) { dx = -dx; dy = -dy; x1 = x2; y1 = y2; }
    int sy = dy < 0 ? -1 : 1;
    int ady = dy < 0 ? -dy : dy;
    int x = x1, y = y1;
    if (ady > dx) {
        int err = ady - (dx + dx), plus = ady + ady, minus = -(dx + dx);
        for (int i = 0; i <= ady; i++) {
            if (x >= 0 && x < w && y >= 0 && y < h) img[(size_t)y * w + x] = 255;
            int m = err < 0;
            err += minus + (m ? plus : 0);
            y += sy;
            if (m) x += 1;
        }
    } else {
        int err = dx - (ady + ady), plus = dx + dx, minus = -(ady + ady);
        for (int i = 0; i <= dx; i++) {
            if (x >= 0 && x < w && y >= 0 && y < h) img[(size_t)y * w + x] = 255;
            int m = err < 0;
            err += minus + (m ? plus : 0);
            x += 1;
            if (m) y += sy;
        }
    }
}

// hull of the selected contour -> filled polygon (drawContours thickness=-1) + bounding rect
constexpr int HULL_LDS_W = 2048;   // frames up to this width keep the column extents and the hull in LDS
__global__ __launch_bounds__(256) void k_hull_fill(const uint32_t *__restrict__ ext_bits, int h, int w,
                                                   const unsigned long long *__restrict__ best, FrameState *__restrict__ st,
                                                   int *__restrict__ lohi /* n * 2 * w */, int *__restrict__ hull /* n * 4 * w */,
                                                   uint8_t *__restrict__ mc)
{
    const int f = blockIdx.x, t = threadIdx.x;
    __shared__ int s_nh;
    const size_t N = (size_t)h * w;
    FrameState &S = st[f];
    if (best[f] == 0) {
        if (t == 0) S.status = CPE_ST_NO_REGION;
        return;
    }
    const int root = (int)(best[f] & 0xFFFFFF);
    __shared__ int s_lo[HULL_LDS_W], s_hi[HULL_LDS_W], s_hp[4 * HULL_LDS_W];
    __shared__ unsigned long long s_win[BW_ROWS * 64];
    const bool in_lds = w <= HULL_LDS_W;
    int *lo = in_lds ? s_lo : lohi + (size_t)f * 2 * w, *hi = in_lds ? s_hi : lohi + (size_t)f * 2 * w + w;
    int *hp = in_lds ? s_hp : hull + (size_t)f * 4 * w;
    for (int x = t; x < w; x += 256) { lo[x] = INT_MAX; hi[x] = INT_MIN; }
    __shared__ int s_box[4], s_scan[4];
    __shared__ uint8_t s_keep[2 * HULL_LDS_W + 2];
    if (t == 0) { s_box[0] = INT_MAX; s_box[1] = INT_MIN; s_box[2] = INT_MAX; s_box[3] = INT_MIN; }
    __syncthreads();
    const int ws = bit_row_words(w);
    // One component in the mask (the usual case): the column extents of the contour are the column extents of the mask (the
    // top / bottom pixel of a column has only background above / below it up to the frame border, so it lies on the outer
    // border), found by all threads from the bit plane instead of one lane walking the border.
    const bool single = S.n_roots == 1;
    if (single) {
        const uint32_t *plane = ext_bits + (size_t)f * h * ws;
        const int nwc = (w + 31) >> 5;              // word columns that hold pixels: words 1 .. nwc of a row
        constexpr int SEG = 4;                      // row segments per word column
        for (int item = t; item < nwc * SEG; item += 256) {
            const int j = item % nwc, seg = item / nwc;
            const int y0 = (int)((long long)h * seg / SEG), y1 = (int)((long long)h * (seg + 1) / SEG);
            // rows in batches of HB: the loads of a batch are independent and in flight together
            constexpr int HB = 16;
            uint32_t seen = 0;
            for (int yb = y0; yb < y1; yb += HB) {
                uint32_t v[HB];
#pragma unroll
                for (int k = 0; k < HB; k++) v[k] = yb + k < y1 ? plane[(size_t)(yb + k) * ws + 1 + j] : 0u;
#pragma unroll
                for (int k = 0; k < HB; k++) {
                    uint32_t nw = v[k] & ~seen;
                    seen |= v[k];
                    while (nw) { const int b = __ffs(nw) - 1; nw &= nw - 1; atomicMin(&lo[32 * j + b], yb + k); }
                }
            }
            seen = 0;
            for (int yb = y1 - 1; yb >= y0; yb -= HB) {
                uint32_t v[HB];
#pragma unroll
                for (int k = 0; k < HB; k++) v[k] = yb - k >= y0 ? plane[(size_t)(yb - k) * ws + 1 + j] : 0u;
#pragma unroll
                for (int k = 0; k < HB; k++) {
                    uint32_t nw = v[k] & ~seen;
                    seen |= v[k];
                    while (nw) { const int b = __ffs(nw) - 1; nw &= nw - 1; atomicMax(&hi[32 * j + b], yb - k); }
                }
            }
        }
        __syncthreads();
        int mnx = INT_MAX, mxx = INT_MIN, mny = INT_MAX, mxy = INT_MIN;
        for (int x = t; x < w; x += 256) {
            if (lo[x] == INT_MAX) continue;
            mnx = min(mnx, x); mxx = max(mxx, x); mny = min(mny, lo[x]); mxy = max(mxy, hi[x]);
        }
        if (mxx >= 0) { atomicMin(&s_box[0], mnx); atomicMax(&s_box[1], mxx); atomicMin(&s_box[2], mny); atomicMax(&s_box[3], mxy); }
        __syncthreads();
    }
    if (t == 0) {
        HullVisitor hv{lo, hi};
        if (single) { hv.minx = s_box[0]; hv.maxx = s_box[1]; hv.miny = s_box[2]; hv.maxy = s_box[3]; }
        else {
            BitWin nz{ext_bits + (size_t)f * h * ws, ws, h, s_win};
            trace_border(nz, root % w, root / w, false, hv, 8 * (w + h) + (1 << 20));
            s_box[0] = hv.minx; s_box[1] = hv.maxx; s_box[2] = hv.miny; s_box[3] = hv.maxy;
        }
        S.rect[0] = hv.minx; S.rect[1] = hv.miny; S.rect[2] = hv.maxx - hv.minx + 1; S.rect[3] = hv.maxy - hv.miny + 1;
    }
    __syncthreads();
    if (in_lds) {
        // cv2.convexHull of the contour = Andrew's monotone chain over the column extents, points sorted by (x, y): the chain from
        // the first point to the last keeps (x, lo[x]) vertices, the chain back keeps (x, hi[x]) vertices, a point is dropped
        // when the turn at it is not strictly convex (cross <= 0).  That chain is a serial walk (2 x ~650 columns with
        // dependent LDS reads: most of this kernel's time in round 2).  Same vertex list, every point on its own: a point
        // i of a chain stays iff every pair a < i < b of the chain turns strictly at it, and with u_a = P_i - P_a,
        // v_b = P_b - P_i (all in one half plane: the points are sorted) that is cross(A, B) > 0 for the u of largest and
        // the v of smallest direction angle -- two running maxima over the chain, exact in int32 (coordinates < 2^12).
        int *cx = reinterpret_cast<int *>(s_win);                 // columns that hold pixels, ascending (the tracer is done with its window)
        int *ql = s_hp, *qu = s_hp + HULL_LDS_W + 1;              // the two chains as x | y << 16
        int *hv_out = s_hp + 2 * (HULL_LDS_W + 1);                // the vertex list (x, y pairs): 2 * (HULL_LDS_W - 1) ints
        const int minx = s_box[0], maxx = s_box[1];
        auto block_scan = [&](int v, int &total) {                // exclusive prefix sum over the 256 threads
            int incl = v;
            for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off, 64); if ((t & 63) >= off) incl += o; }
            if ((t & 63) == 63) s_scan[t >> 6] = incl;
            __syncthreads();
            int base = 0;
            for (int k = 0; k < (t >> 6); k++) base += s_scan[k];
            total = s_scan[0] + s_scan[1] + s_scan[2] + s_scan[3];
            __syncthreads();
            return base + incl - v;
        };
        constexpr int CPT = HULL_LDS_W / 256;                     // columns per thread
        int cnt = 0;
        for (int k = 0; k < CPT; k++) { const int x = minx + CPT * t + k; if (x <= maxx && lo[x] != INT_MAX) cnt++; }
        int m;
        int pos = block_scan(cnt, m);
        for (int k = 0; k < CPT; k++) { const int x = minx + CPT * t + k; if (x <= maxx && lo[x] != INT_MAX) cx[pos++] = x; }
        __syncthreads();
        if (m <= 0) {                                               // (cannot happen for a selected component; uniform)
            if (t == 0) { s_nh = 0; S.hull_n = 0; }
            __syncthreads();
            return;
        }
        const int xf = cx[0], xl = cx[m - 1];
        const int mL = m + (hi[xl] != lo[xl] ? 1 : 0), mU = m + (hi[xf] != lo[xf] ? 1 : 0);
        for (int j = t; j < mL; j += 256) { const int x = j < m ? cx[j] : xl; ql[j] = x | ((j < m ? lo[x] : hi[x]) << 16); }
        for (int j = t; j < mU; j += 256) { const int x = j < m ? cx[m - 1 - j] : xf; qu[j] = x | ((j < m ? hi[x] : lo[x]) << 16); }
        __syncthreads();
        for (int idx = t; idx < mL + mU; idx += 256) {
            const bool up = idx >= mL;
            const int j = up ? idx - mL : idx, len = up ? mU : mL;
            const int *q = up ? qu : ql;
            bool keep = true;
            if (j > 0 && j < len - 1) {
                const int pi = q[j], xi = pi & 0xFFFF, yi = pi >> 16;
                int ax = xi - (q[j - 1] & 0xFFFF), ay = yi - (q[j - 1] >> 16);
                for (int a = j - 2; a >= 0; a--) {
                    const int ux = xi - (q[a] & 0xFFFF), uy = yi - (q[a] >> 16);
                    if (ax * uy - ay * ux > 0) { ax = ux; ay = uy; }
                }
                int bx = (q[j + 1] & 0xFFFF) - xi, by = (q[j + 1] >> 16) - yi;
                for (int b = j + 2; b < len; b++) {
                    const int vx = (q[b] & 0xFFFF) - xi, vy = (q[b] >> 16) - yi;
                    if (vx * by - vy * bx > 0) { bx = vx; by = vy; }
                }
                keep = ax * by - ay * bx > 0;
            }
            s_keep[idx] = keep ? 1 : 0;
        }
        __syncthreads();
        // vertex list: the first chain, then the second without its two end points (they are the first chain's last and first)
        constexpr int EPT = (2 * HULL_LDS_W + 2 + 255) / 256;
        int c2 = 0;
        for (int k = 0; k < EPT; k++) {
            const int idx = EPT * t + k;
            if (idx < mL + mU && s_keep[idx] && !(idx >= mL && (idx == mL || idx == mL + mU - 1))) c2++;
        }
        int nv;
        int o = block_scan(c2, nv);
        if (m == 1 && mL == 1) nv = 0;                              // a single pixel: the chain holds one point and no polygon
        const bool fits = nv <= HULL_LDS_W - 1;
        for (int k = 0; k < EPT && fits; k++) {
            const int idx = EPT * t + k;
            if (idx < mL + mU && s_keep[idx] && !(idx >= mL && (idx == mL || idx == mL + mU - 1))) {
                const int pv = idx >= mL ? qu[idx - mL] : ql[idx];
                hv_out[2 * o] = pv & 0xFFFF; hv_out[2 * o + 1] = pv >> 16; o++;
            }
        }
        if (t == 0) {
            if (!fits) { set_overflow(S, OVF_VERTS); nv = 0; }
            s_nh = nv; S.hull_n = nv;
        }
        hp = hv_out;
    } else if (t == 0) {
        // frames wider than HULL_LDS_W: the serial monotone chain over columns; points sorted by (x,y): per column first lo then hi
        const int bminx = s_box[0], bmaxx = s_box[1];
        int k = 0;
        for (int x = bminx; x <= bmaxx; x++) {
            if (lo[x] == INT_MAX) continue;
            int ys[2] = {lo[x], hi[x]};
            int cnt = (hi[x] != lo[x]) ? 2 : 1;
            for (int q = 0; q < cnt; q++) {
                while (k >= 2 && cross3(hp[2 * (k - 2)], hp[2 * (k - 2) + 1], hp[2 * (k - 1)], hp[2 * (k - 1) + 1], x, ys[q]) <= 0) k--;
                hp[2 * k] = x; hp[2 * k + 1] = ys[q]; k++;
            }
        }
        int tmin = k + 1;
        bool first = true;
        for (int x = bmaxx; x >= bminx; x--) {
            if (lo[x] == INT_MAX) continue;
            int ys[2] = {hi[x], lo[x]};
            int cnt = (hi[x] != lo[x]) ? 2 : 1;
            for (int q = 0; q < cnt; q++) {
                if (first) { first = false; continue; }  // skip the very last sorted point (already on the chain)
                while (k >= tmin && cross3(hp[2 * (k - 2)], hp[2 * (k - 2) + 1], hp[2 * (k - 1)], hp[2 * (k - 1) + 1], x, ys[q]) <= 0) k--;
                hp[2 * k] = x; hp[2 * k + 1] = ys[q]; k++;
            }
        }
        s_nh = k - 1;
        S.hull_n = k - 1;
    }
    __syncthreads();
    const int nh = s_nh;
    uint8_t *out = mc + f * N;
    if (nh <= 0) return;
    // outline: Line() between consecutive hull vertices
    for (int e = t; e < nh; e += 256) {
        int a = (e + nh - 1) % nh;
        dev_line(out, h, w, hp[2 * a], hp[2 * a + 1], hp[2 * e], hp[2 * e + 1]);
    }
    // the fill is k_hull_rows' (many workgroups per frame): it reads the vertices from HBM
    if (in_lds) {
        int *hg = hull + (size_t)f * 4 * w;
        for (int e = t; e < 2 * nh; e += 256) hg[e] = hp[e];
    }
}

// scan-line fill of the hull (FillEdgeCollection): x in 16.16, left ceil / right floor, rows [ymin, ymax).  One workgroup
// per band of HR_ROWS rows of the hull's bounding rectangle: a thread finds the spans of its row (edge crossings, sorted),
// then the wavefronts write the rows with the lanes across x (16-byte stores where the row allows).  In round 2 the one
// workgroup that computed the hull also filled its ~900 rows, 64 at a time per wavefront, with the rest of the CU idle.
constexpr int HR_ROWS = 64, HR_MAXV = 1024;
__global__ __launch_bounds__(256) void k_hull_rows(int h, int w, const unsigned long long *__restrict__ best, const FrameState *__restrict__ st,
                                                   const int *__restrict__ hull /* n * 4 * w */, uint8_t *__restrict__ mc)
{
    const int f = blockIdx.y, t = threadIdx.x;
    const FrameState &S = st[f];
    if (best[f] == 0) return;
    const int nh = S.hull_n;
    if (nh <= 0) return;
    const int ymin = S.rect[1], ymax = min(S.rect[1] + S.rect[3] - 1, h);
    const int yc = ymin + (int)blockIdx.x * HR_ROWS;
    if (yc >= ymax) return;
    __shared__ int s_hp[2 * HR_MAXV];
    __shared__ int s_span[HR_ROWS][8];
    __shared__ int s_nsp[HR_ROWS];
    const int *hg = hull + (size_t)f * 4 * w;
    const bool staged = nh <= HR_MAXV;
    if (staged) for (int e = t; e < 2 * nh; e += 256) s_hp[e] = hg[e];
    __syncthreads();
    const int *hp = staged ? s_hp : hg;
    uint8_t *out = mc + (size_t)f * h * w;
    const bool al16 = (w & 15) == 0 && (((size_t)out) & 15) == 0;
    if (t < HR_ROWS) {
        const int y = yc + t;
        int nsp = 0;
        if (y < ymax) {
            long long xs[8];
            int na = 0;
            for (int e = 0; e < nh && na < 8; e++) {
                int a = (e + nh - 1) % nh;
                long long p0x = (long long)hp[2 * a] << 16, p1x = (long long)hp[2 * e] << 16;
                int p0y = hp[2 * a + 1], p1y = hp[2 * e + 1];
                if (p0y == p1y) continue;
                int ey0, ey1;
                long long ex;
                if (p0y < p1y) { ey0 = p0y; ey1 = p1y; ex = p0x; }
                else { ey0 = p1y; ey1 = p0y; ex = p1x; }
                if (!(ey0 <= y && y < ey1)) continue;
                long long dx = (p1x - p0x) / (p1y - p0y);
                xs[na++] = ex + (long long)(y - ey0) * dx;
            }
            for (int a = 1; a < na; a++) {
                long long kx = xs[a];
                int b = a - 1;
                while (b >= 0 && xs[b] > kx) { xs[b + 1] = xs[b]; b--; }
                xs[b + 1] = kx;
            }
            if (y >= 0) {
                for (int a = 0; a + 1 < na; a += 2) {
                    int x1 = (int)((xs[a] + 65535) >> 16), x2 = (int)(xs[a + 1] >> 16);
                    if (x1 < w && x2 >= 0) {
                        s_span[t][2 * nsp] = max(x1, 0);
                        s_span[t][2 * nsp + 1] = min(x2, w - 1);
                        nsp++;
                    }
                }
            }
        }
        s_nsp[t] = nsp;
    }
    __syncthreads();
    const int lane = t & 63, wave = t >> 6;
    for (int rr = wave; rr < HR_ROWS && yc + rr < ymax; rr += 4) {
        uint8_t *row = out + (size_t)(yc + rr) * w;
        for (int q = 0; q < s_nsp[rr]; q++) {
            const int x1 = s_span[rr][2 * q], x2 = s_span[rr][2 * q + 1];
            for (int xb = (x1 & ~15) + 16 * lane; xb <= x2; xb += 16 * 64) {
                if (al16 && xb >= x1 && xb + 15 <= x2) *reinterpret_cast<uint4 *>(row + xb) = make_uint4(~0u, ~0u, ~0u, ~0u);
                else for (int x = max(xb, x1); x <= min(xb + 15, x2); x++) row[x] = 255;
            }
        }
    }
}

// ---- row f-2 (planar target): get_convex_hull, util_plane.py:2590-2689 -----------------------------------------------
// cv2.dilate with cv2.getStructuringElement(MORPH_ELLIPSE, (ks, ks)) on a 0/255 mask: 64x32 tile + apron in LDS; the element is
// given as one half-width per row (dx[i] = -1: empty row), exactly the spans OpenCV builds.
struct EllipseSE { int ks; int dx[32]; };
__global__ __launch_bounds__(256) void k_dilate_ellipse(const uint8_t *__restrict__ src, int h, int w, int tiles_x, int tiles_y,
                                                        EllipseSE se, uint8_t *__restrict__ dst)
{
    constexpr int TX = 64, TY = 32, RMAX = 15;
    __shared__ uint8_t s_in[(TY + 2 * RMAX) * (TX + 2 * RMAX)];
    const int r = se.ks / 2, IW = TX + 2 * r, IH = TY + 2 * r;
    const int tiles = tiles_x * tiles_y;
    const int f = blockIdx.x / tiles, tt = blockIdx.x - f * tiles;
    const int gx0 = (tt % tiles_x) * TX, gy0 = (tt / tiles_x) * TY;
    const size_t N = (size_t)h * w;
    bool any_set = false;
    for (int i = threadIdx.x; i < IH * IW; i += 256) {
        int ry = i / IW, rx = i - ry * IW;
        int y = gy0 - r + ry, x = gx0 - r + rx;
        s_in[i] = (x >= 0 && x < w && y >= 0 && y < h) ? src[f * N + (size_t)y * w + x] : 0;   // outside: never a source
        any_set |= s_in[i] != 0;
    }
    const bool tile_has_source = __syncthreads_or(any_set ? 1 : 0) != 0;   // most tiles lie outside the hull: all zero
    for (int i = threadIdx.x; i < TY * TX; i += 256) {
        int ry = i / TX, rx = i - ry * TX;
        int y = gy0 + ry, x = gx0 + rx;
        if (y >= h || x >= w) continue;
        bool on = false;
        for (int k = 0; tile_has_source && k < se.ks && !on; k++) {
            const int dxk = se.dx[k];
            if (dxk < 0) continue;
            const uint8_t *row = &s_in[(ry + k) * IW + rx + r];   // source row y + (k - r) (the element is symmetric)
            for (int j = -dxk; j <= dxk; j++)
                if (row[j]) { on = true; break; }
        }
        dst[f * N + (size_t)y * w + x] = on ? 255 : 0;
    }
}

__global__ void k_best_reset(int n, unsigned long long *best)
{
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < n) best[f] = 0;
}


}  // namespace


// lplane: L channel of BGR2LAB of a colour frame (util_cylinder.py:1840 on a 3-channel image), or null: grey frames, L = LUT[grey]
int region_stage(const uint8_t *gray, int n, int h, int w, double clip, const RegionBuffers &B, FrameState *st, hipStream_t s,
                 const RegionSide *side, const uint8_t *lplane)
{
    const int lab_lut = lplane ? 0 : 1;
    if (lplane) gray = lplane;
    const size_t N = (size_t)h * w, total = N * n;
    ClaheGeom g;
    g.tilesX = 4; g.tilesY = 4;
    g.ew = (w % 4 == 0 && h % 4 == 0) ? w : w + (4 - (w % 4));
    g.eh = (w % 4 == 0 && h % 4 == 0) ? h : h + (4 - (h % 4));
    g.tw = g.ew / 4; g.th = g.eh / 4;
    int tileTotal = g.tw * g.th;
    g.clipLimit = 0;
    if (clip > 0.0) { g.clipLimit = (int)(clip * tileTotal / 256); if (g.clipLimit < 1) g.clipLimit = 1; }
    g.lutScale = (float)255 / tileTotal;
    CPE_LAUNCH_BEGIN();
    // (st[].n_groups / n_kp, B.best and the bounding-box accumulators B.nrect were reset by k_state_init)
    (void)hipMemsetAsync(B.hist, 0, (size_t)n * 16 * 256 * sizeof(unsigned int), s);
    const int strips = 8;
    CPE_KLAUNCH(k_clahe_hist, dim3(n * 16 * strips), dim3(256), 0, s, gray, n, h, w, g, strips, B.hist, lab_lut);
    CPE_KLAUNCH(k_clahe_lut, dim3(n * 16), dim3(256), 0, s, B.hist, g, B.lut);
    (void)hipMemsetAsync(B.sw, 0, (size_t)n * SW_STRIDE * sizeof(int), s);   // the sweep's counters: k_clahe_apply already counts the buckets
    CPE_KLAUNCH(k_clahe_apply, dim3((unsigned)((N + CLAHE_BLK_PX - 1) / CLAHE_BLK_PX), n), dim3(256), 0, s, gray, h, w, g, (const uint8_t *)B.lut, B.cl, B.nrect, lab_lut,
                B.sw + SW_BS, (int)SW_STRIDE);
    CPE_CHECK_LAUNCH("clahe");
    int rc;
    if (side) {   // the 17 one-bit planes only need the CLAHE image
        (void)hipEventRecord(side->clahe_done, s);
        (void)hipStreamWaitEvent(side->s, side->clahe_done, 0);
        if ((rc = build_bitplanes(B.cl, n, h, w, 50, 10, NTHR, B.bits, side->s)) != CPE_OK) return rc;
        (void)hipEventRecord(side->traced, side->s);   // the bright sweep on `s` reads them too (k_sw_unite_snap)
    } else if ((rc = build_bitplanes(B.cl, n, h, w, 50, 10, NTHR, B.bits, s)) != CPE_OK) return rc;
    // working rectangle for all 34 labelling passes = bounding box of the pixels brighter than the lowest threshold:
    // every brighter set and every hole of every binarisation lies inside it
    if ((rc = ccl_ctl(st, B.nrect, n, h, w, 2, s)) != CPE_OK) return rc;   // crect = the box k_clahe_apply accumulated
    (void)hipMemsetAsync(B.touch, 0, total, s);
    const int swcap = std::max(32768, (int)std::min<long long>(1 << 20, (long long)N / 12));   // grid sizing only: entries one threshold may hold
    const dim3 gpx((unsigned)((N + 255) / 256), n), glist(frame_waves(4 * n, 4, swcap / 256), n), gtrace(frame_waves(n * NTHR, 8, swcap / 64), n, NTHR), gtrace_h(frame_waves(n * NTHR, 4, swcap / 64), n, NTHR), gbk(std::min(SW_GRID, std::max(16, 6144 / n)), n);
    {
        const dim3 gchunk((unsigned)((N + BK_CHUNK - 1) / BK_CHUNK), n);
        CPE_KLAUNCH(k_bk_pass<true>, gchunk, dim3(256), 0, s, (const uint8_t *)B.cl, h, w, (const FrameState *)st, B.sw, B.bk, B.lab2);
        CPE_CHECK_LAUNCH("grey-level buckets");
    }
    // the dark sweep and the hole borders run on the helper stream (if any) beside the bright sweep: the two forests
    // only meet in k_enclosed_all
    hipStream_t ds = side ? side->s : s;
    if (side) { (void)hipEventRecord(side->dark_done, s); (void)hipStreamWaitEvent(ds, side->dark_done, 0); }
    // ---- ascending thresholds: enclosed dark components (4-conn); B.hl[k] = (first pixel, pixel count)
    const int g_bk = (int)gbk.x, g_list = (int)glist.x, g_roots = (int)frame_waves(n, 8, MAXROOTS / 256);
    const int n_grid = n, nx = n;
    for (int k = 0; k < NTHR; k++) {
        const int thr = 50 + 10 * k, epoch = k + 1;
        if (k == 0) {
            // the bulk of the dark set: run-based labelling, flattened; pixels outside it start as singletons
            if ((rc = ccl_run(B.cl, n, h, w, thr, 1, 0, B.lab, B.roots, false, nullptr, 1, B.cnt, 1, nullptr, st, ds, 3)) != CPE_OK) return rc;   // sparse 3: + the sweep's pre-linked runs
            CPE_KLAUNCH(k_sw_touch, dim3(frame_waves(4 * n, 2, 8), n), dim3(256), 0, ds, (const uint8_t *)B.cl, n, h, w, thr,
                        (const FrameState *)st, (const int *)B.lab, B.touch, epoch);
            // first entries of the pixels that join at the next step (bucket 1) | the roots of the labelling that are holes
            CPE_KLAUNCH(k_sw_new_old<true>, dim3(sw_grid(n_grid, g_bk + g_roots)), dim3(256), 0, ds, nx, g_bk, g_roots, h, w, 0, 1, st,
                        (const int *)B.bk, B.lab, B.cnt, (const uint8_t *)B.touch, epoch, B.hl, B.sw, (int)SW_NH, k, (const int *)B.roots, (size_t)MAXROOTS, (int)MAXROOTS, (const int *)&st[0].n_roots,
                        (int)(sizeof(FrameState) / sizeof(int)), -1);
        } else {
            // unions of the pixels that join at this threshold | totals of the previous threshold frozen
            CPE_KLAUNCH(k_sw_unite_snap<true>, dim3(sw_grid(n_grid, g_bk + g_list)), dim3(256), 0, ds, nx, g_bk, g_list, (const uint32_t *)B.bits, k, k - 1, h, w,
                        k, (const FrameState *)st, (const int *)B.bk, B.lab, B.hl, B.sw, (int)SW_NH, k - 1,
                        (const int *)B.cnt, B.tl, st);
            CPE_KLAUNCH(k_sw_touch, dim3(frame_waves(4 * n, 2, 8), n), dim3(256), 0, ds, (const uint8_t *)B.cl, n, h, w, thr,
                        (const FrameState *)st, (const int *)B.lab, B.touch, epoch);
            CPE_KLAUNCH(k_sw_new_old<true>, dim3(sw_grid(n_grid, g_bk + g_list)), dim3(256), 0, ds, nx, g_bk, g_list, h, w, k, k + 1 < NTHR ? k + 1 : 0, st,
                        (const int *)B.bk, B.lab, B.cnt, (const uint8_t *)B.touch, epoch, B.hl, B.sw, (int)SW_NH, k, (const int *)nullptr, (size_t)0, 0, (const int *)nullptr, 0, k - 1);
        }
        CPE_CHECK_LAUNCH("blob sweep (dark)");
    }
    CPE_KLAUNCH(k_sw_snap, dim3(sw_grid(n_grid, g_list)), dim3(256), 0, ds, nx, g_list, B.hl, B.sw, (int)SW_NH, NTHR - 1, h, w, (const int *)B.cnt, B.tl, st);
    if (side && side->joints_done) {   // blob_ch / blob_d share memory with the label planes of the joints and spot chains
        (void)hipStreamWaitEvent(ds, side->joints_done, 0);
        (void)hipStreamWaitEvent(ds, side->spot_done, 0);
    }
    {
        // hole borders of all thresholds and their radii
        CPE_KLAUNCH(k_blob_trace<1>, gtrace_h, dim3(64), 0, ds, (const uint8_t *)B.cl, h, w, (const int2 *)B.tl, (int)SW_NT, st, B.sw, B.blobs,
                    B.blob_d, B.dists, (const uint32_t *)B.bits, B.pool, B.blob_ch, B.maxch, B.maxdf);
        CPE_KLAUNCH(k_blob_median, dim3(frame_waves(n * NTHR, 16, 128), n, NTHR), dim3(64), 0, ds, 0, B.sw, B.blobs, (const int *)B.blob_d, B.dists,
                    (const uint32_t *)B.pool, (const unsigned short *)B.blob_ch, st, B.maxch, B.maxdf);
        if (side) (void)hipEventRecord(side->medians, ds);
    }
    // ---- descending thresholds: bright components (8-conn); B.bl[k] = (first pixel, pixels of the holes it encloses)
    // (the bright forest's first nodes were written by k_bk_pass)
    if (side) (void)hipStreamWaitEvent(s, side->traced, 0);
    for (int j = 0; j < NTHR; j++) {
        const int k = NTHR - 1 - j;   // members: v > 50 + 10 k (plane k); members before this step: v > 60 + 10 k (plane k + 1; none at j = 0)
        CPE_KLAUNCH(k_sw_unite_snap<false>, dim3(sw_grid(n_grid, g_bk)), dim3(256), 0, s, nx, g_bk, 0, (const uint32_t *)B.bits, k, j == 0 ? -1 : k + 1, h, w, k + 1,
                    (const FrameState *)st, (const int *)B.bk, B.lab2, (int2 *)nullptr, B.sw, 0, 0, (const int *)nullptr, (int2 *)nullptr, st);
        const int go = j > 0 ? g_list : 0;   // survivors of the previous (higher) threshold
        CPE_KLAUNCH(k_sw_new_old<false>, dim3(sw_grid(n_grid, g_bk + go)), dim3(256), 0, s, nx, g_bk, go, h, w, k + 1, k, st, (const int *)B.bk, B.lab2, B.cnt2,
                    (const uint8_t *)nullptr, j, B.bl, B.sw, (int)SW_NL, k,
                    (const int *)nullptr, (size_t)0, 0, (const int *)nullptr, 0, k + 1);
        CPE_CHECK_LAUNCH("blob sweep (bright)");
    }
    if (side) (void)hipStreamWaitEvent(s, side->medians, 0);
    CPE_KLAUNCH(k_enclosed_all, dim3(n), dim3(ENC_NT), 0, s, (const int2 *)B.hl, B.bl, (const int *)B.sw, (const int *)B.lab2,
                h, w, B.cnt2);
    CPE_KLAUNCH(k_blob_trace<0>, gtrace, dim3(64), 0, s, (const uint8_t *)B.cl, h, w, (const int2 *)B.bl, (int)SW_NL, st, B.sw, B.blobs,
                B.blob_d, B.dists, (const uint32_t *)B.bits, B.pool, B.blob_ch, B.maxch, B.maxdf);
    CPE_KLAUNCH(k_blob_median, dim3(frame_waves(n * NTHR, 8, 32), n, NTHR), dim3(64), 0, s, 1, B.sw, B.blobs, (const int *)B.blob_d, B.dists, (const uint32_t *)B.pool,
                (const unsigned short *)B.blob_ch, st, B.maxch, B.maxdf);
    {
        // CPE_MERGE_REPLAY (tests): bit 0: every batch takes the in-order replay path instead of the lane-per-blob one;
        // bit 1: every threshold is ranked by the bucketed method of the noisy frames
        const char *e = getenv("CPE_MERGE_REPLAY");
        CPE_KLAUNCH(k_blob_merge, dim3(n), dim3(MG_NT), 0, s, st, (const int *)B.sw, (const BlobRec *)B.blobs, B.order, B.groups,
                    e ? atoi(e) & 3 : 0, B.gmid, h, w);
    }
    CPE_CHECK_LAUNCH("blob merge");
    (void)hipMemsetAsync(B.ext, 0, total, s);
    (void)hipMemsetAsync(B.mc, 0, total, s);
    CPE_KLAUNCH(k_discs, dim3(frame_waves(n, 4, MAXG / 4), n), dim3(256), 0, s, st, B.groups, h, w, B.ext);
    if ((rc = build_bitplanes(B.ext, n, h, w, 0, 0, 1, B.bits, s)) != CPE_OK) return rc;
    rc = ccl_roots_bits(B.bits, n, h, w, B.lab, B.roots, 0, st, s, 0);   // the labelling reads the one-bit plane (1/8 of the bytes)
    if (rc == CPE_ERR_ARG) rc = ccl_run(B.ext, n, h, w, 0, 0, 1, B.lab, B.roots, false, nullptr, 0, nullptr, 0, nullptr, st, s, 1, 1);
    if (rc != CPE_OK) return rc;
    CPE_KLAUNCH(k_region_area, dim3(frame_waves(n, 8, 64), n), dim3(64), 0, s, (const uint32_t *)B.bits, h, w, B.roots, st, B.best, 1);
    CPE_KLAUNCH(k_hull_fill, dim3(n), dim3(256), 0, s, (const uint32_t *)B.bits, h, w, B.best, st, B.lohi, B.hull, B.mc);
    CPE_KLAUNCH(k_hull_rows, dim3((unsigned)((h + HR_ROWS - 1) / HR_ROWS), n), dim3(256), 0, s, h, w, (const unsigned long long *)B.best, (const FrameState *)st, (const int *)B.hull, B.mc);
    CPE_CHECK_LAUNCH("region hull");
    return CPE_OK;
}

// Region stage of the planar script: mask_contour = filled hull of (filled hull of the largest bright blob, dilated by an
// 11x11 ellipse), st[].rect = its bounding rectangle.  Every piece but the dilation is shared with the cylinder path.
int region_stage_plane(const uint8_t *gray, int n, int h, int w, const RegionBuffers &B, FrameState *st, hipStream_t s)
{
    const size_t N = (size_t)h * w, total = N * n;
    int rc;
    CPE_LAUNCH_BEGIN();
    // (st[].n_groups / n_kp, B.best and the bounding-box accumulators B.nrect were reset by k_state_init)
    (void)hipMemsetAsync(B.ext, 0, total, s);
    (void)hipMemsetAsync(B.mc, 0, total, s);
    EllipseSE se;
    se.ks = 11;
    for (int i = 0; i < 32; i++) se.dx[i] = -1;
    {
        const int r = se.ks / 2, c = se.ks / 2;
        const double inv_r2 = r ? 1. / ((double)r * r) : 0;
        for (int i = 0; i < se.ks; i++) {
            const int dy = i - r;
            int dx = (int)lrint(c * sqrt((r * r - dy * dy) * inv_r2));   // saturate_cast<int>(c * sqrt(...)) of getStructuringElement
            se.dx[i] = std::min(dx, c);
        }
    }
    for (int round = 0; round < 2; round++) {
        const uint8_t *img = round == 0 ? gray : (const uint8_t *)B.touch;
        const int thr = round == 0 ? 127 : 0;
        uint8_t *dst = round == 0 ? B.ext : B.mc;
        if (round == 1) CPE_KLAUNCH(k_best_reset, dim3((n + 63) / 64), dim3(64), 0, s, n, B.best);
        if ((rc = ccl_run(img, n, h, w, thr, 0, 1, B.lab, B.roots, false, nullptr, 0, nullptr, 0, nullptr, st, s, 1, 1)) != CPE_OK) return rc;
        if ((rc = build_bitplanes(img, n, h, w, thr, 0, 1, B.bits, s)) != CPE_OK) return rc;
        CPE_KLAUNCH(k_region_area, dim3(frame_waves(n, 8, 64), n), dim3(64), 0, s, (const uint32_t *)B.bits, h, w, B.roots, st, B.best, 0);
        CPE_KLAUNCH(k_hull_fill, dim3(n), dim3(256), 0, s, (const uint32_t *)B.bits, h, w, B.best, st, B.lohi, B.hull, dst);
        CPE_KLAUNCH(k_hull_rows, dim3((unsigned)((h + HR_ROWS - 1) / HR_ROWS), n), dim3(256), 0, s, h, w, (const unsigned long long *)B.best, (const FrameState *)st, (const int *)B.hull, dst);
        if (round == 0) {
            const int tiles_x = (w + 63) / 64, tiles_y = (h + 31) / 32;
            CPE_KLAUNCH(k_dilate_ellipse, dim3((unsigned)(n * tiles_x * tiles_y)), dim3(256), 0, s, (const uint8_t *)B.ext, h, w, tiles_x,
                        tiles_y, se, B.touch);
        }
        CPE_CHECK_LAUNCH("region_stage_plane");
    }
    return CPE_OK;
}

}  // namespace cpe
