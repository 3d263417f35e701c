// detect_grid for a batch of grey frames: orchestration of the image half of the hot path and its C ABI.
//   reference: python_grid_detection_cylinder.py:68-112 (detect_grid) and
//              util_cylinder.color_and_expand_lines (:2014-2060)
// Stage order (the reference's 1..6, with independent stages hoisted):
//   preprocess -> hmask/vmask/joints mask -> region (blob hull + rect) -> joints in rect, spot ellipse,
//   roi masks, fragment expansion -> labels of the expanded masks -> blur7 -> lines / indexing kernel.
// Everything stays in HBM between the u8 frame read and the point-table write; all scratch lives in the
// caller-supplied workspace (cpe_detect_workspace_bytes), laid out plane-major so that every kernel
// streams [n, h, w] planes with fully coalesced accesses.
#include "cpe_dev.h"
#include <initializer_list>
#include <mutex>
#include <stdlib.h>
#include <algorithm>

namespace cpe {

int ccl_run(const uint8_t *img, int n, int h, int w, int thr, int invert, int conn8, int *L, int *roots, bool holes_only,
            uint8_t *touch, int count_mode, int *cnt, int use_rect, int *nrect, FrameState *st, hipStream_t s, int sparse = 0, int flags = 0, int cnt_sel = 0);
int ccl_ctl(FrameState *st, int *nrect, int n, int h, int w, int op, hipStream_t s);
int region_stage(const uint8_t *gray, int n, int h, int w, double clip, const RegionBuffers &B, FrameState *st, hipStream_t s,
                 const RegionSide *side, const uint8_t *lplane);
int joints_mask_stage(int n, int h, int w, const MaskBuffers &B, FrameState *st, hipStream_t s);
int spot_stage(const uint8_t *gray, int n, int h, int w, const MaskBuffers &B, FrameState *st, hipStream_t s, int planar);
int region_stage_plane(const uint8_t *gray, int n, int h, int w, const RegionBuffers &B, FrameState *st, hipStream_t s);
int masks_stage(const uint8_t *gray, int n, int h, int w, const MaskBuffers &B, FrameState *st, hipStream_t s,
                const RegionSide *side, int planar, hipStream_t sj);
int blur7_u8(const uint8_t *src, int n, int h, int w, const FrameState *st, uint8_t *dst, hipStream_t s);
int blur7_bgr(const uint8_t *bgr, int n, int h, int w, const FrameState *st, uint8_t *dst, hipStream_t s);
size_t lines_ws_bytes();
int lines_stage(const int *lab_h, const int *lab_v, const uint8_t *exp_h, const uint8_t *exp_v, const uint8_t *g7, int n, int h, int w, const int *joints,
                FrameState *st, void *lines_ws, double *o_xy, int *o_id, int *o_n, double *o_center, const uint8_t *gray,
                int subpixel, int sp_window, double sp_step, float *sp_scratch, int sp_cap, hipStream_t s, int planar);

int lines_export(const void *lines_ws, int f, double *eq, int *npts, double *pts, int *n_lines, hipStream_t s);

namespace {

struct Layout {
    size_t off[64];
    size_t bytes_per_frame[64];
    size_t total;
};

enum Plane {
    P_BINARY = 0, P_HMASK, P_VMASK, P_MASK_CONTOUR, P_ROI_H, P_ROI_V, P_EXP_H, P_EXP_V, P_JOINTS, P_STATE, P_CL, P_G19, P_G7,
    P_JOINTS_MASK, P_TMPA, P_TMPB, P_CM, P_EXT, P_BASE_H, P_BASE_V, P_TOUCH, P_TMP16, P_LAB0, P_LAB1, P_ROOTS, P_JTMP,
    P_VERTS, P_BEST, P_SEGS, P_HIST, P_LUT, P_BLOBS, P_BLOB_D, P_ORDER, P_DISTS, P_GROUPS, P_LOHI, P_HULL, P_LINES, P_NRECT, P_LAB2, P_LAB3, P_SW, P_SUBPIX, P_HL, P_BL, P_TL, P_BK, P_BITS, P_POOL, P_BLOB_CH, P_LABP, P_LABS, P_ROOTSP, P_ROOTSS, P_BEST2, P_HPAR, P_HTIME, P_GMID, P_FLJ, P_GRAYIN, P_COUNT
};

static_assert(P_COUNT <= 64, "Layout arrays too small");

// capacities of the blob sweep that grow with the frame (border points per threshold ~ cells x perimeter)
static int region_maxch(int h, int w) { long long v = (long long)h * w / 256; return (int)std::min(65535LL, std::max(8192LL, v)); }
static int region_maxdf(int h, int w) { long long v = (long long)h * w / 16; return (int)std::max(65536LL, v); }   // x 17, pooled

Layout make_layout(int n, int h, int w)
{
    Layout L;
    const size_t N = (size_t)h * w;
    size_t per[P_COUNT];
    for (int i = 0; i < P_COUNT; i++) per[i] = N;  // u8 planes by default
    per[P_JOINTS] = (size_t)MAXJ * 2 * sizeof(int);
    per[P_STATE] = sizeof(FrameState);
    per[P_TMP16] = 16;   // (unused)
    per[P_LAB0] = N * 4;
    per[P_LAB1] = N * 4;
    per[P_ROOTS] = (size_t)MAXROOTS * sizeof(int);
    per[P_JTMP] = (size_t)MAXJ * 3 * sizeof(int);
    per[P_VERTS] = (size_t)MAXV * 2 * sizeof(int);
    per[P_BEST] = sizeof(unsigned long long);
    per[P_SEGS] = (size_t)2 * MAXSEG * sizeof(SegRec);
    per[P_HIST] = 16 * 256 * sizeof(unsigned int);
    per[P_LUT] = 16 * 256;
    per[P_BLOBS] = (size_t)17 * MAXB * sizeof(BlobRec);
    per[P_BLOB_D] = (size_t)17 * MAXB * 2 * sizeof(int);
    per[P_ORDER] = (size_t)2 * MAXB * sizeof(int);   // + scratch of k_blob_merge's bucketed ranking
    per[P_DISTS] = (size_t)17 * region_maxdf(h, w) * sizeof(double);
    per[P_POOL] = (size_t)17 * region_maxch(h, w) * 128;
    per[P_BLOB_CH] = (size_t)17 * MAXB * 16 * sizeof(unsigned short);
    per[P_GROUPS] = (size_t)MAXG * sizeof(Group);
    per[P_LOHI] = (size_t)2 * w * sizeof(int);
    per[P_HULL] = (size_t)4 * w * sizeof(int);
    per[P_LINES] = lines_ws_bytes();
    per[P_NRECT] = 16 * sizeof(int);
    per[P_LAB2] = N * 8;   // the bright forest of the blob sweep: {parent, merge-history word} per pixel
    per[P_LAB3] = N * 4;
    per[P_SW] = 192 * sizeof(int);
    per[P_TL] = (size_t)sweep_pool(h, w, SWL_TRACE) * sizeof(int2);
    per[P_BK] = N * 4;
    per[P_LABP] = N * 4;
    per[P_LABS] = N * 4;
    per[P_ROOTSP] = (size_t)MAXROOTS * sizeof(int);
    per[P_ROOTSS] = (size_t)MAXROOTS * sizeof(int);
    per[P_BEST2] = sizeof(unsigned long long);
    per[P_HPAR] = (size_t)h * bit_row_words(w) * sizeof(uint32_t);   // (slot re-used) one-bit plane of the joints mask, written by k_open20_joints
    per[P_HTIME] = 16;
    per[P_GMID] = (size_t)(MAXG - MAXG_LDS) * 4 * sizeof(double);   // x, y, r, next group in the grid cell
    per[P_FLJ] = (size_t)2 * h * ((w + 63) / 64) * sizeof(unsigned long long);   // joints chain: background / outer-background bit masks
    per[P_BITS] = (size_t)17 * h * bit_row_words(w) * sizeof(uint32_t);
    per[P_HL] = (size_t)sweep_pool(h, w, SWL_DARK) * sizeof(int2);
    per[P_BL] = (size_t)sweep_pool(h, w, SWL_BRIGHT) * sizeof(int2);
    per[P_SUBPIX] = (size_t)2 * MAXL * 2 * (size_t)(std::max(h, w) + 128) * sizeof(float);
    // Buffers whose lifetimes never overlap share memory (the call is three chains -- ridge mask / joints, saturated spot,
    // region -- that run side by side and meet in the masks stage, then the lines stage; only buffers of ONE chain, or of
    // stages separated by the join, may be paired):
    //   border points + distance scratch of the blob tracers (read last by k_blob_median) | blob groups (k_blob_merge .. k_discs)
    //   bright forest + its accumulator (dead after k_enclosed_all / the outer borders)   | the masks stage's eight u8 planes
    //   label scratch of the joints and spot chains (the latter again in the masks stage) | the lines stage's tables
    //   dark forest + its accumulator (dead when the dark sweep ends; the forest plane is used again for the region's own
    //   labelling after k_discs, the two as the expanded masks' label planes)              | blob records (hole borders .. k_blob_merge)
    // Planes 0 .. P_G7 are readable after the call (cpe_detect_workspace_plane): nothing is ever written over them later
    // (mask_contour and blur7 share memory with the bright forest, which is dead before they are written).
    size_t o = 0;
    bool placed[P_COUNT] = {};
    auto put = [&](std::initializer_list<int> a, std::initializer_list<int> b, std::initializer_list<int> c = {}) {
        size_t oa = o, ob = o, oc = o;
        for (int p : a) { L.off[p] = oa; oa += align_up(per[p] * (size_t)n, 256); placed[p] = true; }
        for (int p : b) { L.off[p] = ob; ob += align_up(per[p] * (size_t)n, 256); placed[p] = true; }
        for (int p : c) { L.off[p] = oc; oc += align_up(per[p] * (size_t)n, 256); placed[p] = true; }
        o = std::max(oa, std::max(ob, oc));
    };
    for (int i = 0; i < P_COUNT; i++) L.bytes_per_frame[i] = per[i];
    put({P_DISTS, P_POOL}, {P_GROUPS});
    // (round 3) also on the masks side of the bright forest: the disc-union image, mask_contour and the 7x7 blur -- all first
    // written after k_enclosed_all, the forest's last reader, on the same stream or behind the join.  The label planes of
    // the joints and spot chains are only needed inside those chains' own labelling (roots-only passes): the border-chunk
    // ids / distance offsets of the blob tracers live there between the hole traces and the medians -- the tracers' stream
    // waits for the two chains (RegionSide::joints_done / spot_done) -- and the lines stage's tables after that.
    put({P_LAB2, P_LAB3}, {P_ROI_H, P_ROI_V, P_BASE_H, P_BASE_V, P_EXP_H, P_EXP_V, P_TMPA, P_TMPB, P_EXT, P_MASK_CONTOUR, P_G7});
    put({P_LABP, P_LABS}, {P_LINES, P_SUBPIX}, {P_BLOB_CH, P_BLOB_D});
    put({P_LAB0, P_LAB1}, {P_BLOBS});
    for (int i = 0; i < P_COUNT; i++) {
        if (placed[i]) continue;
        L.off[i] = o;
        o += align_up(per[i] * (size_t)n, 256);
    }
    L.total = o;
    return L;
}

// per-frame state of a call, and the accumulators the three chains start from (one launch in front of the fork instead of a
// reset kernel at the head of every chain)
__global__ void k_state_init(FrameState *st, int n, unsigned long long *best, unsigned long long *best_s, int *nrect)
{
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    FrameState z = {};
    z.srect[0] = INT_MAX; z.srect[1] = INT_MAX; z.srect[2] = -1; z.srect[3] = -1;   // spot window: empty until k_spot_scan finds a tile
    st[f] = z;
    if (best) { best[f] = 0; best_s[f] = 0; }                                      // largest-contour keys (region / spot)
    if (nrect) { nrect[16 * f] = INT_MAX; nrect[16 * f + 1] = INT_MAX; nrect[16 * f + 2] = -1; nrect[16 * f + 3] = -1; }   // CLAHE's bounding box
}

__global__ void k_finish(const FrameState *st, int n, int *status, int *n_pts)
{
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    int s = st[f].status;
    if (st[f].overflow) s = CPE_ST_OVERFLOW;
    status[f] = s;
    if (s != CPE_ST_OK) n_pts[f] = 0;
}

// Helper streams for the independent chains of a call: one set (three streams, their events) per device AND caller stream,
// so two host threads -- or one thread with chunks in flight on several streams (FramePipeline(lanes=2)) -- neither share
// nor serialise their side chains.  Created on first use, kept for the life of the process (at most SIDE_SETS caller
// streams per device get their own set; further ones share the last).  CPE_SERIAL=1 keeps everything on the caller's stream.
struct SideStreams {
    std::mutex mu;      // held by a caller from its fork to its join (cpe_detect_grid_batch_ex)
    bool ok = false;
    hipStream_t key = nullptr;
    bool used = false;
    hipStream_t s1 = nullptr, s2 = nullptr, s3 = nullptr;
    hipEvent_t fork = nullptr, join1 = nullptr, join2 = nullptr, e3a = nullptr, e3b = nullptr, e3c = nullptr, e3d = nullptr;
};
constexpr int SIDE_SETS = 4;
SideStreams &side_streams(hipStream_t caller)
{
    static SideStreams sets[32][SIDE_SETS];
    static std::mutex mu;
    static SideStreams none;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return none;
    const char *e = getenv("CPE_SERIAL");   // looked at on every call: a profiling pass can switch the overlap off
    if (e && e[0] == '1') return none;
    std::lock_guard<std::mutex> lk(mu);
    int slot = SIDE_SETS - 1;
    for (int k = 0; k < SIDE_SETS; k++) {
        if (sets[dev][k].used && sets[dev][k].key == caller) { slot = k; break; }
        if (!sets[dev][k].used) { slot = k; break; }
    }
    SideStreams &X = sets[dev][slot];
    if (!X.used) {
        X.used = true;
        X.key = caller;
        bool good = hipStreamCreateWithFlags(&X.s1, hipStreamNonBlocking) == hipSuccess &&
                    hipStreamCreateWithFlags(&X.s2, hipStreamNonBlocking) == hipSuccess &&
                    hipStreamCreateWithFlags(&X.s3, hipStreamNonBlocking) == hipSuccess &&
                    hipEventCreateWithFlags(&X.e3a, hipEventDisableTiming) == hipSuccess &&
                    hipEventCreateWithFlags(&X.e3b, hipEventDisableTiming) == hipSuccess &&
                    hipEventCreateWithFlags(&X.e3c, hipEventDisableTiming) == hipSuccess &&
                    hipEventCreateWithFlags(&X.e3d, hipEventDisableTiming) == hipSuccess &&
                    hipEventCreateWithFlags(&X.fork, hipEventDisableTiming) == hipSuccess &&
                    hipEventCreateWithFlags(&X.join1, hipEventDisableTiming) == hipSuccess &&
                    hipEventCreateWithFlags(&X.join2, hipEventDisableTiming) == hipSuccess;
        X.ok = good;
        if (!good) (void)hipGetLastError();
    }
    return X;
}

}  // namespace
}  // namespace cpe

using namespace cpe;

extern "C" size_t cpe_detect_workspace_bytes(int32_t n, int32_t h, int32_t w)
{
    if (n <= 0 || h <= 0 || w <= 0) return 0;
    return make_layout(n, h, w).total;
}

extern "C" int32_t cpe_detect_workspace_plane(int32_t n, int32_t h, int32_t w, int32_t plane, size_t *offset,
                                              size_t *bytes_per_frame)
{
    CPE_CHECK_ARG(n > 0 && h > 0 && w > 0 && plane >= 0 && plane <= P_G7 + 2 && offset && bytes_per_frame,
                  "cpe_detect_workspace_plane: bad argument");
    Layout L = make_layout(n, h, w);
    if (plane == P_G7 + 1) plane = P_LAB0;
    else if (plane == P_G7 + 2) plane = P_SW;
    *offset = L.off[plane];
    *bytes_per_frame = L.bytes_per_frame[plane];
    return CPE_OK;
}

extern "C" int32_t cpe_detect_grid_batch(const uint8_t *gray, int32_t n, int32_t h, int32_t w, void *ws, size_t ws_bytes,
                                         double *xy, int32_t *id, int32_t *n_pts, double *center, int32_t *status,
                                         void *stream)
{
    return cpe_detect_grid_batch_ex(gray, n, h, w, nullptr, ws, ws_bytes, xy, id, n_pts, center, status, stream);
}

namespace cpe { namespace {
// L channel of cv2.cvtColor(BGR2LAB) of a colour frame (util_cylinder.py:1840-1841), [ext] OpenCV 4.5.5 RGB2Lab_b: sRGB gamma
// table per channel, Y row of sRGB -> XYZ (D65) in 12-bit fixed point, cube-root table folded into LY (tools/gen_lab_lut.py
// colour).  R = G = B gives region.hip's c_lab_l.
__constant__ uint16_t c_gamma[256] = { 0, 1, 1, 2, 2, 3, 4, 4, 5, 6, 6, 7, 8, 8, 9, 10, 11, 11, 12, 13, 14, 15, 16, 17, 19, 20, 21, 22, 24, 25, 26, 28, 29, 31, 33, 34, 36, 38, 40, 41, 43, 45, 47, 49, 51, 54, 56, 58, 60, 63, 65, 68, 70, 73, 75, 78, 81, 83, 86, 89, 92, 95, 98, 101, 105, 108, 111, 115, 118, 121, 125, 129, 132, 136, 140, 144, 147, 151, 155, 160, 164, 168, 172, 176, 181, 185, 190, 194, 199, 204, 209, 213, 218, 223, 228, 233, 239, 244, 249, 255, 260, 265, 271, 277, 282, 288, 294, 300, 306, 312, 318, 324, 331, 337, 343, 350, 356, 363, 370, 376, 383, 390, 397, 404, 411, 418, 426, 433, 440, 448, 455, 463, 471, 478, 486, 494, 502, 510, 518, 527, 535, 543, 552, 560, 569, 578, 586, 595, 604, 613, 622, 631, 641, 650, 659, 669, 678, 688, 698, 707, 717, 727, 737, 747, 757, 768, 778, 788, 799, 809, 820, 831, 842, 852, 863, 875, 886, 897, 908, 920, 931, 943, 954, 966, 978, 990, 1002, 1014, 1026, 1038, 1050, 1063, 1075, 1088, 1101, 1113, 1126, 1139, 1152, 1165, 1178, 1192, 1205, 1218, 1232, 1245, 1259, 1273, 1287, 1301, 1315, 1329, 1343, 1357, 1372, 1386, 1401, 1415, 1430, 1445, 1460, 1475, 1490, 1505, 1521, 1536, 1551, 1567, 1583, 1598, 1614, 1630, 1646, 1662, 1678, 1695, 1711, 1728, 1744, 1761, 1778, 1794, 1811, 1828, 1846, 1863, 1880, 1897, 1915, 1933, 1950, 1968, 1986, 2004, 2022, 2040 };
__constant__ uint8_t c_ly[2041] = { 0, 1, 2, 3, 5, 6, 7, 8, 9, 10, 11, 12, 14, 15, 16, 17, 18, 19, 20, 21, 23, 24, 25, 26, 27, 27, 28, 29, 30, 31, 32, 33, 33, 34, 35, 36, 36, 37, 38, 38, 39, 40, 40, 41, 42, 42, 43, 43, 44, 45, 45, 46, 46, 47, 47, 48, 48, 49, 50, 50, 51, 51, 52, 52, 53, 53, 54, 54, 54, 55, 55, 56, 56, 57, 57, 58, 58, 58, 59, 59, 60, 60, 61, 61, 61, 62, 62, 63, 63, 63, 64, 64, 65, 65, 65, 66, 66, 66, 67, 67, 68, 68, 68, 69, 69, 69, 70, 70, 70, 71, 71, 71, 72, 72, 72, 73, 73, 73, 74, 74, 74, 75, 75, 75, 76, 76, 76, 77, 77, 77, 77, 78, 78, 78, 79, 79, 79, 80, 80, 80, 80, 81, 81, 81, 82, 82, 82, 82, 83, 83, 83, 83, 84, 84, 84, 85, 85, 85, 85, 86, 86, 86, 86, 87, 87, 87, 87, 88, 88, 88, 88, 89, 89, 89, 89, 90, 90, 90, 90, 91, 91, 91, 91, 92, 92, 92, 92, 93, 93, 93, 93, 94, 94, 94, 94, 95, 95, 95, 95, 95, 96, 96, 96, 96, 97, 97, 97, 97, 97, 98, 98, 98, 98, 99, 99, 99, 99, 99, 100, 100, 100, 100, 101, 101, 101, 101, 101, 102, 102, 102, 102, 102, 103, 103, 103, 103, 103, 104, 104, 104, 104, 104, 105, 105, 105, 105, 105, 106, 106, 106, 106, 106, 107, 107, 107, 107, 107, 108, 108, 108, 108, 108, 109, 109, 109, 109, 109, 109, 110, 110, 110, 110, 110, 111, 111, 111, 111, 111, 112, 112, 112, 112, 112, 112, 113, 113, 113, 113, 113, 114, 114, 114, 114, 114, 114, 115, 115, 115, 115, 115, 115, 116, 116, 116, 116, 116, 116, 117, 117, 117, 117, 117, 117, 118, 118, 118, 118, 118, 119, 119, 119, 119, 119, 119, 119, 120, 120, 120, 120, 120, 120, 121, 121, 121, 121, 121, 121, 122, 122, 122, 122, 122, 122, 123, 123, 123, 123, 123, 123, 124, 124, 124, 124, 124, 124, 124, 125, 125, 125, 125, 125, 125, 126, 126, 126, 126, 126, 126, 126, 127, 127, 127, 127, 127, 127, 128, 128, 128, 128, 128, 128, 128, 129, 129, 129, 129, 129, 129, 129, 130, 130, 130, 130, 130, 130, 130, 131, 131, 131, 131, 131, 131, 131, 132, 132, 132, 132, 132, 132, 132, 133, 133, 133, 133, 133, 133, 133, 134, 134, 134, 134, 134, 134, 134, 135, 135, 135, 135, 135, 135, 135, 135, 136, 136, 136, 136, 136, 136, 136, 137, 137, 137, 137, 137, 137, 137, 138, 138, 138, 138, 138, 138, 138, 138, 139, 139, 139, 139, 139, 139, 139, 139, 140, 140, 140, 140, 140, 140, 140, 141, 141, 141, 141, 141, 141, 141, 141, 142, 142, 142, 142, 142, 142, 142, 142, 143, 143, 143, 143, 143, 143, 143, 143, 144, 144, 144, 144, 144, 144, 144, 144, 145, 145, 145, 145, 145, 145, 145, 145, 146, 146, 146, 146, 146, 146, 146, 146, 147, 147, 147, 147, 147, 147, 147, 147, 147, 148, 148, 148, 148, 148, 148, 148, 148, 149, 149, 149, 149, 149, 149, 149, 149, 149, 150, 150, 150, 150, 150, 150, 150, 150, 151, 151, 151, 151, 151, 151, 151, 151, 151, 152, 152, 152, 152, 152, 152, 152, 152, 152, 153, 153, 153, 153, 153, 153, 153, 153, 154, 154, 154, 154, 154, 154, 154, 154, 154, 155, 155, 155, 155, 155, 155, 155, 155, 155, 156, 156, 156, 156, 156, 156, 156, 156, 156, 156, 157, 157, 157, 157, 157, 157, 157, 157, 157, 158, 158, 158, 158, 158, 158, 158, 158, 158, 159, 159, 159, 159, 159, 159, 159, 159, 159, 159, 160, 160, 160, 160, 160, 160, 160, 160, 160, 161, 161, 161, 161, 161, 161, 161, 161, 161, 161, 162, 162, 162, 162, 162, 162, 162, 162, 162, 163, 163, 163, 163, 163, 163, 163, 163, 163, 163, 164, 164, 164, 164, 164, 164, 164, 164, 164, 164, 165, 165, 165, 165, 165, 165, 165, 165, 165, 165, 166, 166, 166, 166, 166, 166, 166, 166, 166, 166, 167, 167, 167, 167, 167, 167, 167, 167, 167, 167, 168, 168, 168, 168, 168, 168, 168, 168, 168, 168, 168, 169, 169, 169, 169, 169, 169, 169, 169, 169, 169, 170, 170, 170, 170, 170, 170, 170, 170, 170, 170, 170, 171, 171, 171, 171, 171, 171, 171, 171, 171, 171, 172, 172, 172, 172, 172, 172, 172, 172, 172, 172, 172, 173, 173, 173, 173, 173, 173, 173, 173, 173, 173, 173, 174, 174, 174, 174, 174, 174, 174, 174, 174, 174, 174, 175, 175, 175, 175, 175, 175, 175, 175, 175, 175, 176, 176, 176, 176, 176, 176, 176, 176, 176, 176, 176, 176, 177, 177, 177, 177, 177, 177, 177, 177, 177, 177, 177, 178, 178, 178, 178, 178, 178, 178, 178, 178, 178, 178, 179, 179, 179, 179, 179, 179, 179, 179, 179, 179, 179, 180, 180, 180, 180, 180, 180, 180, 180, 180, 180, 180, 180, 181, 181, 181, 181, 181, 181, 181, 181, 181, 181, 181, 181, 182, 182, 182, 182, 182, 182, 182, 182, 182, 182, 182, 183, 183, 183, 183, 183, 183, 183, 183, 183, 183, 183, 183, 184, 184, 184, 184, 184, 184, 184, 184, 184, 184, 184, 184, 185, 185, 185, 185, 185, 185, 185, 185, 185, 185, 185, 185, 186, 186, 186, 186, 186, 186, 186, 186, 186, 186, 186, 186, 187, 187, 187, 187, 187, 187, 187, 187, 187, 187, 187, 187, 187, 188, 188, 188, 188, 188, 188, 188, 188, 188, 188, 188, 188, 189, 189, 189, 189, 189, 189, 189, 189, 189, 189, 189, 189, 190, 190, 190, 190, 190, 190, 190, 190, 190, 190, 190, 190, 190, 191, 191, 191, 191, 191, 191, 191, 191, 191, 191, 191, 191, 191, 192, 192, 192, 192, 192, 192, 192, 192, 192, 192, 192, 192, 193, 193, 193, 193, 193, 193, 193, 193, 193, 193, 193, 193, 193, 194, 194, 194, 194, 194, 194, 194, 194, 194, 194, 194, 194, 194, 195, 195, 195, 195, 195, 195, 195, 195, 195, 195, 195, 195, 195, 196, 196, 196, 196, 196, 196, 196, 196, 196, 196, 196, 196, 196, 196, 197, 197, 197, 197, 197, 197, 197, 197, 197, 197, 197, 197, 197, 198, 198, 198, 198, 198, 198, 198, 198, 198, 198, 198, 198, 198, 199, 199, 199, 199, 199, 199, 199, 199, 199, 199, 199, 199, 199, 199, 200, 200, 200, 200, 200, 200, 200, 200, 200, 200, 200, 200, 200, 200, 201, 201, 201, 201, 201, 201, 201, 201, 201, 201, 201, 201, 201, 202, 202, 202, 202, 202, 202, 202, 202, 202, 202, 202, 202, 202, 202, 203, 203, 203, 203, 203, 203, 203, 203, 203, 203, 203, 203, 203, 203, 204, 204, 204, 204, 204, 204, 204, 204, 204, 204, 204, 204, 204, 204, 204, 205, 205, 205, 205, 205, 205, 205, 205, 205, 205, 205, 205, 205, 205, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 207, 207, 207, 207, 207, 207, 207, 207, 207, 207, 207, 207, 207, 207, 207, 208, 208, 208, 208, 208, 208, 208, 208, 208, 208, 208, 208, 208, 208, 209, 209, 209, 209, 209, 209, 209, 209, 209, 209, 209, 209, 209, 209, 209, 210, 210, 210, 210, 210, 210, 210, 210, 210, 210, 210, 210, 210, 210, 210, 211, 211, 211, 211, 211, 211, 211, 211, 211, 211, 211, 211, 211, 211, 211, 212, 212, 212, 212, 212, 212, 212, 212, 212, 212, 212, 212, 212, 212, 212, 213, 213, 213, 213, 213, 213, 213, 213, 213, 213, 213, 213, 213, 213, 213, 214, 214, 214, 214, 214, 214, 214, 214, 214, 214, 214, 214, 214, 214, 214, 215, 215, 215, 215, 215, 215, 215, 215, 215, 215, 215, 215, 215, 215, 215, 215, 216, 216, 216, 216, 216, 216, 216, 216, 216, 216, 216, 216, 216, 216, 216, 217, 217, 217, 217, 217, 217, 217, 217, 217, 217, 217, 217, 217, 217, 217, 217, 218, 218, 218, 218, 218, 218, 218, 218, 218, 218, 218, 218, 218, 218, 218, 218, 219, 219, 219, 219, 219, 219, 219, 219, 219, 219, 219, 219, 219, 219, 219, 219, 220, 220, 220, 220, 220, 220, 220, 220, 220, 220, 220, 220, 220, 220, 220, 220, 221, 221, 221, 221, 221, 221, 221, 221, 221, 221, 221, 221, 221, 221, 221, 221, 222, 222, 222, 222, 222, 222, 222, 222, 222, 222, 222, 222, 222, 222, 222, 222, 223, 223, 223, 223, 223, 223, 223, 223, 223, 223, 223, 223, 223, 223, 223, 223, 223, 224, 224, 224, 224, 224, 224, 224, 224, 224, 224, 224, 224, 224, 224, 224, 224, 225, 225, 225, 225, 225, 225, 225, 225, 225, 225, 225, 225, 225, 225, 225, 225, 225, 226, 226, 226, 226, 226, 226, 226, 226, 226, 226, 226, 226, 226, 226, 226, 226, 226, 227, 227, 227, 227, 227, 227, 227, 227, 227, 227, 227, 227, 227, 227, 227, 227, 227, 228, 228, 228, 228, 228, 228, 228, 228, 228, 228, 228, 228, 228, 228, 228, 228, 228, 229, 229, 229, 229, 229, 229, 229, 229, 229, 229, 229, 229, 229, 229, 229, 229, 229, 230, 230, 230, 230, 230, 230, 230, 230, 230, 230, 230, 230, 230, 230, 230, 230, 230, 231, 231, 231, 231, 231, 231, 231, 231, 231, 231, 231, 231, 231, 231, 231, 231, 231, 231, 232, 232, 232, 232, 232, 232, 232, 232, 232, 232, 232, 232, 232, 232, 232, 232, 232, 233, 233, 233, 233, 233, 233, 233, 233, 233, 233, 233, 233, 233, 233, 233, 233, 233, 233, 234, 234, 234, 234, 234, 234, 234, 234, 234, 234, 234, 234, 234, 234, 234, 234, 234, 234, 235, 235, 235, 235, 235, 235, 235, 235, 235, 235, 235, 235, 235, 235, 235, 235, 235, 235, 236, 236, 236, 236, 236, 236, 236, 236, 236, 236, 236, 236, 236, 236, 236, 236, 236, 236, 237, 237, 237, 237, 237, 237, 237, 237, 237, 237, 237, 237, 237, 237, 237, 237, 237, 237, 238, 238, 238, 238, 238, 238, 238, 238, 238, 238, 238, 238, 238, 238, 238, 238, 238, 238, 239, 239, 239, 239, 239, 239, 239, 239, 239, 239, 239, 239, 239, 239, 239, 239, 239, 239, 239, 240, 240, 240, 240, 240, 240, 240, 240, 240, 240, 240, 240, 240, 240, 240, 240, 240, 240, 241, 241, 241, 241, 241, 241, 241, 241, 241, 241, 241, 241, 241, 241, 241, 241, 241, 241, 241, 242, 242, 242, 242, 242, 242, 242, 242, 242, 242, 242, 242, 242, 242, 242, 242, 242, 242, 242, 243, 243, 243, 243, 243, 243, 243, 243, 243, 243, 243, 243, 243, 243, 243, 243, 243, 243, 243, 244, 244, 244, 244, 244, 244, 244, 244, 244, 244, 244, 244, 244, 244, 244, 244, 244, 244, 244, 245, 245, 245, 245, 245, 245, 245, 245, 245, 245, 245, 245, 245, 245, 245, 245, 245, 245, 245, 245, 246, 246, 246, 246, 246, 246, 246, 246, 246, 246, 246, 246, 246, 246, 246, 246, 246, 246, 246, 247, 247, 247, 247, 247, 247, 247, 247, 247, 247, 247, 247, 247, 247, 247, 247, 247, 247, 247, 248, 248, 248, 248, 248, 248, 248, 248, 248, 248, 248, 248, 248, 248, 248, 248, 248, 248, 248, 248, 249, 249, 249, 249, 249, 249, 249, 249, 249, 249, 249, 249, 249, 249, 249, 249, 249, 249, 249, 249, 250, 250, 250, 250, 250, 250, 250, 250, 250, 250, 250, 250, 250, 250, 250, 250, 250, 250, 250, 250, 251, 251, 251, 251, 251, 251, 251, 251, 251, 251, 251, 251, 251, 251, 251, 251, 251, 251, 251, 251, 252, 252, 252, 252, 252, 252, 252, 252, 252, 252, 252, 252, 252, 252, 252, 252, 252, 252, 252, 252, 253, 253, 253, 253, 253, 253, 253, 253, 253, 253, 253, 253, 253, 253, 253, 253, 253, 253, 253, 253, 254, 254, 254, 254, 254, 254, 254, 254, 254, 254, 254, 254, 254, 254, 254, 254, 254, 254, 254, 254, 254, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255 };
__global__ __launch_bounds__(256) void k_bgr2labl(const uint8_t *__restrict__ bgr, size_t npx, uint8_t *__restrict__ L)
{
    __shared__ uint16_t s_g[256];
    __shared__ uint8_t s_ly[2048];
    s_g[threadIdx.x] = c_gamma[threadIdx.x];
    for (int i = threadIdx.x; i < 2041; i += 256) s_ly[i] = c_ly[i];
    __syncthreads();
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < npx; p += (size_t)gridDim.x * 256) {
        const int B = s_g[bgr[3 * p]], G = s_g[bgr[3 * p + 1]], R = s_g[bgr[3 * p + 2]];
        L[p] = s_ly[(R * 871 + G * 2929 + B * 296 + (1 << 11)) >> 12];
    }
}
__global__ __launch_bounds__(256) void k_bgr2gray(const uint8_t *__restrict__ bgr, size_t npx, uint8_t *__restrict__ gray);
} }

// the call behind both entry points: grey frames (bgr == null) or true-colour frames (gray == null; the grey plane and the
// L plane are made in the workspace first)
static int32_t detect_impl(const uint8_t *gray, const uint8_t *bgr, int32_t n, int32_t h, int32_t w, const CpeDetectParams *params,
                           void *ws, size_t ws_bytes, double *xy, int32_t *id, int32_t *n_pts,
                           double *center, int32_t *status, void *stream)
{
    CpeDetectParams prm = {0, 7, 1.0, CPE_TARGET_CYLINDER, 0};
    if (params) prm = *params;
    CPE_CHECK_ARG(prm.subpixel == 0 || (prm.subpixel_window >= 1 && prm.subpixel_window <= 13 && prm.subpixel_step > 0),
                  "cpe_detect_grid_batch_ex: bad sub-pixel parameters");
    CPE_CHECK_ARG(prm.target == CPE_TARGET_CYLINDER || prm.target == CPE_TARGET_PLANE, "cpe_detect_grid_batch_ex: unknown target %d", prm.target);
    CPE_CHECK_ARG(!(prm.target == CPE_TARGET_PLANE && prm.subpixel), "cpe_detect_grid_batch_ex: no sub-pixel refinement for the planar target");
    const int planar = prm.target == CPE_TARGET_PLANE ? 1 : 0;
    CPE_CHECK_ARG((gray || bgr) && xy && id && n_pts && center && status, "cpe_detect_grid_batch: null pointer");
    CPE_CHECK_ARG(!(bgr && (planar || prm.subpixel)), "cpe_detect_grid_bgr_batch_ex: colour frames: cylinder target without sub-pixel refinement only");
    CPE_CHECK_ARG(n >= 0 && h >= 64 && w >= 64 && h <= 4096 && w <= 4096,
                  "cpe_detect_grid_batch: need n>=0 and 64 <= h,w <= 4096 (got %d,%d,%d)", n, h, w);
    if (n == 0) return CPE_OK;
    Layout L = make_layout(n, h, w);
    if (!ws || ws_bytes < L.total) {
        cpe::set_error("cpe_detect_grid_batch: workspace too small (%zu < %zu)", ws_bytes, L.total);
        return CPE_ERR_WORKSPACE;
    }
    CPE_CHECK_ARG(((uintptr_t)ws & 255) == 0, "cpe_detect_grid_batch: workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    uint8_t *base = (uint8_t *)ws;
#define PL(T, p) ((T *)(base + L.off[p]))
    FrameState *st = PL(FrameState, P_STATE);
    // colour input: the grey plane lives in the workspace; the L plane borrows the disc plane of the region stage, which is
    // first written (cleared) after CLAHE has read L
    uint8_t *lplane = bgr ? PL(uint8_t, P_EXT) : nullptr;
    if (bgr) gray = PL(uint8_t, P_GRAYIN);
    RegionBuffers R;
    R.cl = PL(uint8_t, P_CL); R.ext = PL(uint8_t, P_EXT); R.mc = PL(uint8_t, P_MASK_CONTOUR); R.touch = PL(uint8_t, P_TOUCH);
    R.lab = PL(int, P_LAB0); R.cnt = PL(int, P_LAB1); R.roots = PL(int, P_ROOTS); R.nrect = PL(int, P_NRECT); R.lab2 = PL(int, P_LAB2); R.cnt2 = PL(int, P_LAB3);
    R.sw = PL(int, P_SW); R.hl = PL(int2, P_HL); R.bl = PL(int2, P_BL); R.tl = PL(int2, P_TL); R.bk = PL(int, P_BK); R.bits = PL(uint32_t, P_BITS); R.pool = PL(uint32_t, P_POOL); R.blob_ch = PL(unsigned short, P_BLOB_CH); R.maxch = region_maxch(h, w); R.maxdf = region_maxdf(h, w); R.gmid = PL(double, P_GMID); R.hpar = PL(int, P_HPAR); R.htime = PL(uint8_t, P_HTIME); R.hist = PL(unsigned int, P_HIST); R.lut = PL(uint8_t, P_LUT);
    R.blobs = PL(BlobRec, P_BLOBS); R.blob_d = PL(int, P_BLOB_D); R.order = PL(int, P_ORDER); R.dists = PL(double, P_DISTS);
    R.groups = PL(Group, P_GROUPS); R.best = PL(unsigned long long, P_BEST); R.lohi = PL(int, P_LOHI); R.hull = PL(int, P_HULL);
    MaskBuffers M;
    M.binary = PL(uint8_t, P_BINARY); M.hmask = PL(uint8_t, P_HMASK); M.vmask = PL(uint8_t, P_VMASK);
    M.joints_mask = PL(uint8_t, P_JOINTS_MASK); M.tmpA = PL(uint8_t, P_TMPA); M.tmpB = PL(uint8_t, P_TMPB);
    M.g19 = PL(uint8_t, P_G19); M.cm = PL(uint8_t, P_CM); M.mc = R.mc; M.roi_h = PL(uint8_t, P_ROI_H);
    M.roi_v = PL(uint8_t, P_ROI_V); M.base_h = PL(uint8_t, P_BASE_H); M.base_v = PL(uint8_t, P_BASE_V);
    M.exp_h = PL(uint8_t, P_EXP_H); M.exp_v = PL(uint8_t, P_EXP_V); M.touch = R.touch; M.bits = R.bits;
    M.lab = R.lab; M.roots = R.roots; M.jtmp = PL(int, P_JTMP); M.joints = PL(int, P_JOINTS); M.verts = PL(int, P_VERTS);
    M.best = R.best; M.segs = PL(SegRec, P_SEGS);
    M.lab_p = PL(int, P_LABP); M.lab_s = PL(int, P_LABS); M.roots_p = PL(int, P_ROOTSP); M.roots_s = PL(int, P_ROOTSS);
    M.best_s = PL(unsigned long long, P_BEST2);
    M.fl_j = PL(unsigned long long, P_FLJ); M.jbits = PL(uint32_t, P_HPAR);
    // three chains that only meet in masks_stage: ridge mask -> line masks -> joints (stream 1), saturated spot
    // (stream 2), region (the caller's stream).  The side chains are mostly ALU / latency bound and fill the CUs the
    // region stage's serial kernels leave idle.  The helper streams and their events are per device and shared by all
    // callers: the enqueue below (fork .. join, host side only, microseconds per kernel) runs under the device's mutex,
    // so two host threads never interleave their forks and joins.
    SideStreams &X = side_streams(s);
    std::unique_lock<std::mutex> lk(X.mu, std::defer_lock);
    if (X.ok) lk.lock();
    bool forked = false;
    auto enqueue = [&]() -> int {
        int rc;
        CPE_LAUNCH_BEGIN();
        CPE_KLAUNCH(k_state_init, dim3((n + 63) / 64), dim3(64), 0, s, st, n, R.best, M.best_s, R.nrect);
        CPE_CHECK_LAUNCH("k_state_init");
        if (bgr) {      // BGR2GRAY (load_and_preprocess_image, mask_roi_around_center) and the L channel of BGR2LAB (detect_largest_blob)
            const size_t npx = (size_t)n * h * w;
            CPE_KLAUNCH(k_bgr2gray, dim3((unsigned)(((npx + 3) / 4 + 255) / 256)), dim3(256), 0, s, bgr, npx, PL(uint8_t, P_GRAYIN));
            CPE_KLAUNCH(k_bgr2labl, dim3((unsigned)std::min<size_t>((npx + 255) / 256, 1 << 16)), dim3(256), 0, s, bgr, npx, lplane);
            CPE_CHECK_LAUNCH("colour planes");
        }
        if (X.ok) {
            CPE_CHECK_HIP(hipEventRecord(X.fork, s));
            CPE_CHECK_HIP(hipStreamWaitEvent(X.s1, X.fork, 0));
            CPE_CHECK_HIP(hipStreamWaitEvent(X.s2, X.fork, 0));
            forked = true;
        }
        hipStream_t s1 = X.ok ? X.s1 : s, s2 = X.ok ? X.s2 : s;
        if ((rc = cpe_preprocess_batch(gray, n, h, w, M.binary, (void *)s1)) != CPE_OK) return rc;
        if ((rc = joints_mask_stage(n, h, w, M, st, s1)) != CPE_OK) return rc;
        if ((rc = spot_stage(gray, n, h, w, M, st, s2, planar)) != CPE_OK) return rc;
        if (X.ok) {   // ends of the joints and the spot chain (nothing more is enqueued on s1 / s2 before the join below)
            CPE_CHECK_HIP(hipEventRecord(X.join1, X.s1));
            CPE_CHECK_HIP(hipEventRecord(X.join2, X.s2));
        }
        RegionSide rside = {X.s3, X.e3a, X.e3b, X.e3c, X.e3d, X.join1, X.join2};
        if (planar) { if ((rc = region_stage_plane(gray, n, h, w, R, st, s)) != CPE_OK) return rc; }
        else if ((rc = region_stage(gray, n, h, w, 4.5, R, st, s, X.ok ? &rside : nullptr, lplane)) != CPE_OK) return rc;
        if (X.ok) {
            CPE_CHECK_HIP(hipStreamWaitEvent(s, X.join1, 0));
            CPE_CHECK_HIP(hipStreamWaitEvent(s, X.join2, 0));
            forked = false;
        }
        M.lab_h = PL(int, P_LAB0); M.lab_v = PL(int, P_LAB1);
        // the 7x7 blur of the indexing step only needs the region rectangle, and only the lines kernel reads the joints: both
        // run on the (now idle) spot stream beside the fragment chains of the masks stage
        if (X.ok) {
            CPE_CHECK_HIP(hipEventRecord(X.fork, s));
            CPE_CHECK_HIP(hipStreamWaitEvent(X.s2, X.fork, 0));
            forked = true;
            if ((rc = bgr ? blur7_bgr(bgr, n, h, w, st, PL(uint8_t, P_G7), X.s2) : blur7_u8(gray, n, h, w, st, PL(uint8_t, P_G7), X.s2)) != CPE_OK) return rc;
        }
        if ((rc = masks_stage(gray, n, h, w, M, st, s, X.ok ? &rside : nullptr, planar, X.ok ? X.s2 : s)) != CPE_OK) return rc;   // joints: on s2 too
        if (X.ok) {
            CPE_CHECK_HIP(hipEventRecord(X.join2, X.s2));
            CPE_CHECK_HIP(hipStreamWaitEvent(s, X.join2, 0));
            forked = false;
        }
        else if ((rc = bgr ? blur7_bgr(bgr, n, h, w, st, PL(uint8_t, P_G7), s) : blur7_u8(gray, n, h, w, st, PL(uint8_t, P_G7), s)) != CPE_OK) return rc;
        if ((rc = lines_stage(PL(int, P_LAB0), PL(int, P_LAB1), M.exp_h, M.exp_v, PL(uint8_t, P_G7), n, h, w, M.joints, st, PL(void, P_LINES), xy, id,
                              n_pts, center, gray, prm.subpixel, prm.subpixel_window, prm.subpixel_step, PL(float, P_SUBPIX),
                              std::max(h, w) + 128, s, planar)) != CPE_OK)
            return rc;
        CPE_LAUNCH_BEGIN();
        CPE_KLAUNCH(k_finish, dim3((n + 63) / 64), dim3(64), 0, s, st, n, status, n_pts);
        CPE_CHECK_LAUNCH("k_finish");
        return CPE_OK;
    };
    const int rc = enqueue();
    if (X.ok && (forked || rc != CPE_OK)) {
        // an error between fork and join: whatever already runs on the helper streams still uses the caller's buffers,
        // so the caller's stream is made to wait for all of them before the error is reported
        hipStream_t hs[3] = {X.s1, X.s2, X.s3};
        hipEvent_t he[3] = {X.join1, X.join2, X.e3a};
        for (int k = 0; k < 3; k++)
            if (hipEventRecord(he[k], hs[k]) == hipSuccess) (void)hipStreamWaitEvent(s, he[k], 0);
        (void)hipGetLastError();
    }
    return rc;
#undef PL
}

extern "C" int32_t cpe_detect_grid_batch_ex(const uint8_t *gray, int32_t n, int32_t h, int32_t w, const CpeDetectParams *params,
                                            void *ws, size_t ws_bytes, double *xy, int32_t *id, int32_t *n_pts,
                                            double *center, int32_t *status, void *stream)
{
    CPE_CHECK_ARG(gray, "cpe_detect_grid_batch: null pointer");
    return detect_impl(gray, nullptr, n, h, w, params, ws, ws_bytes, xy, id, n_pts, center, status, stream);
}

extern "C" int32_t cpe_detect_grid_bgr_batch_ex(const uint8_t *bgr, int32_t n, int32_t h, int32_t w, const CpeDetectParams *params,
                                                void *ws, size_t ws_bytes, double *xy, int32_t *id, int32_t *n_pts,
                                                double *center, int32_t *status, void *stream)
{
    CPE_CHECK_ARG(bgr && (((uintptr_t)bgr) & 3) == 0, "cpe_detect_grid_bgr_batch_ex: bgr must be a 4-byte aligned device pointer");
    return detect_impl(nullptr, bgr, n, h, w, params, ws, ws_bytes, xy, id, n_pts, center, status, stream);
}

extern "C" int32_t cpe_detect_line_tables(const void *ws, size_t ws_bytes, int32_t n, int32_t h, int32_t w, int32_t frame,
                                          double *eq, int32_t *npts, double *pts, int32_t *n_lines, void *stream)
{
    CPE_CHECK_ARG(ws && eq && npts && pts && n_lines && n > 0 && frame >= 0 && frame < n && h >= 64 && w >= 64,
                  "cpe_detect_line_tables: bad argument");
    static_assert(CPE_MAXL == MAXL, "cpe.h and cpe_dev.h disagree on the line capacity");
    Layout L = make_layout(n, h, w);
    CPE_CHECK_ARG(ws_bytes >= L.total && ((uintptr_t)ws & 255) == 0, "cpe_detect_line_tables: not the workspace of an (n,h,w) call");
    return lines_export((const uint8_t *)ws + L.off[P_LINES], frame, eq, npts, pts, n_lines, (hipStream_t)stream);
}

// BGR2GRAY of the entry point for colour input (load_and_preprocess_image, util_cylinder.py:1781-1789)
namespace cpe { namespace {
__global__ __launch_bounds__(256) void k_bgr2gray(const uint8_t *__restrict__ bgr, size_t npx, uint8_t *__restrict__ gray)
{
    // 4 pixels (12 bytes in, 1 dword out) per thread; cv2.cvtColor 8-bit: (B*3735 + G*19235 + R*9798 + 2^14) >> 15
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t p0 = q * 4;
    if (p0 >= npx) return;
    if (p0 + 4 <= npx) {
        const uint32_t *src = (const uint32_t *)(bgr + p0 * 3);     // p0 * 3 is a multiple of 12
        const uint32_t a = src[0], b = src[1], c = src[2];
        const uint32_t px[4][3] = {{a & 255, (a >> 8) & 255, (a >> 16) & 255}, {a >> 24, b & 255, (b >> 8) & 255},
                                   {(b >> 16) & 255, b >> 24, c & 255}, {(c >> 8) & 255, (c >> 16) & 255, c >> 24}};
        uint32_t out = 0;
        for (int k = 0; k < 4; k++) out |= ((px[k][0] * 3735u + px[k][1] * 19235u + px[k][2] * 9798u + 16384u) >> 15) << (8 * k);
        *(uint32_t *)(gray + p0) = out;
    } else {
        for (size_t p = p0; p < npx; p++)
            gray[p] = (uint8_t)((bgr[3 * p] * 3735u + bgr[3 * p + 1] * 19235u + bgr[3 * p + 2] * 9798u + 16384u) >> 15);
    }
}
} }

extern "C" int32_t cpe_bgr2gray_batch(const uint8_t *bgr, int32_t n, int32_t h, int32_t w, uint8_t *gray, void *stream)
{
    CPE_CHECK_ARG(bgr && gray && n >= 0 && h > 0 && w > 0, "cpe_bgr2gray_batch: bad argument");
    CPE_CHECK_ARG(((uintptr_t)bgr & 3) == 0 && ((uintptr_t)gray & 3) == 0, "cpe_bgr2gray_batch: buffers must be 4-byte aligned");
    if (n == 0) return CPE_OK;
    const size_t npx = (size_t)n * h * w;
    const size_t quads = (npx + 3) / 4;
    CPE_LAUNCH_BEGIN();
    CPE_KLAUNCH(k_bgr2gray, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, bgr, npx, gray);
    CPE_CHECK_LAUNCH("k_bgr2gray");
    return CPE_OK;
}

// cv2.findContours(mask, RETR_EXTERNAL, .) reduced to what its callers keep of the hierarchy: the raster-first pixels of the
// components that do NOT lie inside a hole of another component (tests of the RETR_EXTERNAL rule, util_cylinder.py:161,1817)
namespace cpe { namespace {
__global__ __launch_bounds__(256) void k_list_external(const int *__restrict__ roots, const FrameState *__restrict__ st, int w,
                                                       const unsigned long long *__restrict__ outside, size_t plane_words,
                                                       int *__restrict__ first_px, int cap, int *__restrict__ count)
{
    const int f = blockIdx.y;
    const int ncomp = min(st[f].n_roots, MAXROOTS);
    for (int k = blockIdx.x * 256 + threadIdx.x; k < ncomp; k += gridDim.x * 256) {
        const int root = roots[(size_t)f * MAXROOTS + k];
        if (!comp_is_external(outside + f * plane_words, w, root, 0)) continue;
        const int q = atomicAdd(&count[f], 1);
        if (q < cap) first_px[(size_t)f * cap + q] = root;
    }
}
} }

extern "C" int32_t cpe_debug_external_components(const uint8_t *mask, int32_t n, int32_t h, int32_t w, void *ws, size_t ws_bytes,
                                                 int32_t *first_px, int32_t cap, int32_t *count, void *stream)
{
    CPE_CHECK_ARG(mask && ws && first_px && count && n > 0 && h >= 64 && w >= 64 && w <= 4096 && cap > 0, "cpe_debug_external_components: bad argument");
    Layout L = make_layout(n, h, w);
    CPE_CHECK_ARG(ws_bytes >= L.total && ((uintptr_t)ws & 255) == 0, "cpe_debug_external_components: workspace too small or misaligned");
    uint8_t *base = (uint8_t *)ws;
    hipStream_t s = (hipStream_t)stream;
    FrameState *st = (FrameState *)(base + L.off[P_STATE]);
    int *roots = (int *)(base + L.off[P_ROOTS]);
    int rc;
    CPE_LAUNCH_BEGIN();
    CPE_KLAUNCH(k_state_init, dim3((n + 63) / 64), dim3(64), 0, s, st, n, (unsigned long long *)nullptr, (unsigned long long *)nullptr, (int *)nullptr);
    CPE_CHECK_HIP(hipMemsetAsync(count, 0, (size_t)n * sizeof(int), s));
    if ((rc = ccl_run(mask, n, h, w, 0, 0, 1, (int *)(base + L.off[P_LAB0]), roots, false, nullptr, 0, nullptr, 0, nullptr, st, s, 1, 1, 0)) != CPE_OK) return rc;
    const size_t bit_words = (size_t)n * h * bit_row_words(w), fl_words = (size_t)h * bit_row_words(w) / 2;
    unsigned long long *bgw = (unsigned long long *)(base + L.off[P_BITS]);
    unsigned long long *out = (unsigned long long *)((uint32_t *)(base + L.off[P_BITS]) + ((bit_words + 1) & ~(size_t)1));
    if ((rc = outside_flood(mask, n, h, w, st, 0, bgw, out, fl_words, s)) != CPE_OK) return rc;
    CPE_KLAUNCH(k_list_external, dim3(32, n), dim3(256), 0, s, (const int *)roots, (const FrameState *)st, w, (const unsigned long long *)out, fl_words,
                first_px, cap, count);
    CPE_CHECK_LAUNCH("k_list_external");
    return CPE_OK;
}

// Stand-alone labelling pass over the workspace's label planes (profiling / tests): labels of
// {(img > thr) != invert} land in the CPE_PLANE_LABELS plane.
extern "C" int32_t cpe_debug_ccl(const uint8_t *img, int32_t n, int32_t h, int32_t w, int32_t thr, int32_t invert,
                                 int32_t conn8, int32_t count_mode, int32_t want_bbox, int32_t want_roots, void *ws,
                                 size_t ws_bytes, void *stream)
{
    CPE_CHECK_ARG(img && ws && n > 0 && h >= 64 && w >= 64, "cpe_debug_ccl: bad argument");
    Layout L = make_layout(n, h, w);
    CPE_CHECK_ARG(ws_bytes >= L.total && ((uintptr_t)ws & 255) == 0, "cpe_debug_ccl: workspace too small or misaligned");
    uint8_t *base = (uint8_t *)ws;
    hipStream_t s = (hipStream_t)stream;
    FrameState *st = (FrameState *)(base + L.off[P_STATE]);
    return ccl_run(img, n, h, w, thr, invert, conn8, (int *)(base + L.off[P_LAB0]), want_roots ? (int *)(base + L.off[P_ROOTS]) : nullptr,
                   invert != 0, (uint8_t *)(base + L.off[P_TOUCH]), count_mode, (int *)(base + L.off[P_LAB1]), (want_bbox >> 1) & 1,
                   (want_bbox & 1) ? (int *)(base + L.off[P_NRECT]) : nullptr, st, s);
}
