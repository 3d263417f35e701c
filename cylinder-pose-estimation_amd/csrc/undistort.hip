// Row f-3 -- the undistortion pre-step in front of detect_grid (utils/iotool.py:22-39: cv2.undistort(image, K, coeffs)).
// [ext] OpenCV 4.5.5 semantics restated exactly as in oracle/src/orc_undistort.c (parity unpinned vs cv2).
//
// cv2.undistort rebuilds its fixed-point map for every image; the map only depends on the camera, so it is built once
// (`cpe_undistort_map`) and every frame is one gather pass (`cpe_remap_bilinear_batch`):
//   k_undistort_map   one thread per image row.  OpenCV walks a row with running sums (_x += ir[0] ...), so column j
//                     carries the rounding of all columns before it: the row is the unit of parallelism.  Stripes of
//                     4096/cols rows get their own inverse matrix (principal point shifted by the stripe's first row).
//   k_remap_bilinear  4 pixels per thread: 16 B of (x, y) + 8 B of fractions in, 4 x 4 gathered source bytes, one dword
//                     out, for 8 frames per thread (the map bytes are read once per 8 frames).  Algorithmic bytes per
//                     frame: 2 B/px (source + destination) + 6/8 B/px of map: HBM bound.
#include "cpe_internal.h"

namespace {

struct CamArgs {
    double K[9];
    double c[12];
};

__device__ __forceinline__ int sat_int(double v)
{
    if (v >= 2147483647.0) return 2147483647;
    if (v <= -2147483648.0) return (-2147483647 - 1);
    return __double2int_rn(v);
}

__global__ __launch_bounds__(64) void k_undistort_map(CamArgs cam, int h, int w, int stripe0, int16_t *__restrict__ map_xy,
                                                      uint16_t *__restrict__ map_f)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= h) return;
    const int y0 = (row / stripe0) * stripe0, i = row - y0;
    const double *S = cam.K;
    const double a5 = S[5] - y0;   // Ar(1,2) = v0 - y0
    double d = S[0] * (S[4] * S[8] - a5 * S[7]) - S[1] * (S[3] * S[8] - a5 * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
    d = 1. / d;
    double ir[9];
    ir[0] = (S[4] * S[8] - a5 * S[7]) * d;
    ir[1] = (S[2] * S[7] - S[1] * S[8]) * d;
    ir[2] = (S[1] * a5 - S[2] * S[4]) * d;
    ir[3] = (a5 * S[6] - S[3] * S[8]) * d;
    ir[4] = (S[0] * S[8] - S[2] * S[6]) * d;
    ir[5] = (S[2] * S[3] - S[0] * a5) * d;
    ir[6] = (S[3] * S[7] - S[4] * S[6]) * d;
    ir[7] = (S[1] * S[6] - S[0] * S[7]) * d;
    ir[8] = (S[0] * S[4] - S[1] * S[3]) * d;
    const double k1 = cam.c[0], k2 = cam.c[1], p1 = cam.c[2], p2 = cam.c[3], k3 = cam.c[4], k4 = cam.c[5], k5 = cam.c[6],
                 k6 = cam.c[7], s1 = cam.c[8], s2 = cam.c[9], s3 = cam.c[10], s4 = cam.c[11];
    const double fx = S[0], fy = S[4], u0 = S[2], v0 = S[5];
    int16_t *m1 = map_xy + (size_t)row * w * 2;
    uint16_t *m2 = map_f + (size_t)row * w;
    double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
    for (int j = 0; j < w; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
        double ww = 1. / _w, x = _x * ww, y = _y * ww;
        double x2 = x * x, y2 = y * y;
        double r2 = x2 + y2, _2xy = 2 * x * y;
        double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
        double xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2) + s1 * r2 + s2 * r2 * r2);
        double yd = (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy + s3 * r2 + s4 * r2 * r2);
        double invProj = 1.;
        double u = fx * invProj * xd + u0;
        double v = fy * invProj * yd + v0;
        int iu = sat_int(u * 32), iv = sat_int(v * 32);
        m1[j * 2] = (int16_t)(iu >> 5);
        m1[j * 2 + 1] = (int16_t)(iv >> 5);
        m2[j] = (uint16_t)((iv & 31) * 32 + (iu & 31));
    }
}

__device__ __forceinline__ int remap_one(const uint8_t *__restrict__ s, int h, int w, int sx, int sy, int f)
{
    const int fx = f & 31, fy = (f >> 5) & 31;
    const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    // unconditional loads from clamped addresses (independent, one latency), masked afterwards: constant border 0
    const int cx0 = min(max(sx, 0), w - 1), cx1 = min(max(sx + 1, 0), w - 1);
    const int cy0 = min(max(sy, 0), h - 1), cy1 = min(max(sy + 1, 0), h - 1);
    const int a = s[(size_t)cy0 * w + cx0], b = s[(size_t)cy0 * w + cx1], c = s[(size_t)cy1 * w + cx0], d = s[(size_t)cy1 * w + cx1];
    const bool x0 = (unsigned)sx < (unsigned)w, x1 = (unsigned)(sx + 1) < (unsigned)w;
    const bool y0 = (unsigned)sy < (unsigned)h, y1 = (unsigned)(sy + 1) < (unsigned)h;
    const int sum = (x0 && y0 ? a : 0) * w00 + (x1 && y0 ? b : 0) * w01 + (x0 && y1 ? c : 0) * w10 + (x1 && y1 ? d : 0) * w11;
    const int r = (sum + (1 << 14)) >> 15;
    return r < 0 ? 0 : (r > 255 ? 255 : r);
}

// grid = (ceil(N / 1024), ceil(n / REMAP_FRAMES)): 4 consecutive pixels per thread (N = h*w must be a multiple of 4 for
// this kernel); the 24 map bytes of those pixels are read once and applied to REMAP_FRAMES frames
constexpr int REMAP_FRAMES = 8;
__global__ __launch_bounds__(256) void k_remap_bilinear4(const uint8_t *__restrict__ src, int n, int h, int w,
                                                         const int16_t *__restrict__ map_xy, const uint16_t *__restrict__ map_f,
                                                         uint8_t *__restrict__ dst)
{
    const int N = h * w;
    const int i0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= N) return;
    const int4 xy = *reinterpret_cast<const int4 *>(map_xy + (size_t)i0 * 2);
    const uint2 fr = *reinterpret_cast<const uint2 *>(map_f + i0);
    const int q[4] = {xy.x, xy.y, xy.z, xy.w};
    const int fq[4] = {(int)(fr.x & 0xFFFFu), (int)(fr.x >> 16), (int)(fr.y & 0xFFFFu), (int)(fr.y >> 16)};
    const int f0 = blockIdx.y * REMAP_FRAMES, f1 = min(f0 + REMAP_FRAMES, n);
    for (int f = f0; f < f1; f++) {
        const uint8_t *s = src + f * (size_t)N;
        unsigned out = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int sx = (int)(short)(q[k] & 0xFFFF), sy = q[k] >> 16;
            out |= (unsigned)remap_one(s, h, w, sx, sy, fq[k]) << (8 * k);
        }
        *reinterpret_cast<unsigned *>(dst + f * (size_t)N + i0) = out;
    }
}

__global__ __launch_bounds__(256) void k_remap_bilinear1(const uint8_t *__restrict__ src, int h, int w,
                                                         const int16_t *__restrict__ map_xy, const uint16_t *__restrict__ map_f,
                                                         uint8_t *__restrict__ dst)
{
    const int N = h * w;
    const size_t f = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    dst[f * (size_t)N + i] = (uint8_t)remap_one(src + f * (size_t)N, h, w, map_xy[2 * (size_t)i], map_xy[2 * (size_t)i + 1], map_f[i]);
}

// ---- second mode of row f-3: undistortImage(I, cameraParams, 'cubic') of the MATLAB entry point (utils/preProcessing.m:3-4)
// [ext] restated as in oracle/src/orc_undistort.c (parity unpinned vs MATLAB): f64 distortPoints per output pixel into a
// float32 (x, y) map, then cubic convolution (Keys, a = -1/2) in single precision, fill value outside the image.
struct MatlabCam { double fx, skew, cx, fy, cy, k1, k2, k3, p1, p2; };

__global__ __launch_bounds__(256) void k_undistort_map_matlab(MatlabCam c, int h, int w, float2 *__restrict__ map)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= h * w) return;
    const int i = p / w, j = p - i * w;
    const double u = j + 1, v = i + 1;
    const double y = (v - c.cy) / c.fy, x = ((u - c.cx) - c.skew * y) / c.fx;
    const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r2 * r4;
    const double alpha = (c.k1 * r2 + c.k2 * r4) + c.k3 * r6;
    const double xy = x * y;
    const double dx = 2 * c.p1 * xy + c.p2 * (r2 + 2 * (x * x));
    const double dy = c.p1 * (r2 + 2 * (y * y)) + 2 * c.p2 * xy;
    const double xd = (x + x * alpha) + dx, yd = (y + y * alpha) + dy;
    const double ud = (xd * c.fx + c.cx) + c.skew * yd, vd = yd * c.fy + c.cy;
    map[p] = make_float2((float)(ud - 1.0), (float)(vd - 1.0));
}

__device__ __forceinline__ void keys_weights(float t, float *wt)
{
    const float t2 = t * t, t3 = t2 * t;
    wt[0] = ((-t3 + 2.0f * t2) - t) * 0.5f;
    wt[1] = ((3.0f * t3 - 5.0f * t2) + 2.0f) * 0.5f;
    wt[2] = ((-3.0f * t3 + 4.0f * t2) + t) * 0.5f;
    wt[3] = (t3 - t2) * 0.5f;
}

// one thread per output pixel; the map entry and the 8 weights are made once and applied to REMAP_FRAMES frames.
// The 16 taps are unconditional loads from clamped addresses (independent, one latency); a tap one step outside the image
// is replaced by Keys' extrapolation of the other three.
__global__ __launch_bounds__(256) void k_remap_cubic(const uint8_t *__restrict__ src, int n, int h, int w,
                                                     const float2 *__restrict__ map, int fill, uint8_t *__restrict__ dst)
{
    const int N = h * w;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= N) return;
    const float2 m = map[p];
    const float x = m.x, y = m.y;
    const bool inside = x >= 0.0f && y >= 0.0f && x <= (float)(w - 1) && y <= (float)(h - 1);
    int ix = (int)floorf(x), iy = (int)floorf(y);
    ix = min(max(ix, 0), w - 2); iy = min(max(iy, 0), h - 2);
    float wx[4], wy[4];
    keys_weights(x - (float)ix, wx);
    keys_weights(y - (float)iy, wy);
    const int f0 = blockIdx.y * REMAP_FRAMES, f1 = min(f0 + REMAP_FRAMES, n);
    for (int f = f0; f < f1; f++) {
        int out = fill;
        if (inside) {
            const uint8_t *s0 = src + f * (size_t)N;
            float row[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int yy = min(max(iy - 1 + r, 0), h - 1);
                float s[4];
#pragma unroll
                for (int c = 0; c < 4; c++) s[c] = (float)s0[(size_t)yy * w + min(max(ix - 1 + c, 0), w - 1)];
                if (ix - 1 < 0) s[0] = (3.0f * s[1] - 3.0f * s[2]) + s[3];
                if (ix + 2 >= w) s[3] = (3.0f * s[2] - 3.0f * s[1]) + s[0];
                row[r] = ((s[0] * wx[0] + s[1] * wx[1]) + s[2] * wx[2]) + s[3] * wx[3];
            }
            if (iy - 1 < 0) row[0] = (3.0f * row[1] - 3.0f * row[2]) + row[3];
            if (iy + 2 >= h) row[3] = (3.0f * row[2] - 3.0f * row[1]) + row[0];
            const float v = ((row[0] * wy[0] + row[1] * wy[1]) + row[2] * wy[2]) + row[3] * wy[3];
            out = v <= 0.0f ? 0 : (v >= 255.0f ? 255 : (int)floorf(v + 0.5f));
        }
        dst[f * (size_t)N + p] = (uint8_t)out;
    }
}

}  // namespace

extern "C" int32_t cpe_undistort_map(const double *K, const double *dist, int32_t n_dist, int32_t h, int32_t w,
                                     int16_t *map_xy, uint16_t *map_f, void *stream)
{
    CPE_CHECK_ARG(K && map_xy && map_f && (dist || n_dist == 0), "cpe_undistort_map: null pointer");
    CPE_CHECK_ARG(n_dist == 0 || n_dist == 4 || n_dist == 5 || n_dist == 8 || n_dist == 12,
                  "cpe_undistort_map: %d distortion coefficients (OpenCV takes 4, 5, 8 or 12)", n_dist);
    CPE_CHECK_ARG(h > 0 && w > 0 && h <= 32767 && w <= 32767, "cpe_undistort_map: bad size %dx%d", w, h);
    CamArgs cam;
    for (int i = 0; i < 9; i++) cam.K[i] = K[i];
    for (int i = 0; i < 12; i++) cam.c[i] = i < n_dist ? dist[i] : 0.0;
    {
        // every stripe inverts K with its own principal point; all of them are singular or none
        const double *S = cam.K;
        const double det = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
        CPE_CHECK_ARG(det != 0.0 && S[0] * S[4] != 0.0, "cpe_undistort_map: singular camera matrix");
    }
    int stripe0 = (1 << 12) / (w > 1 ? w : 1);
    if (stripe0 < 1) stripe0 = 1;
    if (stripe0 > h) stripe0 = h;
    CPE_LAUNCH_BEGIN();
    CPE_KLAUNCH(k_undistort_map, dim3((h + 63) / 64), dim3(64), 0, (hipStream_t)stream, cam, h, w, stripe0, map_xy, map_f);
    CPE_CHECK_LAUNCH("k_undistort_map");
    return CPE_OK;
}

extern "C" int32_t cpe_remap_bilinear_batch(const uint8_t *src, int32_t n, int32_t h, int32_t w, const int16_t *map_xy,
                                            const uint16_t *map_f, uint8_t *dst, void *stream)
{
    CPE_CHECK_ARG(src && dst && map_xy && map_f, "cpe_remap_bilinear_batch: null pointer");
    CPE_CHECK_ARG(src != dst, "cpe_remap_bilinear_batch: in-place remap is not possible");
    CPE_CHECK_ARG(n >= 0 && h > 0 && w > 0 && h <= 32767 && w <= 32767, "cpe_remap_bilinear_batch: bad size");
    if (n == 0) return CPE_OK;
    const size_t N = (size_t)h * w;
    const bool vec = (N % 4 == 0) && ((((size_t)dst) | ((size_t)map_xy) | ((size_t)map_f)) % 16 == 0);
    CPE_LAUNCH_BEGIN();
    if (vec)
        CPE_KLAUNCH(k_remap_bilinear4, dim3((unsigned)((N + 1023) / 1024), (n + REMAP_FRAMES - 1) / REMAP_FRAMES), dim3(256), 0,
                    (hipStream_t)stream, src, n, h, w, map_xy, map_f, dst);
    else
        CPE_KLAUNCH(k_remap_bilinear1, dim3((unsigned)((N + 255) / 256), n), dim3(256), 0, (hipStream_t)stream, src, h, w, map_xy, map_f,
                    dst);
    CPE_CHECK_LAUNCH("k_remap_bilinear");
    return CPE_OK;
}

extern "C" int32_t cpe_undistort_map_matlab(const double *K, const double *radial, int32_t n_radial, const double *tangential,
                                            int32_t h, int32_t w, float *map, void *stream)
{
    CPE_CHECK_ARG(K && map && (radial || n_radial == 0), "cpe_undistort_map_matlab: null pointer");
    CPE_CHECK_ARG(n_radial >= 0 && n_radial <= 3, "cpe_undistort_map_matlab: %d radial coefficients (MATLAB takes 2 or 3)", n_radial);
    CPE_CHECK_ARG(h >= 3 && w >= 3 && (long long)h * w < (1ll << 31), "cpe_undistort_map_matlab: bad size %dx%d", w, h);
    CPE_CHECK_ARG(K[0] != 0.0 && K[4] != 0.0, "cpe_undistort_map_matlab: zero focal length");
    MatlabCam c;
    c.fx = K[0]; c.skew = K[1]; c.cx = K[2]; c.fy = K[4]; c.cy = K[5];
    c.k1 = n_radial > 0 ? radial[0] : 0.0; c.k2 = n_radial > 1 ? radial[1] : 0.0; c.k3 = n_radial > 2 ? radial[2] : 0.0;
    c.p1 = tangential ? tangential[0] : 0.0; c.p2 = tangential ? tangential[1] : 0.0;
    CPE_LAUNCH_BEGIN();
    CPE_KLAUNCH(k_undistort_map_matlab, dim3((unsigned)(((size_t)h * w + 255) / 256)), dim3(256), 0, (hipStream_t)stream, c, h, w,
                reinterpret_cast<float2 *>(map));
    CPE_CHECK_LAUNCH("k_undistort_map_matlab");
    return CPE_OK;
}

extern "C" int32_t cpe_remap_cubic_batch(const uint8_t *src, int32_t n, int32_t h, int32_t w, const float *map, int32_t fill,
                                         uint8_t *dst, void *stream)
{
    CPE_CHECK_ARG(src && dst && map, "cpe_remap_cubic_batch: null pointer");
    CPE_CHECK_ARG(src != dst, "cpe_remap_cubic_batch: in-place remap is not possible");
    CPE_CHECK_ARG(n >= 0 && h >= 3 && w >= 3 && (long long)h * w < (1ll << 31), "cpe_remap_cubic_batch: bad size");
    CPE_CHECK_ARG(fill >= 0 && fill <= 255 && (((size_t)map) & 7) == 0, "cpe_remap_cubic_batch: fill value / map alignment");
    if (n == 0) return CPE_OK;
    CPE_LAUNCH_BEGIN();
    CPE_KLAUNCH(k_remap_cubic, dim3((unsigned)(((size_t)h * w + 255) / 256), (n + REMAP_FRAMES - 1) / REMAP_FRAMES), dim3(256), 0,
                (hipStream_t)stream, src, n, h, w, reinterpret_cast<const float2 *>(map), fill, dst);
    CPE_CHECK_LAUNCH("k_remap_cubic");
    return CPE_OK;
}
