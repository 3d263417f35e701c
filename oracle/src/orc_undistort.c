/*
 * ORACLE (test infrastructure) -- row f-3: the undistortion pre-step of the CLI entry point
 *   utils/iotool.py:22-39  undistort_image(image, camera_params)
 *       distortion_coeffs = hstack((RadialDistortion, TangentialDistortion)); cv2.undistort(image, K, coeffs)
 *
 * [ext] cv2.undistort is OpenCV 4.5.5 code that is not in this image: PARITY UNPINNED.  Restated from the published
 * algorithm (modules/calib3d/src/undistort.dispatch.cpp, modules/imgproc/src/imgwarp.cpp), scalar code path:
 *   undistort():  newCameraMatrix = cameraMatrix; the frame is processed in stripes of
 *                 stripe = min(max(1, 4096 / cols), rows) rows; for the stripe that starts at row y0 the principal
 *                 point of the NEW matrix is shifted: Ar(1,2) = cy - y0; initUndistortRectifyMap(A, dist, I, Ar,
 *                 (cols, stripe), CV_16SC2) then remap(INTER_LINEAR, BORDER_CONSTANT = 0).
 *   initUndistortRectifyMap(): iR = inv(Ar * I) (3x3 closed form of cv::invert), per row i of the stripe
 *                 _x = i*ir[1] + ir[2], _y = i*ir[4] + ir[5], _w = i*ir[7] + ir[8], stepped by ir[0], ir[3], ir[6] per
 *                 column (running sums, so the rounding of column j depends on all columns before it);
 *                 x = _x/_w ... distortion polynomial with (k1,k2,p1,p2,k3,k4,k5,k6,s1..s4), no tilt;
 *                 fixed point: iu = cvRound(u*32), map1 = (iu>>5, iv>>5) as int16, map2 = (iv&31)*32 + (iu&31).
 *   remap(INTER_LINEAR) on u8: weights (32-fx)(32-fy)*32 ... (BilinearTab_i, exact: they always sum to 2^15),
 *                 D = (sum_k w_k S_k + 2^14) >> 15, constant border 0 for neighbours outside the source.
 * The coefficient vector is used the way OpenCV reads it -- (k1, k2, p1, p2[, k3[, k4, k5, k6[, s1..s4]]]) -- so a
 * three-term MATLAB RadialDistortion stacked in front of the tangential pair is misread exactly as the reference does.
 */
#include "orc_common.h"

static int orc_cvround(double v) { return (int)lrint(v); }   /* round half to even (default rounding mode) */

static int orc_sat_int(double v)
{
    /* saturate_cast<int>(double) = cvRound; out-of-range doubles are undefined in C: clamp them */
    if (v >= 2147483647.0) return 2147483647;
    if (v <= -2147483648.0) return (-2147483647 - 1);
    return orc_cvround(v);
}

static int orc_invert3(const double *S, double *t)
{
    double d = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
    if (d == 0.) return 0;
    d = 1. / d;
    t[0] = (S[4] * S[8] - S[5] * S[7]) * d;
    t[1] = (S[2] * S[7] - S[1] * S[8]) * d;
    t[2] = (S[1] * S[5] - S[2] * S[4]) * d;
    t[3] = (S[5] * S[6] - S[3] * S[8]) * d;
    t[4] = (S[0] * S[8] - S[2] * S[6]) * d;
    t[5] = (S[2] * S[3] - S[0] * S[5]) * d;
    t[6] = (S[3] * S[7] - S[4] * S[6]) * d;
    t[7] = (S[1] * S[6] - S[0] * S[7]) * d;
    t[8] = (S[0] * S[4] - S[1] * S[3]) * d;
    return 1;
}

/* K: 3x3 row major camera matrix; dist: nd in {0,4,5,8,12} coefficients in OpenCV order.
 * map_xy: h*w*2 int16 (x, y), map_f: h*w uint16.  Returns 0, or -1 for a singular matrix / bad nd. */
ORC_API int orc_undistort_map(const double *K, const double *dist, int nd, int h, int w, int16_t *map_xy, uint16_t *map_f)
{
    if (!(nd == 0 || nd == 4 || nd == 5 || nd == 8 || nd == 12)) return -1;
    double c[12] = {0};
    for (int i = 0; i < nd; i++) c[i] = dist[i];
    const double k1 = c[0], k2 = c[1], p1 = c[2], p2 = c[3], k3 = c[4], k4 = c[5], k5 = c[6], k6 = c[7];
    const double s1 = c[8], s2 = c[9], s3 = c[10], s4 = c[11];
    const double fx = K[0], fy = K[4], u0 = K[2], v0 = K[5];
    int stripe0 = (1 << 12) / (w > 1 ? w : 1);
    if (stripe0 < 1) stripe0 = 1;
    if (stripe0 > h) stripe0 = h;
    for (int y0 = 0; y0 < h; y0 += stripe0) {
        int stripe = stripe0 < h - y0 ? stripe0 : h - y0;
        double Ar[9], ir[9];
        memcpy(Ar, K, sizeof(Ar));
        Ar[5] = v0 - y0;
        if (!orc_invert3(Ar, ir)) return -1;
        for (int i = 0; i < stripe; i++) {
            int16_t *m1 = map_xy + (size_t)(y0 + i) * w * 2;
            uint16_t *m2 = map_f + (size_t)(y0 + i) * w;
            double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
            for (int j = 0; j < w; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
                double ww = 1. / _w, x = _x * ww, y = _y * ww;
                double x2 = x * x, y2 = y * y;
                double r2 = x2 + y2, _2xy = 2 * x * y;
                double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
                double xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2) + s1 * r2 + s2 * r2 * r2);
                double yd = (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy + s3 * r2 + s4 * r2 * r2);
                /* no tilt: vecTilt = (xd, yd, 1), invProj = 1 */
                double invProj = 1.;
                double u = fx * invProj * xd + u0;
                double v = fy * invProj * yd + v0;
                int iu = orc_sat_int(u * 32), iv = orc_sat_int(v * 32);
                m1[j * 2] = (int16_t)(iu >> 5);
                m1[j * 2 + 1] = (int16_t)(iv >> 5);
                m2[j] = (uint16_t)((iv & 31) * 32 + (iu & 31));
            }
        }
    }
    return 0;
}

ORC_API void orc_remap_bilinear(const uint8_t *src, int h, int w, const int16_t *map_xy, const uint16_t *map_f, uint8_t *dst)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            size_t i = (size_t)y * w + x;
            int sx = map_xy[2 * i], sy = map_xy[2 * i + 1];
            int f = map_f[i] & 1023, fx = f & 31, fy = f >> 5;
            int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
            int v00 = 0, v01 = 0, v10 = 0, v11 = 0;
            if (sy >= 0 && sy < h) {
                if (sx >= 0 && sx < w) v00 = src[(size_t)sy * w + sx];
                if (sx + 1 >= 0 && sx + 1 < w) v01 = src[(size_t)sy * w + sx + 1];
            }
            if (sy + 1 >= 0 && sy + 1 < h) {
                if (sx >= 0 && sx < w) v10 = src[(size_t)(sy + 1) * w + sx];
                if (sx + 1 >= 0 && sx + 1 < w) v11 = src[(size_t)(sy + 1) * w + sx + 1];
            }
            int s = v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11;
            int d = (s + (1 << 14)) >> 15;
            dst[i] = (uint8_t)(d < 0 ? 0 : (d > 255 ? 255 : d));
        }
}

ORC_API int orc_undistort(const uint8_t *src, int h, int w, const double *K, const double *dist, int nd, uint8_t *dst)
{
    int16_t *mxy = (int16_t *)malloc((size_t)h * w * 2 * sizeof(int16_t));
    uint16_t *mf = (uint16_t *)malloc((size_t)h * w * sizeof(uint16_t));
    int rc = orc_undistort_map(K, dist, nd, h, w, mxy, mf);
    if (rc == 0) orc_remap_bilinear(src, h, w, mxy, mf, dst);
    free(mxy); free(mf);
    return rc;
}

/* ------------------------------------------------------------------------------------------------------------------
 * Second mode of row f-3: the MATLAB entry point undistorts with
 *     undistortImage(I, cameraParams, 'cubic')            utils/preProcessing.m:3-4 (and :15)
 * [ext] Computer Vision Toolbox code that is not in this image: PARITY UNPINNED.  Restated from the documented model:
 *   - output view 'same', fill value 0: output pixel (u, v) (MATLAB pixel coordinates, 1-based) looks up the source at
 *     distortPoints([u v]):  y = (v - cy) / fy;  x = (u - cx - skew * y) / fx;  r2 = x^2 + y^2;
 *     alpha = k1 r2 + k2 r2^2 + k3 r2^3;  dx = 2 p1 x y + p2 (r2 + 2 x^2);  dy = p1 (r2 + 2 y^2) + 2 p2 x y;
 *     xd = x + x alpha + dx;  yd = y + y alpha + dy;  ud = xd fx + cx + skew yd;  vd = yd fy + cy        (f64)
 *   - interp2d(I, X, Y, 'cubic', 0) on a uint8 image works in single precision: cubic convolution (Keys, a = -1/2),
 *     separable, the sample one step outside the image extrapolated as 3 f0 - 3 f1 + f2 (Keys' boundary condition, what
 *     interp2 'cubic' pads with), points outside [1, W] x [1, H] take the fill value, the result is cast back to uint8
 *     (round half away from zero, saturate).
 * K is the 3x3 matrix of the camera JSON (createCameraDataJSON.m:7: IntrinsicMatrix' = [fx s cx; 0 fy cy; 0 0 1], MATLAB's
 * 1-based principal point).  The map holds 0-based source coordinates as float32 (x, y). */
ORC_API void orc_undistort_map_matlab(const double *K, const double *radial, int nr, const double *tang, int h, int w, float *map)
{
    const double fx = K[0], skew = K[1], cx = K[2], fy = K[4], cy = K[5];
    const double k1 = nr > 0 ? radial[0] : 0, k2 = nr > 1 ? radial[1] : 0, k3 = nr > 2 ? radial[2] : 0;
    const double p1 = tang ? tang[0] : 0, p2 = tang ? tang[1] : 0;
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            const double u = j + 1, v = i + 1;
            const double y = (v - cy) / fy, x = ((u - cx) - skew * y) / fx;
            const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r2 * r4;
            const double alpha = (k1 * r2 + k2 * r4) + k3 * r6;
            const double xy = x * y;
            const double dx = 2 * p1 * xy + p2 * (r2 + 2 * (x * x));
            const double dy = p1 * (r2 + 2 * (y * y)) + 2 * p2 * xy;
            const double xd = (x + x * alpha) + dx, yd = (y + y * alpha) + dy;
            const double ud = (xd * fx + cx) + skew * yd, vd = yd * fy + cy;
            map[((size_t)i * w + j) * 2] = (float)(ud - 1.0);
            map[((size_t)i * w + j) * 2 + 1] = (float)(vd - 1.0);
        }
}

static void orc_keys_weights(float t, float *wt)
{
    const float t2 = t * t, t3 = t2 * t;
    wt[0] = ((-t3 + 2.0f * t2) - t) * 0.5f;
    wt[1] = ((3.0f * t3 - 5.0f * t2) + 2.0f) * 0.5f;
    wt[2] = ((-3.0f * t3 + 4.0f * t2) + t) * 0.5f;
    wt[3] = (t3 - t2) * 0.5f;
}

ORC_API void orc_remap_cubic(const uint8_t *src, int h, int w, const float *map, int fill, uint8_t *dst)
{
    for (size_t p = 0; p < (size_t)h * w; p++) {
        const float x = map[2 * p], y = map[2 * p + 1];
        int out = fill;
        if (h >= 3 && w >= 3 && x >= 0.0f && y >= 0.0f && x <= (float)(w - 1) && y <= (float)(h - 1)) {
            int ix = (int)floorf(x), iy = (int)floorf(y);
            if (ix > w - 2) ix = w - 2;
            if (iy > h - 2) iy = h - 2;
            float wx[4], wy[4], row[4];
            orc_keys_weights(x - (float)ix, wx);
            orc_keys_weights(y - (float)iy, wy);
            for (int r = 0; r < 4; r++) {
                const int yy = iy - 1 + r;
                if (yy < 0 || yy >= h) { row[r] = 0; continue; }
                float s[4];
                for (int c = 0; c < 4; c++) {
                    const int xx = ix - 1 + c;
                    s[c] = (xx >= 0 && xx < w) ? (float)src[(size_t)yy * w + xx] : 0.0f;
                }
                if (ix - 1 < 0) s[0] = (3.0f * s[1] - 3.0f * s[2]) + s[3];
                if (ix + 2 >= w) s[3] = (3.0f * s[2] - 3.0f * s[1]) + s[0];
                row[r] = ((s[0] * wx[0] + s[1] * wx[1]) + s[2] * wx[2]) + s[3] * wx[3];
            }
            if (iy - 1 < 0) row[0] = (3.0f * row[1] - 3.0f * row[2]) + row[3];
            if (iy + 2 >= h) row[3] = (3.0f * row[2] - 3.0f * row[1]) + row[0];
            const float v = ((row[0] * wy[0] + row[1] * wy[1]) + row[2] * wy[2]) + row[3] * wy[3];
            out = v <= 0.0f ? 0 : (v >= 255.0f ? 255 : (int)floorf(v + 0.5f));
        }
        dst[p] = (uint8_t)out;
    }
}
