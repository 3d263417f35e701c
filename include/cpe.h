/*
 * cpe.h -- C ABI of libcpe_hip.so: the MI355X (gfx950) implementation of the hot path of
 * cv3vpl-lab/cylinder-pose-estimation.
 *
 * The reference has no FFI for this path: its boundary is the Python function
 *     detect_grid(input_img) -> (col_img, result_json, rows_updated, cols_updated)
 *         (python_grid_detection_cylinder.py:68-112, called from MATLAB makePyGridPts.m:29)
 * and the MATLAB function
 *     [pts3, cylT, fvals, meanError] = fitSingleCylinder(...)   (utils/fitSingleCylinder.m:1)
 * Each entry point below names the reference code it replaces.  INTEGRATION.md shows the ctypes
 * binding a maintainer adds to python_grid_detection_cylinder.py.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; every buffer is caller-owned DEVICE memory
 *     (e.g. torch tensors, pass tensor.data_ptr()); no allocation inside, scratch comes from
 *     the caller-supplied workspace; no global state except the thread-local error string.
 *   - every call is asynchronous on `stream` (a hipStream_t, passed as void*; NULL = default
 *     stream) and graph-capturable; no host synchronisation inside.
 *   - return value: 0 ok, <0 argument / launch error (text via cpe_last_error_string()).
 *   - per-frame failures never abort a batch: they are reported in a status[n] array
 *     (CPE_ST_*), mirroring the places where the reference raises inside detect_grid.
 *   - images are row-major u8, frame stride h*w, no padding.
 */
#ifndef CPE_H
#define CPE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CPE_VERSION 100

#if defined(__GNUC__)
#define CPE_API __attribute__((visibility("default")))
#else
#define CPE_API
#endif

#define CPE_OK 0
#define CPE_ERR_ARG (-1)
#define CPE_ERR_LAUNCH (-2)
#define CPE_ERR_WORKSPACE (-3)

/* per-frame status codes */
#define CPE_ST_OK 0
#define CPE_ST_NO_REGION 1   /* no blob contour: cv2.convexHull(None), util_cylinder.py:1894 */
#define CPE_ST_NO_SPOT 2     /* no pixel > 240 after blur19: circle_radius0 unbound, :1974-2007 */
#define CPE_ST_NO_LINES 3    /* no valid rows / cols: indexing_data early returns, :1430-1460 */
#define CPE_ST_EMPTY 4       /* make_json raises on empty point list, :1703-1704 */
#define CPE_ST_FEW_POINTS 5  /* too few 3-D points for the fit (fitCylinderWPts3.m:8, estCurvatures.m:5) */
#define CPE_ST_OVERFLOW 6    /* a fixed capacity of the workspace was exceeded (build-defined) */

CPE_API int32_t cpe_version(void);
CPE_API const char *cpe_last_error_string(void);

/* ------------------------------------------------------------------------------------------
 * Stage a-1: load_and_preprocess_image (util_cylinder.py:1769-1802) for a batch of frames.
 *   gray  u8[n,h,w]  ->  mask u8[n,h,w]  (255 = ridge: Hessian(sigma 3) smaller eigenvalue <= Sauvola
 *   threshold of itself, window 15, k 0.5, R 128; the reference's `binary_img`).
 * One fused kernel: gray tile -> LDS -> 5x5 binomial (integer) -> separable 25-tap Gaussian (f64)
 * -> two central differences -> eigenvalue -> 15x15 box mean / mean-of-squares -> compare.
 * No workspace.  h, w >= 8.
 */
CPE_API int32_t cpe_preprocess_batch(const uint8_t *gray, int32_t n, int32_t h, int32_t w, uint8_t *mask,
                             void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CPE_H */
