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
 *     the caller-supplied workspace.  Process state: the thread-local error string, the optional profiling timers,
 *     and sets of three helper streams with their events, one set per (device, caller stream) for up to four caller
 *     streams per device (further ones share the last set): cpe_detect_grid_batch* overlaps the independent chains of a
 *     call on them, so calls on different caller streams do not serialise each other; the enqueue of a call (host side,
 *     fork to join) holds its set's mutex, so the library is thread-safe; CPE_SERIAL=1 in the environment keeps
 *     everything on the caller's stream.
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

#define CPE_VERSION 103   /* 100 + round: exports are only ever added (103: cpe_detect_grid_bgr_batch_ex, cpe_detect_constants) */

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
#define CPE_ST_SUBPIXEL_RAISED 7 /* optional sub-pixel stage: a line sample leaves the image through the top / left edge;
                                   compute_center_of_gravity_x/y raise there (util_cylinder.py:722,741,769,786) */

CPE_API int32_t cpe_version(void);
CPE_API const char *cpe_last_error_string(void);

/* Per-kernel hipEvent timers (the build's stand-in for the reference's commented line_profiler hooks,
 * util_cylinder.py:2010-2011).  Off by default; while on, every internal launch is bracketed by events on
 * its stream.  cpe_profile_report synchronises them, writes "kernel,calls,total_ms" lines (descending)
 * into csv and clears the records; returns the number of distinct kernels. */
CPE_API void cpe_profile_enable(int32_t on);
CPE_API int32_t cpe_profile_report(char *csv, size_t cap);

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

/* ------------------------------------------------------------------------------------------
 * detect_grid(input_img) for a batch of n grey frames (python_grid_detection_cylinder.py:68-112 ->
 * util_cylinder.py stages 1-6): gray u8[n,h,w] (the 2-D array makePyGridPts.m:26 passes; a BGR frame
 * with three identical channels is the same image) ->
 *   xy f64[n,CPE_MAXP,2], id i32[n,CPE_MAXP,2] = (col,row), n_pts i32[n], center f64[n,2]:
 *   exactly the content make_json serialises (util_cylinder.py:1674-1727): points with col >= 0 sorted
 *   by (col,row), and the centre point;
 *   status i32[n]: CPE_ST_* (the places where the reference raises inside detect_grid and returns None).
 * ws: cpe_detect_workspace_bytes(n,h,w) bytes of 256-byte aligned device scratch.  64 <= h,w <= 4096.
 */
CPE_API size_t cpe_detect_workspace_bytes(int32_t n, int32_t h, int32_t w);
CPE_API int32_t cpe_detect_grid_batch(const uint8_t *gray, int32_t n, int32_t h, int32_t w, void *ws, size_t ws_bytes,
                                      double *xy, int32_t *id, int32_t *n_pts, double *center, int32_t *status,
                                      void *stream);

/* The same with options.  subpixel != 0 inserts the reference's (disabled) grey-level centre-of-gravity refinement
 * of the fitted lines -- modify_grayscale_Cline(img, rows, cols, degree=2, sample_step, window_size), whose call is
 * commented out at util_cylinder.py:2040 -- between remove_label and the intersection step.  params NULL = defaults
 * = the live reference path (subpixel 0). */
typedef struct CpeDetectParams {
    int32_t subpixel;        /* 0 = off (reference behaviour) */
    int32_t subpixel_window; /* window_size (the commented call uses 7) */
    double subpixel_step;    /* sample_step (1.0) */
    int32_t target;          /* CPE_TARGET_CYLINDER (python_grid_detection_cylinder.py) or CPE_TARGET_PLANE (row f-2:
                                python_grid_detection_plane.py over utils/util_plane.py; point ids are (row, col) there,
                                every column is kept, sub-pixel refinement is not available) */
    int32_t reserved;
} CpeDetectParams;
#define CPE_TARGET_CYLINDER 0
#define CPE_TARGET_PLANE 1
CPE_API int32_t cpe_detect_grid_batch_ex(const uint8_t *gray, int32_t n, int32_t h, int32_t w, const CpeDetectParams *params,
                                         void *ws, size_t ws_bytes, double *xy, int32_t *id, int32_t *n_pts,
                                         double *center, int32_t *status, void *stream);

/* The reference's inline constants, as this build was compiled with them ("Config": the reference has no
 * configuration object -- every value below is a literal in util_cylinder.py / util_plane.py or an OpenCV default).  They are
 * compile-time constants of the kernels (window sizes decide register windows and LDS rings), so this is a REPORT, not a
 * setter: a maintainer who changes a literal in the reference finds here what to change in the kernels, and the tests pin the
 * table to the reference's values.  target: CPE_TARGET_CYLINDER or CPE_TARGET_PLANE. */
typedef struct CpeDetectConstants {
    int32_t blur_ksize;            /* cv2.GaussianBlur(gray, (5,5), 0)                         util_cylinder.py:1790 */
    double hessian_sigma;          /* detect_ridges(blurred, sigma=3.0)                        :1793 */
    int32_t sauvola_window;        /* sauvola_threshold_fast(b, window_size=15, k=0.5, R=128)  :1797 */
    double sauvola_k, sauvola_R;
    int32_t open_len;              /* MORPH_RECT (20,1) / (1,20) openings                      :1810-1811 */
    double clahe_clip;             /* detect_largest_blob(..., clipLimit=4.5)                  python_grid_detection_cylinder.py:88 */
    int32_t clahe_tiles;           /* tileGridSize=(4,4) (the colour branch: the image is always 3-channel here) :1843 */
    int32_t blob_thr_min, blob_thr_step, blob_thr_count;   /* SimpleBlobDetector defaults 50 .. 220 step 10: 17 binarisations */
    double blob_min_area, blob_max_area;                   /* params.minArea = 10 (:1858), default maxArea 5000 */
    double blob_min_dist;          /* default minDistBetweenBlobs 10 */
    int32_t blob_min_repeat;       /* default minRepeatability 2 */
    int32_t disc_extra_radius;     /* int(size / 2 + 4)                                        :1876 */
    int32_t spot_blur_ksize;       /* cv2.GaussianBlur(gray, (19,19), 0)                       :1962 */
    int32_t spot_threshold;        /* cv2.threshold(blurred, 240, ...)                         :1965 */
    int32_t spot_small_radius, spot_small_add, spot_large_add;   /* radius < 30: + 20, else + 5   :1981-1984 */
    int32_t frag_patch, frag_min_pixels, frag_max_pixels;        /* expand_line_roi(patch 15, 5 .. 200 pixels) :137 (plane: 8 .. 700) */
    int32_t frag_kernel_base;      /* kernel_size = 91 + circle_radius0 (:2022); plane: the fixed 201 (util_plane.py:2806) */
    int32_t index_blur_ksize;      /* cv2.GaussianBlur(img, (7,7), 0) of indexing_data         :1433 */
    int32_t poly_degree;           /* fit_and_draw_polynomial: 2 (cylinder), 1 (plane) */
    int32_t plane_threshold, plane_dilate_ksize;   /* get_convex_hull: threshold 127, MORPH_ELLIPSE (11,11) (util_plane.py:2590-2689); 0 for the cylinder */
    int32_t max_points, max_lines, max_joints, max_groups_per_dir, max_joints_per_group;   /* build capacities (CPE_ST_OVERFLOW) */
} CpeDetectConstants;
CPE_API int32_t cpe_detect_constants(int32_t target, CpeDetectConstants *out);

/* rows_updated / cols_updated of ONE frame of the last cpe_detect_grid_batch call on this workspace: the third and fourth
 * return values of detect_grid (python_grid_detection_cylinder.py:110; built by find_and_assign_intersections_P and
 * clean_and_relabel, util_cylinder.py:1106-1206).  Side 0 = rows ("row1".."rowR", ordered by mean y), side 1 = columns
 * ("col1".."colC", ordered by mean x; negative columns included -- remove_minus_labels only prunes the JSON).
 *   eq      f64[2, CPE_MAXL, 6]            [a2, a1, a0, min-50, max+50, span] of the quadratic fit (util_cylinder.py:473-550)
 *   npts    i32[2, CPE_MAXL]               intersections on the line
 *   pts     f64[2, CPE_MAXL, CPE_MAXL, 2]  (x, y) in the reference's loop order
 *   n_lines i32[2]                         R, C  (0, 0 for a frame that failed before this stage)
 * All device buffers; asynchronous on `stream`. */
#define CPE_MAXL 256
/* Capacities of the line stage, part of the boundary: the reference's lists are unbounded (util_cylinder.py:376-389,
 * 1106-1151); a frame with more joints inside the region rectangle than CPE_MAXJ, more label groups per direction than
 * CPE_MAXL, more joints in one group than CPE_MAXLP or more grid points than CPE_MAXP ends with CPE_ST_OVERFLOW -- in this
 * library AND in the test oracle (oracle/src/orc_lines.c reads these constants), so the two agree on every frame.
 * Seen: 8 000 joints / 5 500 inside the rectangle / 250 groups / 255 joints in a group on 1920x1200 frames degraded with
 * +-9 DN of noise and an intensity ramp (tools/stress_parity.py seeds 4011, 9508); clean frames: < 1 500 / < 64 / < 64. */
#define CPE_MAXJ 16384
#define CPE_MAXLP 1024
CPE_API int32_t cpe_detect_line_tables(const void *ws, size_t ws_bytes, int32_t n, int32_t h, int32_t w, int32_t frame,
                                       double *eq, int32_t *npts, double *pts, int32_t *n_lines, void *stream);

/* Colour input.  The reference's CLI hands detect_grid the H x W x 3 BGR array of cv2.imread (python_grid_detection_cylinder.py:34-44);
 * load_and_preprocess_image (util_cylinder.py:1781-1789) and mask_roi_around_center (:1957) work on cv2.cvtColor(BGR2GRAY) of it:
 * 8-bit fixed point, gray = (B*3735 + G*19235 + R*9798 + 2^14) >> 15 ([ext] OpenCV 4.5.5; identity on grey-replicated frames);
 * detect_largest_blob takes the L channel of cv2.cvtColor(BGR2LAB) of the COLOUR image (:1840) and indexing_data blurs the colour
 * image 7x7 channel by channel before converting it (:1433-1435).
 *   cpe_bgr2gray_batch             bgr u8[n,h,w,3] interleaved -> gray u8[n,h,w]; both 4-byte aligned device buffers.
 *   cpe_detect_grid_bgr_batch_ex   the whole of detect_grid on true-colour frames: same outputs, workspace and status codes as
 *                                  cpe_detect_grid_batch_ex (cylinder target, no sub-pixel refinement); on grey-replicated
 *                                  frames it returns what cpe_detect_grid_batch returns for the grey plane. */
CPE_API int32_t cpe_detect_grid_bgr_batch_ex(const uint8_t *bgr, int32_t n, int32_t h, int32_t w, const CpeDetectParams *params,
                                             void *ws, size_t ws_bytes, double *xy, int32_t *id, int32_t *n_pts,
                                             double *center, int32_t *status, void *stream);
CPE_API int32_t cpe_bgr2gray_batch(const uint8_t *bgr, int32_t n, int32_t h, int32_t w, uint8_t *gray, void *stream);

/* Where an intermediate of the last cpe_detect_grid_batch call lives inside the workspace (for
 * stage-by-stage parity tests and debugging): plane-major, frame f at offset + f * bytes_per_frame. */
#define CPE_PLANE_BINARY 0        /* u8[h,w]  load_and_preprocess_image -> binary_img */
#define CPE_PLANE_HMASK 1         /* u8[h,w]  extract_joints -> horizontal_mask */
#define CPE_PLANE_VMASK 2         /* u8[h,w]  extract_joints -> vertical_mask */
#define CPE_PLANE_MASK_CONTOUR 3  /* u8[h,w]  detect_largest_blob -> mask_contour */
#define CPE_PLANE_ROI_H 4         /* u8[h,w]  mask_roi_around_center -> mask_roi_h */
#define CPE_PLANE_ROI_V 5
#define CPE_PLANE_EXP_H 6         /* u8[h,w]  expands_line_roi -> horizontal_expanded */
#define CPE_PLANE_EXP_V 7
#define CPE_PLANE_JOINTS 8        /* i32[CPE_MAXJ,2] cylinder_centroids in contour order */
#define CPE_PLANE_STATE 9         /* per-frame state record (see csrc/cpe_dev.h FrameState) */
#define CPE_PLANE_CLAHE 10        /* u8[h,w]  CLAHE'd L channel */
#define CPE_PLANE_BLUR19 11       /* u8[h,w] */
#define CPE_PLANE_BLUR7 12        /* u8[h,w] */
#define CPE_PLANE_SWEEP 14        /* i32[192] blob-sweep counters: [8+k] dark components, [25+k] bright components,
                                     [42+k] blobs of threshold 50+10k (k < 17); see csrc/region.hip SW_* */
CPE_API int32_t cpe_detect_workspace_plane(int32_t n, int32_t h, int32_t w, int32_t plane, size_t *offset,
                                           size_t *bytes_per_frame);

/* One stand-alone connected-component pass (cv2.connectedComponents / the front end of cv2.findContours,
 * util_cylinder.py:28,161,1817,1883,1968) over img u8[n,h,w]: set = (img > thr) != invert, 8- or 4-connected;
 * labels (raster index of the component's first pixel, -1 outside the set) land in workspace plane
 * CPE_PLANE_LABELS.  count_mode / want_bbox / want_roots switch the optional by-products (profiling aid). */
#define CPE_PLANE_LABELS 13       /* i32[h,w] */
CPE_API int32_t cpe_debug_ccl(const uint8_t *img, int32_t n, int32_t h, int32_t w, int32_t thr, int32_t invert,
                              int32_t conn8, int32_t count_mode, int32_t want_bbox, int32_t want_roots, void *ws,
                              size_t ws_bytes, void *stream);

/* The RETR_EXTERNAL rule of cv2.findContours as the detector applies it (util_cylinder.py:161, 1817; the other two call
 * sites, :1883 and :1968, keep only the largest contour, which is never a nested one): of the 8-connected components of
 * mask != 0, the raster-first pixels (y * w + x, any order) of those that do not lie inside a hole of another component.
 * first_px i32[n,cap], count i32[n] (may exceed cap: then only cap entries were stored).  Test / debugging aid. */
CPE_API int32_t cpe_debug_external_components(const uint8_t *mask, int32_t n, int32_t h, int32_t w, void *ws, size_t ws_bytes,
                                              int32_t *first_px, int32_t cap, int32_t *count, void *stream);

/* ------------------------------------------------------------------------------------------
 * Grid-point tables.  One table per image: xy f64[n,CPE_MAXP,2] pixel coordinates, id i32[n,CPE_MAXP,2]
 * (col,row) grid indices, cnt i32[n] -- the padded form of the reference's N x 4 matrix
 * [x y colIdx rowIdx] (makePyGridPts.m:39-41, pointsStruct2mat.m:16).
 */
#define CPE_MAXP 2048          /* capacity of one grid-point table (a 3840x2160 frame holds ~1000-1400 points) */
#define CPE_FIT_TABLE_DIM 128  /* (col,row) indices of one frame must span < 128 in each direction */

#define CPE_FIT_FLAG_FALLBACK 1 /* selector found nothing -> plain index join (chooseIdx.m:101-104) */
#define CPE_FIT_FLAG_OVERFLOW 2 /* index span exceeds CPE_FIT_TABLE_DIM or |index| > 9999: frame skipped */

#define CPE_SEL_CHOOSE_IDX 0     /* chooseIdx(gp1,gp2,.,.,patch,th)          fitSingleCylinder.m:12 (live) */
#define CPE_SEL_THRESHOLD 1      /* triangulateWithThreshold(gp1,gp2,.,.,th) fitSingleCylinder.m:11 */
#define CPE_SEL_JOIN 2           /* findGridCorrespondences(gp1,gp2)         fitSingleCylinder.m:10 */

CPE_API size_t cpe_fit_workspace_bytes(int32_t n);

/* Index matching + triangulation for n stereo frames (one wavefront per frame).
 * Replaces chooseIdx.m / triangulateWithThreshold.m / findGridCorrespondences.m followed by
 * triangulate(cgp1, cgp2, stereoParams) at fitSingleCylinder.m:12-17.
 *   K1, K2   f64[9]  row-major intrinsics (getCamParams.m:6-7), T21 f64[16] row-major T_C2_C1 (:9)
 *   outputs  p1,p2 f64[n,CPE_MAXP,2] selected pixel pairs; idx i32[n,CPE_MAXP,2]; X f64[n,CPE_MAXP,3]
 *            points in the camera-1 frame; err f64[n,CPE_MAXP] per-point reprojection error;
 *            m i32[n] number of selected points; mean_err f64[n]; flags i32[n] (CPE_FIT_FLAG_*)
 *   ws       cpe_fit_workspace_bytes(n) bytes of device scratch
 */
CPE_API int32_t cpe_select_triangulate_batch(const double *xy1, const int32_t *id1, const int32_t *cnt1,
                                             const double *xy2, const int32_t *id2, const int32_t *cnt2, int32_t n,
                                             const double *K1, const double *K2, const double *T21, int32_t selector,
                                             int32_t patch, double th, void *ws, size_t ws_bytes, double *p1,
                                             double *p2, int32_t *idx, double *X, double *err, int32_t *m,
                                             double *mean_err, int32_t *flags, void *stream);

/* The two halves of the above on their own (SURVEY 8b lists them as separate entry points).
 * cpe_choose_idx_batch: [cgp1, cgp2] = chooseIdx(gp1, gp2, imgInfo, stereoParams, patch, th) (chooseIdx.m:1-105; call at
 *   fitSingleCylinder.m:12 with patch 3, th 0.3), fallback to findGridCorrespondences included (flags).  Outputs p1, p2, idx,
 *   m, flags as above; ws as above.
 * cpe_triangulate_batch: [worldPoints, reprojectionErrors] = triangulate(cgp1, cgp2, stereoParams) for pairs that are already
 *   matched, and meanError = mean(reprojectionErrors) (fitSingleCylinder.m:15-17).  p1, p2 f64[n,CPE_MAXP,2], cnt i32[n] ->
 *   X f64[n,CPE_MAXP,3] (camera-1 frame), err f64[n,CPE_MAXP], mean_err f64[n]. */
CPE_API int32_t cpe_choose_idx_batch(const double *xy1, const int32_t *id1, const int32_t *cnt1, const double *xy2,
                                     const int32_t *id2, const int32_t *cnt2, int32_t n, const double *K1, const double *K2,
                                     const double *T21, int32_t patch, double th, void *ws, size_t ws_bytes, double *p1,
                                     double *p2, int32_t *idx, int32_t *m, int32_t *flags, void *stream);
CPE_API int32_t cpe_triangulate_batch(const double *p1, const double *p2, const int32_t *cnt, int32_t n, const double *K1,
                                      const double *K2, const double *T21, double *X, double *err, double *mean_err,
                                      void *stream);

typedef struct CpeFitParams {
    double tol_x;          /* fminsearch TolX  (fitCylinderWPts3.m:33: 1e-5) */
    double tol_f;          /* fminsearch TolFun (1e-5) */
    int32_t max_iter;      /* MaxIter (1e5) */
    int32_t max_fun_evals; /* MaxFunEvals (1e5) */
    int32_t mode;          /* CPE_FIT_NELDER_MEAD (reference behaviour, default) or CPE_FIT_LM */
    int32_t reserved;      /* 0 */
} CpeFitParams;
#define CPE_FIT_NELDER_MEAD 0 /* fminsearch clone: what fitCylinderWPts3.m:38 runs */
#define CPE_FIT_LM 1          /* Levenberg-Marquardt on the same objective (fast mode; not in the reference) */

/* fitCylinderWPts3(pts3, radius) + applyCylParamsPrior + cylParams2T for n frames, one wavefront per
 * frame (fitSingleCylinder.m:20-25).  X f64[n,CPE_MAXP,3], cnt i32[n].  params NULL = reference values.
 *   cyl_raw f64[n,2,6]  [cylParams0; cylParams] as returned by fitCylinderWPts3
 *   cyl     f64[n,2,6]  the same after applyCylParamsPrior
 *   T       f64[n,16]   row-major cylT = cylParams2T(cyl(2,:))
 *   fvals   f64[n,2]    [f0, f]        iters i32[n,2] = [iterations, function evaluations]
 *   status  i32[n]      CPE_ST_OK or CPE_ST_FEW_POINTS
 */
CPE_API int32_t cpe_fit_cylinder_batch(const double *X, const int32_t *cnt, int32_t n, double radius,
                                       const CpeFitParams *params, double *cyl_raw, double *cyl, double *T,
                                       double *fvals, int32_t *iters, int32_t *status, void *stream);

/* BUILD-DEFINED extension (BASELINE.json configs[4] "RANSAC-wrapped fitSingleCylinder"; the reference has no RANSAC):
 * per frame, `hypotheses` LM fits (`hyp_iters` iterations each) on random subsets -- hypothesis 0 = all points, the others
 * keep a point with probability sample / count, decided by a counter-based hash of (seed, frame0 + frame, hypothesis, point)
 * -- scored by the number of points with |dist(point, axis) - radius| < tau; the final fit (`params->mode`) runs on the
 * inliers of the best hypothesis.  Outputs as cpe_fit_cylinder_batch (cyl_raw row 0 = the all-points initial cylinder,
 * fvals[0] = objective there, fvals[1] = objective of the final fit over the inliers) plus
 *   n_inliers i32[n], inlier_mask u8[n,CPE_MAXP] (1 = used by the final fit). */
typedef struct {
    int32_t hypotheses;   /* default 64 */
    int32_t sample;       /* expected subset size, default 12 (>= 6) */
    double tau;           /* inlier band around the radius, same unit as the points, default 0.5 */
    uint64_t seed;
    uint64_t frame0;      /* global index of frame 0 of this call (shards of one batch get different streams) */
    int32_t hyp_iters;    /* LM iterations per hypothesis, default 8 */
    int32_t reserved;
} CpeRansacParams;
CPE_API int32_t cpe_fit_cylinder_ransac_batch(const double *X, const int32_t *cnt, int32_t n, double radius,
                                              const CpeFitParams *params, const CpeRansacParams *ransac, double *cyl_raw,
                                              double *cyl, double *T, double *fvals, int32_t *iters, int32_t *status,
                                              int32_t *n_inliers, uint8_t *inlier_mask, void *stream);

/* Row f-3: the undistortion pre-step of the CLI entry point, utils/iotool.py:22-39
 *   undistort_image(image, camera_params) = cv2.undistort(image, IntrinsicMatrix, hstack(Radial, Tangential))
 * cv2.undistort rebuilds its fixed-point map per image; here the map is built once per camera and applied per frame.
 *   cpe_undistort_map: K f64[9] row-major and dist f64[n_dist] are HOST pointers (n_dist in {0,4,5,8,12}, OpenCV order
 *     k1 k2 p1 p2 [k3 [k4 k5 k6 [s1 s2 s3 s4]]]); map_xy i16[h,w,2] and map_f u16[h,w] are device buffers in OpenCV's
 *     CV_16SC2 / CV_16UC1 convention (integer source pixel; 5-bit fractions fy*32 + fx).
 *   cpe_remap_bilinear_batch: dst[n,h,w] = remap(src[n,h,w], map, INTER_LINEAR, BORDER_CONSTANT 0); src != dst. */
CPE_API int32_t cpe_undistort_map(const double *K, const double *dist, int32_t n_dist, int32_t h, int32_t w,
                                  int16_t *map_xy, uint16_t *map_f, void *stream);
CPE_API int32_t cpe_remap_bilinear_batch(const uint8_t *src, int32_t n, int32_t h, int32_t w, const int16_t *map_xy,
                                         const uint16_t *map_f, uint8_t *dst, void *stream);

/* Row f-3, second mode: the MATLAB entry point undistorts with undistortImage(I, cameraParams, 'cubic')
 * (utils/preProcessing.m:3-4, :15).  Output view 'same', fill value `fill`.
 *   cpe_undistort_map_matlab: K f64[9] row-major as the camera JSON holds it (createCameraDataJSON.m:7: [fx s cx; 0 fy cy;
 *     0 0 1] with MATLAB's 1-based principal point), radial f64[n_radial] (n_radial 0..3: k1 k2 [k3]), tangential f64[2] or
 *     NULL -- HOST pointers; map f32[h,w,2] device buffer: 0-based source (x, y) of every output pixel (distortPoints in f64).
 *   cpe_remap_cubic_batch: dst[n,h,w] = interp2d(src[n,h,w], map, 'cubic', fill): cubic convolution (a = -1/2) in single
 *     precision, result rounded half away from zero and saturated; src != dst; h, w >= 3. */
CPE_API int32_t cpe_undistort_map_matlab(const double *K, const double *radial, int32_t n_radial, const double *tangential,
                                         int32_t h, int32_t w, float *map, void *stream);
CPE_API int32_t cpe_remap_cubic_batch(const uint8_t *src, int32_t n, int32_t h, int32_t w, const float *map, int32_t fill,
                                      uint8_t *dst, void *stream);

/* Row f-1: the per-frame terms of the multi-frame objective of fitCylinderWPts3sAngs.m:82-94 (`dist`):
 * terms[i] = mean((getDistPts3ToLine(Pts3s{i}, line(T * TAGVcyls{i})) - radius)^2), one wavefront per frame.
 * X f64[n,CPE_MAXP,3], cnt i32[n], TAGVcyl f64[n,16] (row-major getTAGVcyl(pan,tilt)), T f64[16] (device, row-major
 * vec2T(agvPose)).  The 6-parameter Nelder-Mead around it runs on the host (cpe_amd/multiframe.py). */
CPE_API int32_t cpe_multi_frame_terms(const double *X, const int32_t *cnt, int32_t n, const double *TAGVcyl,
                                      const double *T, double radius, double *terms, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CPE_H */
