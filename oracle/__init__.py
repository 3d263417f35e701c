"""ORACLE -- test infrastructure only (see oracle/src/orc_common.h).

ctypes face of oracle/liboracle.so: a plain-C, single-thread restatement of the reference hot path.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """compile oracle/liboracle.so with gcc (make)."""
    so = os.path.join(_HERE, 'liboracle.so')
    if force or not os.path.exists(so) or any(
            os.path.getmtime(os.path.join(_HERE, 'src', f)) > os.path.getmtime(so)
            for f in os.listdir(os.path.join(_HERE, 'src')) if f.endswith(('.c', '.h'))):
        subprocess.check_call(['make', '-s', '-C', _HERE], stdout=subprocess.DEVNULL)
    return so


def lib():
    """ORACLE_ASAN=1: the AddressSanitizer + UBSan build (`make -C oracle asan`; the process must have been started with
    LD_PRELOAD=libasan.so, see oracle/Makefile)"""
    global _LIB
    if _LIB is None:
        if os.environ.get('ORACLE_ASAN') == '1':
            so = os.path.join(_HERE, 'liboracle_asan.so')
            if not os.path.exists(so):
                subprocess.check_call(['make', '-s', '-C', _HERE, 'asan'], stdout=subprocess.DEVNULL)
        else:
            so = os.path.join(_HERE, 'liboracle.so')
            if not os.path.exists(so):
                build()
        _LIB = C.CDLL(so)
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a


# ---------------------------------------------------------------- stage a-1
def blur5(gray):
    gray = _u8(gray); h, w = gray.shape
    out = np.empty_like(gray)
    lib().orc_blur5(_p(gray, C.c_uint8), h, w, _p(out, C.c_uint8))
    return out


def gauss_sigma3(img):
    img = _u8(img); h, w = img.shape
    G = np.empty((h, w), np.float64)
    lib().orc_gauss_sigma3(_p(img, C.c_uint8), h, w, _p(G, C.c_double))
    return G


def hessian_eigs(G):
    G = np.ascontiguousarray(G, np.float64); h, w = G.shape
    emin = np.empty_like(G); emax = np.empty_like(G)
    lib().orc_hessian_eigs(_p(G, C.c_double), h, w, _p(emin, C.c_double), _p(emax, C.c_double))
    return emax, emin


def detect_ridges(img_u8):
    """restatement of util_cylinder.detect_ridges(gray, sigma=3.0) -> (larger, smaller)"""
    return hessian_eigs(gauss_sigma3(img_u8))


def sauvola_mask(b):
    b = np.ascontiguousarray(b, np.float64); h, w = b.shape
    m = np.empty((h, w), np.uint8)
    lib().orc_sauvola_mask(_p(b, C.c_double), h, w, _p(m, C.c_uint8))
    return m


def preprocess(gray, want_b=False):
    """load_and_preprocess_image: gray u8 -> (blurred u8, binary mask u8[, b f64])"""
    gray = _u8(gray); h, w = gray.shape
    blurred = np.empty_like(gray); mask = np.empty_like(gray)
    b = np.empty((h, w), np.float64) if want_b else None
    lib().orc_preprocess(_p(gray, C.c_uint8), h, w, _p(blurred, C.c_uint8), _p(mask, C.c_uint8),
                         _p(b, C.c_double) if want_b else None)
    return (blurred, mask, b) if want_b else (blurred, mask)


# ---------------------------------------------------------------- geometric half (MATLAB side)
def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def triangulate(p1, p2, K1, K2, T21):
    """[ext] MATLAB triangulate: (n,2),(n,2) -> X (n,3) in camera-1 frame, per-point reprojection error"""
    p1 = _f64(p1); p2 = _f64(p2); n = p1.shape[0]
    X = np.empty((n, 3)); err = np.empty(n)
    K1 = _f64(K1); K2 = _f64(K2); T21 = _f64(T21)
    lib().orc_triangulate(_p(p1, C.c_double), _p(p2, C.c_double), n, _p(K1, C.c_double), _p(K2, C.c_double),
                          _p(T21, C.c_double), _p(X, C.c_double), _p(err, C.c_double))
    return X, err


def choose_idx(gp1, gp2, K1, K2, T21, patch=3, th=0.3):
    """chooseIdx.m -> (cgp1 (m,2), cgp2 (m,2), idx (m,2) int, used_fallback)"""
    gp1 = _f64(gp1); gp2 = _f64(gp2); cap = max(len(gp1), len(gp2)) + 1
    c1 = np.empty((cap, 2)); c2 = np.empty((cap, 2)); idx = np.empty((cap, 2), np.int32)
    fb = C.c_int(0)
    K1 = _f64(K1); K2 = _f64(K2); T21 = _f64(T21)
    m = lib().orc_choose_idx(_p(gp1, C.c_double), len(gp1), _p(gp2, C.c_double), len(gp2), _p(K1, C.c_double),
                             _p(K2, C.c_double), _p(T21, C.c_double), patch, C.c_double(th), _p(c1, C.c_double),
                             _p(c2, C.c_double), _p(idx, C.c_int), C.byref(fb))
    return c1[:m].copy(), c2[:m].copy(), idx[:m].copy(), bool(fb.value)


def find_correspondences(gp1, gp2):
    gp1 = _f64(gp1); gp2 = _f64(gp2); cap = len(gp1) + 1
    c1 = np.empty((cap, 2)); c2 = np.empty((cap, 2)); idx = np.empty((cap, 2), np.int32)
    m = lib().orc_find_correspondences(_p(gp1, C.c_double), len(gp1), _p(gp2, C.c_double), len(gp2),
                                       _p(c1, C.c_double), _p(c2, C.c_double), _p(idx, C.c_int))
    return c1[:m].copy(), c2[:m].copy(), idx[:m].copy()


def dist_pts3_to_line(P, p1, p2):
    """getDistPts3ToLine.m (P is (n,3))"""
    P = _f64(P); d = np.empty(len(P)); p1 = _f64(p1); p2 = _f64(p2)
    lib().orc_dist_pts3_to_line(_p(P, C.c_double), len(P), _p(p1, C.c_double), _p(p2, C.c_double), _p(d, C.c_double))
    return d


def cyl_objective(x, P, R):
    lib().orc_cyl_objective.restype = C.c_double
    x = _f64(x); P = _f64(P)
    return lib().orc_cyl_objective(_p(x, C.c_double), _p(P, C.c_double), len(P), C.c_double(R))


def fit_cylinder(P, R, tolx=1e-5, tolf=1e-5, maxiter=100000, maxfun=100000, mode=0):
    """fitCylinderWPts3.m: P (n,3) -> dict(cyl0[6], cyl[6], fvals[2], iters, evals, status); mode 1 = the build's LM"""
    P = _f64(P)
    cyl0 = np.empty(6); cyl = np.empty(6); fv = np.empty(2); it = C.c_int(0); ev = C.c_int(0)
    st = lib().orc_fit_cylinder_mode(_p(P, C.c_double), len(P), C.c_double(R), C.c_double(tolx), C.c_double(tolf),
                                     maxiter, maxfun, mode, _p(cyl0, C.c_double), _p(cyl, C.c_double), _p(fv, C.c_double),
                                     C.byref(it), C.byref(ev))
    return dict(cyl0=cyl0, cyl=cyl, fvals=fv, iters=it.value, evals=ev.value, status=st)


def fit_cylinder_ransac(P, R, hypotheses=64, sample=12, tau=0.5, seed=0, frame=0, hyp_iters=8, tolx=1e-5, tolf=1e-5,
                        maxiter=100000, maxfun=100000, mode=1):
    """BUILD-DEFINED RANSAC around the fit (orc_fit_cylinder_ransac): -> dict(cyl0, cyl (raw), fvals, iters, evals, n_inliers, mask, status)"""
    P = _f64(P)
    cyl0 = np.empty(6); cyl = np.empty(6); fv = np.empty(2); it = C.c_int(0); ev = C.c_int(0); ni = C.c_int(0)
    mask = np.zeros(len(P), dtype=np.uint8)
    st = lib().orc_fit_cylinder_ransac(_p(P, C.c_double), len(P), C.c_double(R), hypotheses, sample, C.c_double(tau),
                                       C.c_uint64(seed), C.c_uint64(frame), hyp_iters, C.c_double(tolx), C.c_double(tolf), maxiter,
                                       maxfun, mode, _p(cyl0, C.c_double), _p(cyl, C.c_double), _p(fv, C.c_double), C.byref(it),
                                       C.byref(ev), C.byref(ni), _p(mask, C.c_uint8))
    return dict(cyl0=cyl0, cyl=cyl, fvals=fv, iters=it.value, evals=ev.value, n_inliers=ni.value, mask=mask, status=st)


def apply_prior(cyl, P):
    cyl = _f64(cyl).copy(); P = _f64(P)
    lib().orc_apply_prior(_p(cyl, C.c_double), _p(P, C.c_double), len(P))
    return cyl


def cyl2T(cyl):
    cyl = _f64(cyl); T = np.empty((4, 4))
    lib().orc_cyl2T(_p(cyl, C.c_double), _p(T, C.c_double))
    return T


def fit_single_cylinder(gp1, gp2, K1, K2, T21, R, selector=0, th=0.3):
    """fitSingleCylinder.m -> dict(pts3 (m,3), cyl (2,6), T (4,4), fvals, mean_err, iters, evals, status, fallback)"""
    gp1 = _f64(gp1); gp2 = _f64(gp2); cap = max(len(gp1), len(gp2)) + 1
    pts3 = np.empty((cap, 3)); cyl = np.zeros((2, 6)); T = np.zeros((4, 4)); fv = np.zeros(2)
    m = C.c_int(0); me = C.c_double(0); it = C.c_int(0); ev = C.c_int(0); fb = C.c_int(0)
    K1 = _f64(K1); K2 = _f64(K2); T21 = _f64(T21)
    st = lib().orc_fit_single_cylinder(_p(gp1, C.c_double), len(gp1), _p(gp2, C.c_double), len(gp2),
                                       _p(K1, C.c_double), _p(K2, C.c_double), _p(T21, C.c_double), C.c_double(R),
                                       selector, C.c_double(th), _p(pts3, C.c_double), C.byref(m),
                                       _p(cyl, C.c_double), _p(T, C.c_double), _p(fv, C.c_double), C.byref(me),
                                       C.byref(it), C.byref(ev), C.byref(fb))
    return dict(pts3=pts3[:m.value].copy(), cyl=cyl, T=T, fvals=fv, mean_err=me.value, iters=it.value,
                evals=ev.value, status=st, fallback=bool(fb.value))


# ---------------------------------------------------------------- row f-1: multi-frame AGV-pose fit
def get_TAGVcyl(pan, tilt):
    T = np.empty(16)
    lib().orc_get_TAGVcyl(C.c_double(pan), C.c_double(tilt), _p(T, C.c_double))
    return T


def vec2T(x):
    x = _f64(x); T = np.empty(16)
    lib().orc_vec2T(_p(x, C.c_double), _p(T, C.c_double)); return T


def T2vec(T):
    T = _f64(T).ravel(); x = np.empty(6)
    lib().orc_T2vec(_p(T, C.c_double), _p(x, C.c_double)); return x


def multi_objective(x, P, cnt, TAGV, R):
    x = _f64(x); P = _f64(P); cnt = np.ascontiguousarray(cnt, np.int32); TAGV = _f64(TAGV)
    lib().orc_multi_objective.restype = C.c_double
    return lib().orc_multi_objective(_p(x, C.c_double), _p(P, C.c_double), _p(cnt, C.c_int), len(cnt), P.shape[1],
                                     _p(TAGV, C.c_double), C.c_double(R))


def multi_fit(P, cnt, TAGV, cyl_raw, R):
    """fitCylinderWPts3sAngs: P (F,cap,3), cnt (F,), TAGV (F,16), cyl_raw (F,2,6) -> dict(x0, x, T, fvals, iters, evals)"""
    P = _f64(P); cnt = np.ascontiguousarray(cnt, np.int32); TAGV = _f64(TAGV); cyl_raw = _f64(cyl_raw)
    x0 = np.empty(6); x = np.empty(6); T = np.empty(16); fv = np.empty(2); it = C.c_int(0); ev = C.c_int(0)
    lib().orc_multi_fit(_p(P, C.c_double), _p(cnt, C.c_int), len(cnt), P.shape[1], _p(TAGV, C.c_double),
                        _p(cyl_raw, C.c_double), C.c_double(R), _p(x0, C.c_double), _p(x, C.c_double), _p(T, C.c_double),
                        _p(fv, C.c_double), C.byref(it), C.byref(ev))
    return dict(x0=x0, x=x, T=T, fvals=fv, iters=it.value, evals=ev.value)


# ---- row f-3: cv2.undistort restatement (orc_undistort.c) -----------------------------------------------------
def undistort_map(K, dist, h, w):
    """-> (map_xy int16 [h,w,2], map_f uint16 [h,w]) as cv2.initUndistortRectifyMap(..., CV_16SC2) inside cv2.undistort"""
    L = lib()
    K = np.ascontiguousarray(np.asarray(K, dtype=np.float64).reshape(9))
    d = np.ascontiguousarray(np.asarray(dist, dtype=np.float64).ravel())
    mxy = np.empty((h, w, 2), dtype=np.int16); mf = np.empty((h, w), dtype=np.uint16)
    rc = L.orc_undistort_map(K.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), C.c_int(d.size), C.c_int(h), C.c_int(w),
                             mxy.ctypes.data_as(C.c_void_p), mf.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise ValueError('orc_undistort_map: singular matrix or unsupported coefficient count')
    return mxy, mf


def remap_bilinear(src, map_xy, map_f):
    L = lib()
    src = np.ascontiguousarray(src, dtype=np.uint8)
    h, w = src.shape
    dst = np.empty_like(src)
    L.orc_remap_bilinear(src.ctypes.data_as(C.c_void_p), C.c_int(h), C.c_int(w), np.ascontiguousarray(map_xy).ctypes.data_as(C.c_void_p),
                         np.ascontiguousarray(map_f).ctypes.data_as(C.c_void_p), dst.ctypes.data_as(C.c_void_p))
    return dst


def undistort(src, K, dist):
    mxy, mf = undistort_map(K, dist, src.shape[0], src.shape[1])
    return remap_bilinear(src, mxy, mf)


def undistort_map_matlab(K, radial, tangential, h, w):
    """distortPoints of every output pixel (undistortImage, OutputView 'same') -> float32 [h,w,2] 0-based source (x, y)"""
    L = lib()
    K = np.ascontiguousarray(np.asarray(K, dtype=np.float64).reshape(9))
    r = np.ascontiguousarray(np.asarray(radial, dtype=np.float64).ravel()); t = np.ascontiguousarray(np.asarray(tangential, dtype=np.float64).ravel())
    m = np.empty((h, w, 2), dtype=np.float32)
    L.orc_undistort_map_matlab(K.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p), C.c_int(r.size),
                               t.ctypes.data_as(C.c_void_p) if t.size else None, C.c_int(h), C.c_int(w), m.ctypes.data_as(C.c_void_p))
    return m


def remap_cubic(src, map_xy, fill=0):
    """interp2d(src, X, Y, 'cubic', fill) on a uint8 image"""
    L = lib()
    src = np.ascontiguousarray(src, dtype=np.uint8); h, w = src.shape
    m = np.ascontiguousarray(map_xy, dtype=np.float32); dst = np.empty_like(src)
    L.orc_remap_cubic(src.ctypes.data_as(C.c_void_p), C.c_int(h), C.c_int(w), m.ctypes.data_as(C.c_void_p), C.c_int(fill),
                      dst.ctypes.data_as(C.c_void_p))
    return dst


def undistort_matlab(src, K, radial, tangential):
    return remap_cubic(src, undistort_map_matlab(K, radial, tangential, src.shape[0], src.shape[1]))
