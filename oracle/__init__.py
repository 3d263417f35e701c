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
    global _LIB
    if _LIB is None:
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
