"""ctypes loader for libcpe_hip.so (C ABI: include/cpe.h).

There is NO CPU fallback: if the HIP library is missing or a call fails, this raises."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, 'libcpe_hip.so')
_lib = None


class CpeError(RuntimeError):
    pass


def build(verbose=False):
    """compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ['make', '-C', os.path.join(_HERE, 'csrc'), '-j4']
    if not verbose:
        cmd.insert(1, '-s')
    subprocess.check_call(cmd)
    return SO_PATH


_SIGS = {
    'cpe_version': (C.c_int32, []),
    'cpe_last_error_string': (C.c_char_p, []),
    'cpe_profile_enable': (None, [C.c_int32]),
    'cpe_profile_report': (C.c_int32, [C.c_char_p, C.c_size_t]),
    'cpe_preprocess_batch': (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    'cpe_detect_workspace_bytes': (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    'cpe_detect_constants': (C.c_int32, [C.c_int32, C.c_void_p]),
    'cpe_detect_grid_batch': (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t] +
                              [C.c_void_p] * 6),
    'cpe_detect_grid_batch_ex': (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t] +
                                 [C.c_void_p] * 6),
    'cpe_detect_grid_bgr_batch_ex': (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t] +
                                     [C.c_void_p] * 6),
    'cpe_detect_line_tables': (C.c_int32, [C.c_void_p, C.c_size_t, C.c_int32, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 5),
    'cpe_bgr2gray_batch': (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    'cpe_detect_workspace_plane': (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    'cpe_debug_external_components': (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int32,
                                                  C.c_void_p, C.c_void_p]),
    'cpe_debug_ccl': (C.c_int32, [C.c_void_p] + [C.c_int32] * 9 + [C.c_void_p, C.c_size_t, C.c_void_p]),
    'cpe_fit_workspace_bytes': (C.c_size_t, [C.c_int32]),
    'cpe_select_triangulate_batch': (C.c_int32, [C.c_void_p] * 6 + [C.c_int32] + [C.c_void_p] * 3 +
                                     [C.c_int32, C.c_int32, C.c_double, C.c_void_p, C.c_size_t] + [C.c_void_p] * 9),
    'cpe_choose_idx_batch': (C.c_int32, [C.c_void_p] * 6 + [C.c_int32] + [C.c_void_p] * 3 + [C.c_int32, C.c_double, C.c_void_p, C.c_size_t] +
                             [C.c_void_p] * 6),
    'cpe_triangulate_batch': (C.c_int32, [C.c_void_p] * 3 + [C.c_int32] + [C.c_void_p] * 7),
    'cpe_multi_frame_terms': (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]),
    'cpe_fit_cylinder_ransac_batch': (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_void_p, C.c_void_p] +
                                      [C.c_void_p] * 9),
    'cpe_undistort_map': (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    'cpe_remap_bilinear_batch': (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'cpe_undistort_map_matlab': (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    'cpe_remap_cubic_batch': (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    'cpe_fit_cylinder_batch': (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_void_p] +
                               [C.c_void_p] * 7),
}


class CpeDetectParams(C.Structure):
    _fields_ = [('subpixel', C.c_int32), ('subpixel_window', C.c_int32), ('subpixel_step', C.c_double),
                ('target', C.c_int32), ('reserved', C.c_int32)]


class CpeDetectConstants(C.Structure):
    """include/cpe.h: the reference's inline constants as this build was compiled with them (a report, not a setter)"""
    _fields_ = [('blur_ksize', C.c_int32), ('hessian_sigma', C.c_double), ('sauvola_window', C.c_int32), ('sauvola_k', C.c_double),
                ('sauvola_R', C.c_double), ('open_len', C.c_int32), ('clahe_clip', C.c_double), ('clahe_tiles', C.c_int32),
                ('blob_thr_min', C.c_int32), ('blob_thr_step', C.c_int32), ('blob_thr_count', C.c_int32), ('blob_min_area', C.c_double),
                ('blob_max_area', C.c_double), ('blob_min_dist', C.c_double), ('blob_min_repeat', C.c_int32),
                ('disc_extra_radius', C.c_int32), ('spot_blur_ksize', C.c_int32), ('spot_threshold', C.c_int32),
                ('spot_small_radius', C.c_int32), ('spot_small_add', C.c_int32), ('spot_large_add', C.c_int32), ('frag_patch', C.c_int32),
                ('frag_min_pixels', C.c_int32), ('frag_max_pixels', C.c_int32), ('frag_kernel_base', C.c_int32),
                ('index_blur_ksize', C.c_int32), ('poly_degree', C.c_int32), ('plane_threshold', C.c_int32),
                ('plane_dilate_ksize', C.c_int32), ('max_points', C.c_int32), ('max_lines', C.c_int32), ('max_joints', C.c_int32),
                ('max_groups_per_dir', C.c_int32), ('max_joints_per_group', C.c_int32)]


def detect_constants(target='cylinder'):
    """dict of the reference's inline constants as the kernels use them (cpe_detect_constants)"""
    c = CpeDetectConstants()
    check(load().cpe_detect_constants(dict(cylinder=0, plane=1)[target], C.byref(c)), 'cpe_detect_constants')
    return {k: getattr(c, k) for k, _ in CpeDetectConstants._fields_}


class CpeFitParams(C.Structure):
    _fields_ = [('tol_x', C.c_double), ('tol_f', C.c_double), ('max_iter', C.c_int32), ('max_fun_evals', C.c_int32),
                ('mode', C.c_int32), ('reserved', C.c_int32)]


class CpeRansacParams(C.Structure):
    _fields_ = [('hypotheses', C.c_int32), ('sample', C.c_int32), ('tau', C.c_double), ('seed', C.c_uint64),
                ('frame0', C.c_uint64), ('hyp_iters', C.c_int32), ('reserved', C.c_int32)]


MAXP = 2048
MAXL = 256


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise CpeError(f'{SO_PATH} not found: run `python -c "import __graft_entry__ as g; g.build()"` '
                           '(the product path has no CPU fallback)')
        lib = C.CDLL(SO_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        msg = load().cpe_last_error_string().decode()
        raise CpeError(f'{what} failed (rc={rc}): {msg}')


def profile(on):
    load().cpe_profile_enable(1 if on else 0)


def profile_report():
    """-> list of (kernel, calls, total_ms) sorted by total time (descending); clears the timers"""
    buf = C.create_string_buffer(1 << 16)
    load().cpe_profile_report(buf, len(buf))
    out = []
    for line in buf.value.decode().splitlines():
        name, calls, ms = line.rsplit(',', 2)
        out.append((name, int(calls), float(ms)))
    return out


def declared_symbols():
    """names declared with CPE_API in include/cpe.h"""
    import re
    hdr = os.path.join(_HERE, '..', 'include', 'cpe.h')
    txt = open(hdr).read()
    return sorted(set(re.findall(r'CPE_API[^;(]*?\b(cpe_\w+)\s*\(', txt)))
