"""ORACLE -- test infrastructure only.  ctypes wrappers of the image stages a-2 .. a-14."""
import ctypes as C

import numpy as np

from . import lib, _p, _u8


class _Pt(C.Structure):
    _fields_ = [('x', C.c_int), ('y', C.c_int)]


def _morph(name, src, kw, kh):
    src = _u8(src); h, w = src.shape; dst = np.empty_like(src)
    getattr(lib(), name)(_p(src, C.c_uint8), h, w, kw, kh, _p(dst, C.c_uint8))
    return dst


def erode_rect(src, kw, kh): return _morph('orc_erode_rect', src, kw, kh)
def dilate_rect(src, kw, kh): return _morph('orc_dilate_rect', src, kw, kh)
def open_rect(src, kw, kh): return _morph('orc_open_rect', src, kw, kh)
def close_rect(src, kw, kh): return _morph('orc_close_rect', src, kw, kh)


def find_contours(mask, mode='external', method='simple'):
    """cv2.findContours -> list of (points (k,2) int32, is_hole) in OpenCV's order"""
    mask = _u8(mask); h, w = mask.shape
    L = lib()
    L.orc_find_contours.restype = C.c_void_p
    L.orc_contour_points.restype = C.POINTER(_Pt)
    cs = C.c_void_p(L.orc_find_contours(_p(mask, C.c_uint8), h, w, 0 if mode == 'external' else 1,
                                        2 if method == 'simple' else 1))
    out = []
    for i in range(L.orc_contours_count(cs)):
        n = L.orc_contour_size(cs, i)
        pts = np.ctypeslib.as_array(C.cast(L.orc_contour_points(cs, i), C.POINTER(C.c_int)), shape=(n, 2)).copy()
        out.append((pts, bool(L.orc_contour_is_hole(cs, i))))
    L.orc_contours_free(cs)
    return out


def contour_moments(pts):
    pts = np.ascontiguousarray(pts, np.int32)
    a = C.c_double(); b = C.c_double(); c = C.c_double()
    lib().orc_contour_moments(_p(pts, _Pt), len(pts), C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def contour_area(pts):
    pts = np.ascontiguousarray(pts, np.int32)
    lib().orc_contour_area.restype = C.c_double
    return lib().orc_contour_area(_p(pts, _Pt), len(pts))


def bounding_rect(pts):
    pts = np.ascontiguousarray(pts, np.int32); r = np.zeros(4, np.int32)
    lib().orc_bounding_rect(_p(pts, _Pt), len(pts), _p(r, C.c_int))
    return tuple(int(v) for v in r)


def convex_hull(pts):
    pts = np.ascontiguousarray(pts, np.int32); out = np.zeros((len(pts) + 1, 2), np.int32)
    n = lib().orc_convex_hull(_p(pts, _Pt), len(pts), _p(out, _Pt))
    return out[:n].copy()


def min_enclosing_circle(pts):
    pts = np.ascontiguousarray(pts, np.int32)
    cx = C.c_float(); cy = C.c_float(); r = C.c_float()
    lib().orc_min_enclosing_circle(_p(pts, _Pt), len(pts), C.byref(cx), C.byref(cy), C.byref(r))
    return cx.value, cy.value, r.value


def fill_poly(shape, pts, c=255):
    img = np.zeros(shape, np.uint8); pts = np.ascontiguousarray(pts, np.int32)
    lib().orc_fill_poly(_p(img, C.c_uint8), shape[0], shape[1], _p(pts, _Pt), len(pts), c)
    return img


def circle_fill(img, cx, cy, r, c=255):
    lib().orc_circle_fill(_p(img, C.c_uint8), img.shape[0], img.shape[1], int(cx), int(cy), int(r), c)
    return img


def ellipse_fill(img, cx, cy, a, b, c=0):
    lib().orc_ellipse_fill(_p(img, C.c_uint8), img.shape[0], img.shape[1], int(cx), int(cy), int(a), int(b), c)
    return img


def clahe(src, clip=4.5, tiles=(4, 4)):
    src = _u8(src); h, w = src.shape; dst = np.empty_like(src)
    lib().orc_clahe(_p(src, C.c_uint8), h, w, C.c_double(clip), tiles[0], tiles[1], _p(dst, C.c_uint8))
    return dst


def lab_l(gray):
    gray = _u8(gray); h, w = gray.shape; dst = np.empty_like(gray)
    lib().orc_lab_l(_p(gray, C.c_uint8), h, w, _p(dst, C.c_uint8))
    return dst


def simple_blob_detector(gray, cap=65536):
    gray = _u8(gray); h, w = gray.shape
    kp = np.zeros((cap, 3), np.float32); stats = np.zeros(17, np.int32)
    n = lib().orc_simple_blob_detector(_p(gray, C.c_uint8), h, w, _p(kp, C.c_float), cap, _p(stats, C.c_int))
    return kp[:min(n, cap)].copy(), stats


def detect_largest_blob(gray, clip=4.5):
    """-> (status, mask_contour u8, rect (x,y,w,h), clahe image, n keypoints)"""
    gray = _u8(gray); h, w = gray.shape
    mask = np.empty_like(gray); cl = np.empty_like(gray); rect = np.zeros(4, np.int32); nk = C.c_int(0)
    st = lib().orc_detect_largest_blob(_p(gray, C.c_uint8), h, w, C.c_double(clip), _p(mask, C.c_uint8),
                                       _p(rect, C.c_int), _p(cl, C.c_uint8), C.byref(nk))
    return st, mask, tuple(int(v) for v in rect), cl, nk.value


# ---------------------------------------------------------------- masks / expansion / CCL
def extract_joints(binary, cap=1 << 16):
    binary = _u8(binary); h, w = binary.shape
    hm = np.empty_like(binary); vm = np.empty_like(binary); cent = np.zeros((cap, 2), np.int32)
    n = lib().orc_extract_joints(_p(binary, C.c_uint8), h, w, _p(hm, C.c_uint8), _p(vm, C.c_uint8), _p(cent, C.c_int), cap)
    return hm, vm, cent[:min(n, cap)].copy()


def blur19(src):
    src = _u8(src); dst = np.empty_like(src)
    lib().orc_blur19(_p(src, C.c_uint8), src.shape[0], src.shape[1], _p(dst, C.c_uint8)); return dst


def blur7(src):
    src = _u8(src); dst = np.empty_like(src)
    lib().orc_blur7(_p(src, C.c_uint8), src.shape[0], src.shape[1], _p(dst, C.c_uint8)); return dst


def mask_roi_around_center(hmask, vmask, mask_contour, gray):
    hmask = _u8(hmask); vmask = _u8(vmask); mask_contour = _u8(mask_contour); gray = _u8(gray); h, w = gray.shape
    rh = np.zeros_like(gray); rv = np.zeros_like(gray); r0 = C.c_int(0); spot = np.zeros(4, np.int32)
    st = lib().orc_mask_roi_around_center(_p(hmask, C.c_uint8), _p(vmask, C.c_uint8), _p(mask_contour, C.c_uint8),
                                          _p(gray, C.c_uint8), h, w, _p(rh, C.c_uint8), _p(rv, C.c_uint8),
                                          C.byref(r0), _p(spot, C.c_int))
    return st, rh, rv, r0.value, tuple(int(v) for v in spot)


def pca_endpoints(pts):
    pts = np.ascontiguousarray(pts, np.float32); p1 = np.zeros(2, np.float32); p2 = np.zeros(2, np.float32)
    ok = lib().orc_pca_endpoints(_p(pts, C.c_float), len(pts), _p(p1, C.c_float), _p(p2, C.c_float))
    return (None, None) if not ok else (p1.copy(), p2.copy())


def rotated_line_kernel(size, angle):
    k = np.zeros((size, size), np.uint8)
    lib().orc_rotated_line_kernel(size, C.c_double(angle), _p(k, C.c_uint8)); return k


def expand_line_roi(mask_roi, mask_contour, kernel_size):
    mask_roi = _u8(mask_roi); mask_contour = _u8(mask_contour); h, w = mask_roi.shape
    out = np.empty_like(mask_roi); dbg = np.zeros(2, np.int32)
    lib().orc_expand_line_roi(_p(mask_roi, C.c_uint8), _p(mask_contour, C.c_uint8), h, w, kernel_size,
                              _p(out, C.c_uint8), _p(dbg, C.c_int))
    return out, tuple(int(v) for v in dbg)


def connected_components(mask):
    mask = _u8(mask); h, w = mask.shape; lab = np.zeros((h, w), np.int32)
    n = lib().orc_connected_components(_p(mask, C.c_uint8), h, w, _p(lab, C.c_int32))
    return n, lab


# ---------------------------------------------------------------- line sets (a-8 .. a-14)
MAXL, MAXLP = 256, 1024     # include/cpe.h: CPE_MAXL, CPE_MAXLP (oracle/src/orc_lines.h)


class LineSet(C.Structure):
    _fields_ = [('nlines', C.c_int), ('npts', C.c_int * MAXL), ('pts', ((C.c_double * 2) * MAXLP) * MAXL),
                ('eq', (C.c_double * 6) * MAXL), ('has_eq', C.c_int * MAXL), ('label', C.c_int * MAXL)]

    def points(self):
        return [[(self.pts[g][k][0], self.pts[g][k][1]) for k in range(self.npts[g])] for g in range(self.nlines)]

    def equations(self):
        return [list(self.eq[g]) for g in range(self.nlines)]


def group_points(centroids, labels, x_off, y_off):
    cent = np.ascontiguousarray(centroids, np.int32).reshape(-1, 2); labels = np.ascontiguousarray(labels, np.int32)
    ls = LineSet()
    assert C.sizeof(ls) == lib().orc_lineset_size()
    lib().orc_group_points(_p(cent, C.c_int), len(cent), _p(labels, C.c_int32), labels.shape[0], labels.shape[1],
                           x_off, y_off, C.byref(ls))
    return ls


def fit_lines(ls, is_row):
    lib().orc_fit_lines(C.byref(ls), 1 if is_row else 0); return ls


def remove_label(rows, cols):
    lib().orc_remove_label(C.byref(rows), C.byref(cols)); return rows, cols


def poly_intersection(row_eq, col_eq):
    a = (C.c_double * 6)(*row_eq); b = (C.c_double * 6)(*col_eq); x = C.c_double(); y = C.c_double()
    ok = lib().orc_poly_intersection(a, b, C.byref(x), C.byref(y))
    return (x.value, y.value) if ok else None


def intersections(rows, cols, rect):
    r = (C.c_int * 4)(*rect)
    lib().orc_intersections(C.byref(rows), C.byref(cols), r); return rows, cols


def clean_and_relabel(rows, cols):
    lib().orc_clean_and_relabel(C.byref(rows), C.byref(cols)); return rows, cols


def index_points(rows, cols, gauss7, r0, cap=4096):
    gauss7 = _u8(gauss7); h, w = gauss7.shape
    center = np.zeros(2); xy = np.zeros((cap, 2)); ids = np.zeros((cap, 2), np.int32)
    n = lib().orc_index_points(C.byref(rows), C.byref(cols), _p(gauss7, C.c_uint8), h, w, r0, _p(center, C.c_double),
                               _p(xy, C.c_double), _p(ids, C.c_int), cap)
    if n < 0:
        return n, center, xy[:0], ids[:0]
    return n, center, xy[:n].copy(), ids[:n].copy()


class DetectDebug(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ('binary', 'hmask', 'vmask', 'mask_contour', 'roi_h', 'roi_v', 'exp_h', 'exp_v')] + \
               [('joints', C.c_void_p), ('cap_joints', C.c_int), ('n_joints', C.c_int), ('n_cyl_joints', C.c_int),
                ('rect', C.c_int * 4), ('r0', C.c_int), ('spot', C.c_int * 4), ('n_rows', C.c_int), ('n_cols', C.c_int),
                ('n_keypoints', C.c_int), ('rows_out', C.c_void_p), ('cols_out', C.c_void_p)]


def bgr2gray(bgr):
    """cv2.cvtColor(BGR2GRAY), 8-bit"""
    bgr = np.ascontiguousarray(bgr, np.uint8); h, w, c = bgr.shape
    assert c == 3
    out = np.empty((h, w), np.uint8)
    lib().orc_bgr2gray(_p(bgr, C.c_uint8), C.c_size_t(h * w), _p(out, C.c_uint8))
    return out


def line_dict(ls, prefix):
    """LineSet after clean_and_relabel -> {'points': {'<prefix>1': [(x, y), ..]}, 'equations': {'<prefix>1': [6 floats]}}"""
    pts = ls.points(); eqs = ls.equations()
    return {'points': {f'{prefix}{g + 1}': pts[g] for g in range(ls.nlines)},
            'equations': {f'{prefix}{g + 1}': eqs[g] for g in range(ls.nlines)}}


def detect_grid(gray, cap=4096, debug=False, subpixel=False, window=7, step=1.0, lines=False):
    """detect_grid restated: -> dict(status, center (2,), xy (n,2), id (n,2) [, debug images][, rows / cols: the third and
    fourth return values of the reference's detect_grid])"""
    gray = _u8(gray); h, w = gray.shape
    center = np.zeros(2); xy = np.zeros((cap, 2)); ids = np.zeros((cap, 2), np.int32); n = C.c_int(0)
    dbg = DetectDebug(); imgs = {}
    if lines:
        rows_ls, cols_ls = LineSet(), LineSet()
        dbg.rows_out = C.addressof(rows_ls); dbg.cols_out = C.addressof(cols_ls)
    if debug:
        for k in ('binary', 'hmask', 'vmask', 'mask_contour', 'roi_h', 'roi_v', 'exp_h', 'exp_v'):
            imgs[k] = np.zeros((h, w), np.uint8)
            setattr(dbg, k, imgs[k].ctypes.data)
        imgs['joints'] = np.zeros((1 << 16, 2), np.int32)
        dbg.joints = imgs['joints'].ctypes.data; dbg.cap_joints = 1 << 16
    st = lib().orc_detect_grid_ex(_p(gray, C.c_uint8), h, w, 1 if subpixel else 0, window, C.c_double(step),
                                  _p(center, C.c_double), _p(xy, C.c_double), _p(ids, C.c_int), cap, C.byref(n), C.byref(dbg))
    out = dict(status=st, center=center, xy=xy[:n.value].copy(), id=ids[:n.value].copy(), rect=tuple(dbg.rect),
               r0=dbg.r0, spot=tuple(dbg.spot), n_joints=dbg.n_joints, n_cyl_joints=dbg.n_cyl_joints,
               n_rows=dbg.n_rows, n_cols=dbg.n_cols, n_keypoints=dbg.n_keypoints)
    if debug:
        imgs['joints'] = imgs['joints'][:dbg.n_joints].copy()
        out.update(imgs)
    if lines:
        out['rows'] = line_dict(rows_ls, 'row'); out['cols'] = line_dict(cols_ls, 'col')
    return out


def detect_grid_bgr(bgr, cap=4096):
    """detect_grid on a true-colour frame H x W x 3 (BGR): the blob stage sees the L channel of BGR2LAB of the colour image,
    indexing_data the 7x7 blur of the colour image converted to grey (util_cylinder.py:1840, :1433-1435)"""
    bgr = np.ascontiguousarray(bgr, np.uint8); h, w, _ = bgr.shape
    center = np.zeros(2); xy = np.zeros((cap, 2)); ids = np.zeros((cap, 2), np.int32); n = C.c_int(0)
    dbg = DetectDebug()
    st = lib().orc_detect_grid_bgr(_p(bgr, C.c_uint8), h, w, _p(center, C.c_double), _p(xy, C.c_double), _p(ids, C.c_int), cap,
                                   C.byref(n), C.byref(dbg))
    return dict(status=st, center=center, xy=xy[:n.value].copy(), id=ids[:n.value].copy(), rect=tuple(dbg.rect), r0=dbg.r0,
                n_keypoints=dbg.n_keypoints)


def lab_l_bgr(bgr):
    bgr = np.ascontiguousarray(bgr, np.uint8); h, w, _ = bgr.shape; out = np.zeros((h, w), np.uint8)
    lib().orc_lab_l_bgr(_p(bgr, C.c_uint8), h, w, _p(out, C.c_uint8)); return out


def lineset_from_equations(eqs):
    """LineSet carrying only equations (for the sub-pixel refinement tests)"""
    ls = LineSet(); ls.nlines = len(eqs)
    for g, e in enumerate(eqs):
        for k in range(6):
            ls.eq[g][k] = float(e[k])
    return ls


def subpixel_refine(gray, rows, cols, window=7, step=1.0):
    """modify_grayscale_Cline(gray2d, rows, cols, draw_points=False, degree=2, step, window): status 0 or 7 (raises)"""
    gray = _u8(gray); h, w = gray.shape
    return lib().orc_subpixel_refine(_p(gray, C.c_uint8), h, w, C.byref(rows), C.byref(cols), window, C.c_double(step))


# ---------------------------------------------------------------- row f-2: planar-target variant (orc_plane.c)
def fit_lines_plane(rows, cols):
    lib().orc_fit_lines_plane(C.byref(rows), C.byref(cols)); return rows, cols


def intersections_plane(rows, cols, rect):
    r = (C.c_int * 4)(*rect)
    lib().orc_intersections_plane(C.byref(rows), C.byref(cols), r); return rows, cols


def clean_plane(rows, cols):
    lib().orc_clean_plane(C.byref(rows)); lib().orc_clean_plane(C.byref(cols)); return rows, cols


def get_convex_hull(gray, thr=127, expansion=5):
    gray = _u8(gray); h, w = gray.shape
    mask = np.zeros((h, w), np.uint8); rect = (C.c_int * 4)()
    st = lib().orc_get_convex_hull(_p(gray, C.c_uint8), h, w, thr, expansion, _p(mask, C.c_uint8), rect)
    return st, mask, tuple(rect)


def ellipse_se(ks):
    se = np.zeros((ks, ks), np.uint8)
    lib().orc_ellipse_se(ks, _p(se, C.c_uint8)); return se


def detect_grid_plane(gray, cap=4096, debug=False):
    """detect_grid of python_grid_detection_plane.py restated: -> dict(status, center, xy (n,2), id (n,2) = (row, col) ...)"""
    gray = _u8(gray); h, w = gray.shape
    center = np.zeros(2); xy = np.zeros((cap, 2)); ids = np.zeros((cap, 2), np.int32); n = C.c_int(0)
    dbg = DetectDebug(); imgs = {}
    if debug:
        for k in ('binary', 'hmask', 'vmask', 'mask_contour', 'roi_h', 'roi_v', 'exp_h', 'exp_v'):
            imgs[k] = np.zeros((h, w), np.uint8)
            setattr(dbg, k, imgs[k].ctypes.data)
        imgs['joints'] = np.zeros((1 << 16, 2), np.int32)
        dbg.joints = imgs['joints'].ctypes.data; dbg.cap_joints = 1 << 16
    st = lib().orc_detect_grid_plane(_p(gray, C.c_uint8), h, w, _p(center, C.c_double), _p(xy, C.c_double), _p(ids, C.c_int), cap,
                                     C.byref(n), C.byref(dbg))
    out = dict(status=st, center=center, xy=xy[:n.value].copy(), id=ids[:n.value].copy(), rect=tuple(dbg.rect),
               r0=dbg.r0, spot=tuple(dbg.spot), n_joints=dbg.n_joints, n_cyl_joints=dbg.n_cyl_joints,
               n_rows=dbg.n_rows, n_cols=dbg.n_cols, n_keypoints=0)
    if debug:
        imgs['joints'] = imgs['joints'][:dbg.n_joints].copy()
        out.update(imgs)
    return out
