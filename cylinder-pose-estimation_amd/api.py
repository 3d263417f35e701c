"""Python face of the image half of the hot path.

`detect_grid_batch` is the MI355X-native form of python_grid_detection_cylinder.py::detect_grid
(:68-112): all frames of a batch go through the HIP kernels behind the C ABI (include/cpe.h) and come
back as padded point tables; `detect_grid(input_img)` keeps the reference's single-image signature and
return shape, `make_json` its JSON (util_cylinder.py:1674-1727, decoded by makePyGridPts.m:39-41), `save_mat`
writes the structs the MATLAB side holds after makePyGridPts / fitSingleCylinder.
There is no CPU fallback: without a GPU and libcpe_hip.so these raise."""
import ctypes as C
import json

import numpy as np
import torch

from . import lib as _lib
from .fit import GridTables, MAXP

PLANES = dict(binary=0, hmask=1, vmask=2, mask_contour=3, roi_h=4, roi_v=5, exp_h=6, exp_v=7, joints=8, state=9,
              clahe=10, blur19=11, blur7=12, labels=13, sweep=14)
TARGETS = dict(cylinder=0, plane=1)
STATUS_TEXT = {0: 'ok', 1: 'no region (cv2.convexHull(None))', 2: 'no saturated spot (circle_radius0 unbound)',
               3: 'no valid rows/cols', 4: 'empty point list', 5: 'too few points', 6: 'workspace capacity exceeded',
               7: 'sub-pixel refinement raised (line sample above / left of the image)'}

_STATE_FIELDS = ['status', 'rect0', 'rect1', 'rect2', 'rect3', 'r0', 'spot0', 'spot1', 'spot2', 'spot3', 'n_roots',
                 'n_comps', 'n_joints_all', 'n_joints', 'n_blobs', 'n_groups', 'n_groups_prev', 'n_kp', 'n_verts',
                 'n_dists', 'best_comp', 'n_seg0', 'n_seg1', 'gang0', 'gang1', 'glen0', 'glen1', 'n_rows', 'n_cols',
                 'overflow', 'hull_n', 'crect0', 'crect1', 'crect2', 'crect3', 'nrect0', 'nrect1', 'nrect2', 'nrect3', 'n_roots_p', 'n_roots_s', 'spot_fail',
                 'srect0', 'srect1', 'srect2', 'srect3']


class DetectWorkspace:
    """device scratch for cpe_detect_grid_batch, reusable across calls with the same frame size and up to n frames (a
    smaller batch, e.g. the ragged last chunk of a run, is laid out inside the same buffer)"""

    def __init__(self, n, h, w, device):
        self.n, self.h, self.w = n, h, w
        self.capacity_n = n
        L = _lib.load()
        self.bytes = L.cpe_detect_workspace_bytes(n, h, w)
        self.capacity = self.bytes
        self.buf = torch.empty(self.bytes + 256, dtype=torch.uint8, device=device)
        off = (-self.buf.data_ptr()) % 256
        self.view = self.buf[off:off + self.bytes]

    def fits(self, n, h, w):
        return (h, w) == (self.h, self.w) and n <= self.capacity_n and \
            _lib.load().cpe_detect_workspace_bytes(n, h, w) <= self.capacity

    def use(self, n):
        """lay the buffer out for a batch of n frames (n <= the n it was made for).  Every use is a new generation of the
        buffer's contents: a result dict remembers the generation it was made in, and the functions that read
        intermediates through it (line_tables, frame_result) refuse a workspace that has served another call since."""
        if n != self.n:
            self.n = n
            self.bytes = _lib.load().cpe_detect_workspace_bytes(n, self.h, self.w)
        self.generation = getattr(self, 'generation', 0) + 1
        return self

    def plane(self, name):
        """intermediate of the last call: u8 [n,h,w] planes, i32 [n,CPE_MAXJ,2] joints, or the state records"""
        L = _lib.load()
        off = C.c_size_t(); per = C.c_size_t()
        _lib.check(L.cpe_detect_workspace_plane(self.n, self.h, self.w, PLANES[name], C.byref(off), C.byref(per)),
                   'cpe_detect_workspace_plane')
        raw = self.view[off.value:off.value + per.value * self.n]
        if name == 'joints':
            return raw.view(torch.int32).reshape(self.n, -1, 2)
        if name == 'labels':
            return raw.view(torch.int32).reshape(self.n, self.h, self.w)
        if name in ('state', 'sweep'):
            return raw.view(torch.int32).reshape(self.n, -1)
        return raw.reshape(self.n, self.h, self.w)

    def state(self):
        """list of dicts (one per frame) of the per-frame state record"""
        arr = self.plane('state').cpu().numpy()
        out = []
        for row in arr:
            d = {}
            for k, name in enumerate(_STATE_FIELDS):
                v = row[k]
                d[name] = float(np.int32(v).view(np.float32)) if name[:4] in ('gang', 'glen') else int(v)
            out.append(d)
        return out


def detect_grid_batch(frames, ws=None, subpixel=False, subpixel_window=7, subpixel_step=1.0, target='cylinder'):
    """frames: u8 tensor [n,h,w] (grey) or [n,h,w,3] (BGR, as cv2.imread delivers) on the GPU -> dict(xy f64[n,MAXP,2],
    id i32[n,MAXP,2], n i32[n], center f64[n,2], status i32[n], ws).  target='plane': the planar-target script
    (python_grid_detection_plane.py, row f-2); ids are (row, col) there."""
    if not (isinstance(frames, torch.Tensor) and frames.is_cuda and frames.dtype == torch.uint8 and
            (frames.dim() == 3 or (frames.dim() == 4 and frames.shape[3] == 3))):
        raise TypeError('frames must be a CUDA uint8 tensor [n,h,w] (grey) or [n,h,w,3] (BGR)')
    frames = frames.contiguous()
    colour = frames.dim() == 4
    if colour and (target != 'cylinder' or subpixel):
        frames = bgr_to_gray(frames); colour = False       # the planar script / the sub-pixel stage: luma only
    n, h, w = frames.shape[:3]
    dev = frames.device
    L = _lib.load()
    if ws is None or not ws.fits(n, h, w) or ws.view.device != dev:
        ws = DetectWorkspace(n, h, w, dev)
    ws.use(n)
    xy = torch.zeros((n, MAXP, 2), dtype=torch.float64, device=dev)
    ids = torch.zeros((n, MAXP, 2), dtype=torch.int32, device=dev)
    cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    center = torch.zeros((n, 2), dtype=torch.float64, device=dev)
    status = torch.zeros(n, dtype=torch.int32, device=dev)
    prm = _lib.CpeDetectParams(1 if subpixel else 0, subpixel_window, subpixel_step, TARGETS[target], 0)
    entry = L.cpe_detect_grid_bgr_batch_ex if colour else L.cpe_detect_grid_batch_ex
    _lib.check(entry(frames.data_ptr(), n, h, w, C.addressof(prm), ws.view.data_ptr(), ws.bytes,
                     xy.data_ptr(), ids.data_ptr(), cnt.data_ptr(), center.data_ptr(),
                     status.data_ptr(), torch.cuda.current_stream().cuda_stream),
               'cpe_detect_grid_bgr_batch_ex' if colour else 'cpe_detect_grid_batch_ex')
    return dict(xy=xy, id=ids, n=cnt, center=center, status=status, ws=ws, ws_generation=ws.generation)


def tables_of(det):
    """detect_grid_batch output -> GridTables (the N x 4 [x y col row] matrices of makePyGridPts.m:41)"""
    return GridTables(det['xy'], det['id'], det['n'])


def make_json(center, xy, ids):
    """the JSON string make_json returns (util_cylinder.py:1674-1727): indent 4, keys id/x/y in this order"""
    pts = [{"id": [int(c), int(r)], "x": float(x), "y": float(y)} for (x, y), (c, r) in zip(xy, ids)]
    return json.dumps({"center_point": [float(center[0]), float(center[1])], "points": pts}, indent=4, ensure_ascii=False)


def bgr_to_gray(bgr):
    """u8 tensor [n,h,w,3] (BGR, interleaved) on the GPU -> u8 [n,h,w]: cv2.cvtColor(BGR2GRAY) of
    load_and_preprocess_image (util_cylinder.py:1781-1789), cpe_bgr2gray_batch"""
    if not (isinstance(bgr, torch.Tensor) and bgr.is_cuda and bgr.dtype == torch.uint8 and bgr.dim() == 4 and bgr.shape[3] == 3):
        raise TypeError('bgr must be a CUDA uint8 tensor [n,h,w,3]')
    bgr = bgr.contiguous()
    n, h, w, _ = bgr.shape
    gray = torch.empty((n, h, w), dtype=torch.uint8, device=bgr.device)
    with torch.cuda.device(bgr.device):
        _lib.check(_lib.load().cpe_bgr2gray_batch(bgr.data_ptr(), n, h, w, gray.data_ptr(), torch.cuda.current_stream().cuda_stream),
                   'cpe_bgr2gray_batch')
    return gray


def frames_to_device(images, device='cuda:0'):
    """list of numpy u8 images of one size, each H x W (grey) or H x W x 3 (BGR as cv2.imread gives it) -> u8 tensor on the
    device: [n,h,w] if every image is grey, else [n,h,w,3] (grey images replicated into the three channels: the colour path
    makes of a grey-replicated frame exactly what the grey path makes of the plane)."""
    arrs = [np.asarray(a) for a in images]
    for a in arrs:
        if a.dtype != np.uint8:
            raise TypeError('detect_grid expects uint8 images')
        if not (a.ndim == 2 or (a.ndim == 3 and a.shape[2] == 3)):
            raise ValueError(f'Unexpected input dimensions: {a.ndim}')        # util_cylinder.py:1788
    if all(a.ndim == 2 for a in arrs):
        return torch.from_numpy(np.stack(arrs)).to(device)
    return torch.from_numpy(np.stack([np.ascontiguousarray(a) if a.ndim == 3 else np.repeat(a[..., None], 3, 2) for a in arrs])).to(device)


def line_tables(det, frame, target='cylinder'):
    """rows_updated, cols_updated of one frame of a detect_grid_batch result: the third and fourth return values of the
    reference's detect_grid (built by find_and_assign_intersections_P + clean_and_relabel, util_cylinder.py:1106-1206):
    {'points': {'row1': [(x, y), ...], ...}, 'equations': {'row1': [a2, a1, a0, lo, hi, span], ...}}"""
    ws = det['ws']
    if det.get('ws_generation', ws.generation) != ws.generation:
        raise RuntimeError('line_tables: the workspace of this result has served another detect call since (its line tables are '
                           'gone); read them before the next call or give every result its own DetectWorkspace')
    dev = det['xy'].device
    ML = _lib.MAXL
    eq = torch.empty((2, ML, 6), dtype=torch.float64, device=dev)
    npts = torch.empty((2, ML), dtype=torch.int32, device=dev)
    pts = torch.empty((2, ML, ML, 2), dtype=torch.float64, device=dev)
    nl = torch.empty(2, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().cpe_detect_line_tables(ws.view.data_ptr(), ws.bytes, ws.n, ws.h, ws.w, int(frame), eq.data_ptr(),
                                                      npts.data_ptr(), pts.data_ptr(), nl.data_ptr(),
                                                      torch.cuda.current_stream().cuda_stream), 'cpe_detect_line_tables')
    eq, npts, pts, nl = eq.cpu().numpy(), npts.cpu().numpy(), pts.cpu().numpy(), nl.cpu().numpy()
    out = []
    for sd, prefix in ((0, 'row'), (1, 'col')):
        d = {'points': {}, 'equations': {}}
        for g in range(int(nl[sd])):
            d['points'][f'{prefix}{g + 1}'] = [(float(x), float(y)) for x, y in pts[sd, g, :npts[sd, g]]]
            d['equations'][f'{prefix}{g + 1}'] = [float(v) for v in eq[sd, g]]
        out.append(d)
    return out[0], out[1]


def draw_points(gray, xy):
    """the returned picture: BGR copy of the frame with the grid points marked (deterministic; the reference draws random
    colours, util_cylinder.py:1600-1601)"""
    g = np.asarray(gray)
    col_img = g.copy() if g.ndim == 3 else np.repeat(g[..., None], 3, axis=2)
    for (x, y) in xy:
        xi, yi = int(x), int(y)
        col_img[max(yi - 2, 0):yi + 3, max(xi - 2, 0):xi + 3] = (0, 255, 0)
    return col_img


def frame_result(det, k, gray, target='cylinder'):
    """the 4-tuple detect_grid returns for frame k of a batch result, or None (after printing why) for a failed frame"""
    st = int(det['status'][k])
    if st != 0:
        print(f'Error in detect_grid: {STATUS_TEXT.get(st, st)}')
        return None
    m = int(det['n'][k])
    xy = det['xy'][k, :m].cpu().numpy(); ids = det['id'][k, :m].cpu().numpy(); center = det['center'][k].cpu().numpy()
    rows, cols = line_tables(det, k, target)
    return draw_points(gray, xy), make_json(center, xy, ids), rows, cols


def detect_grid(input_img, device='cuda:0', target='cylinder'):
    """detect_grid(input_img) -> (col_img, result_json, rows_updated, cols_updated)
    (python_grid_detection_cylinder.py:68-110; target='plane': python_grid_detection_plane.py:74-119, whose ids are
    (row, col)).  On a per-frame failure prints and returns None (:111-112)."""
    frames = frames_to_device([input_img], device)
    det = detect_grid_batch(frames, target=target)
    return frame_result(det, 0, frames[0].cpu().numpy(), target)


def grid_struct(det, k):
    """gridPts of makePyGridPts.m:39-41 for frame k: center_point (2 x 1), points (N x 4 = [x y colIdx rowIdx])"""
    m = int(det['n'][k])
    xy = det['xy'][k, :m].cpu().numpy(); ids = det['id'][k, :m].cpu().numpy().astype(np.float64)
    return dict(center_point=det['center'][k].cpu().numpy().reshape(2, 1), points=np.concatenate([xy, ids], 1).reshape(-1, 4))


def save_mat(path, grid_left=None, grid_right=None, fits=None, names=None):
    """the .mat hand-off of SURVEY 8(b): what the MATLAB pipeline holds after makePyGridPts (gridPtsPair) and
    fitSingleCylinder, written with scipy.io.savemat so `load(path)` gives the same variables.

        gridPtsPair  F x 2 struct array, fields center_point (2 x 1), points (N x 4 [x y colIdx rowIdx])
                     (makePyGridPts.m:39-41, pointsStruct2mat.m:16; column 1 = left image, 2 = right)
        frames       1 x F struct array, fields pts3 (3 x N), cylParams (2 x 6 = [cylParams0; cylParams]), cylT (4 x 4),
                     fvals (1 x 2), meanError (scalar), status        (fitSingleCylinder.m:1, fitCylinderWPts3.m:41)
        names        F x 1 cell of the image stems (getUniqueName.m)

    grid_left / grid_right: lists of dict(center_point, points) (api.grid_struct); fits: the dict
    fit.fit_single_cylinder_batch returns."""
    from scipy.io import savemat
    out = {}
    if grid_left is not None:
        F = len(grid_left)
        if grid_right is not None and len(grid_right) != F:
            raise ValueError('save_mat: left and right tables differ in length')
        pair = np.zeros((F, 2 if grid_right is not None else 1), dtype=[('center_point', 'O'), ('points', 'O')])
        for i in range(F):
            for c, g in enumerate((grid_left, grid_right) if grid_right is not None else (grid_left,)):
                pair[i, c]['center_point'] = np.asarray(g[i]['center_point'], np.float64).reshape(2, 1)
                pair[i, c]['points'] = np.asarray(g[i]['points'], np.float64).reshape(-1, 4)
        out['gridPtsPair'] = pair
    if fits is not None:
        m = fits['m'].cpu().numpy(); F = len(m)
        pts3 = fits['pts3'].cpu().numpy(); cyl = fits['cyl'].cpu().numpy(); T = fits['T'].cpu().numpy()
        fv = fits['fvals'].cpu().numpy(); me = fits['mean_err'].cpu().numpy(); st = fits['status'].cpu().numpy()
        fr = np.zeros((1, F), dtype=[(k, 'O') for k in ('pts3', 'cylParams', 'cylT', 'fvals', 'meanError', 'status')])
        for i in range(F):
            fr[0, i]['pts3'] = np.ascontiguousarray(pts3[i, :m[i]].T)
            fr[0, i]['cylParams'] = cyl[i].reshape(2, 6)
            fr[0, i]['cylT'] = T[i].reshape(4, 4)
            fr[0, i]['fvals'] = fv[i].reshape(1, 2)
            fr[0, i]['meanError'] = float(me[i])
            fr[0, i]['status'] = float(st[i])
        out['frames'] = fr
    if names is not None:
        out['names'] = np.array(list(names), dtype=object).reshape(-1, 1)
    savemat(path, out, oned_as='column')
    return path
