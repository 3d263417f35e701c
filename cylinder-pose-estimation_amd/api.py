"""Python face of the image half of the hot path.

`detect_grid_batch` is the MI355X-native form of python_grid_detection_cylinder.py::detect_grid
(:68-112): all frames of a batch go through the HIP kernels behind the C ABI (include/cpe.h) and come
back as padded point tables; `detect_grid(input_img)` keeps the reference's single-image signature and
return shape, `make_json` its JSON (util_cylinder.py:1674-1727, decoded by makePyGridPts.m:39-41).
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
                 'overflow', 'hull_n', 'crect0', 'crect1', 'crect2', 'crect3', 'nrect0', 'nrect1', 'nrect2', 'nrect3', 'n_roots_p', 'n_roots_s', 'spot_fail']


class DetectWorkspace:
    """device scratch for cpe_detect_grid_batch, reusable across calls with the same (n, h, w)"""

    def __init__(self, n, h, w, device):
        self.n, self.h, self.w = n, h, w
        L = _lib.load()
        self.bytes = L.cpe_detect_workspace_bytes(n, h, w)
        self.buf = torch.empty(self.bytes + 256, dtype=torch.uint8, device=device)
        off = (-self.buf.data_ptr()) % 256
        self.view = self.buf[off:off + self.bytes]

    def plane(self, name):
        """intermediate of the last call: u8 [n,h,w] planes, i32 [n,4096,2] joints, or the state records"""
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
    """frames: u8 tensor [n,h,w] on the GPU -> dict(xy f64[n,MAXP,2], id i32[n,MAXP,2], n i32[n], center f64[n,2],
    status i32[n], ws).  target='plane': the planar-target script (python_grid_detection_plane.py, row f-2); ids are
    (row, col) there."""
    if not (isinstance(frames, torch.Tensor) and frames.is_cuda and frames.dtype == torch.uint8 and frames.dim() == 3):
        raise TypeError('frames must be a CUDA uint8 tensor [n,h,w]')
    frames = frames.contiguous()
    n, h, w = frames.shape
    dev = frames.device
    L = _lib.load()
    if ws is None or (ws.n, ws.h, ws.w) != (n, h, w):
        ws = DetectWorkspace(n, h, w, dev)
    xy = torch.zeros((n, MAXP, 2), dtype=torch.float64, device=dev)
    ids = torch.zeros((n, MAXP, 2), dtype=torch.int32, device=dev)
    cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    center = torch.zeros((n, 2), dtype=torch.float64, device=dev)
    status = torch.zeros(n, dtype=torch.int32, device=dev)
    prm = _lib.CpeDetectParams(1 if subpixel else 0, subpixel_window, subpixel_step, TARGETS[target], 0)
    _lib.check(L.cpe_detect_grid_batch_ex(frames.data_ptr(), n, h, w, C.addressof(prm), ws.view.data_ptr(), ws.bytes,
                                          xy.data_ptr(), ids.data_ptr(), cnt.data_ptr(), center.data_ptr(),
                                          status.data_ptr(), torch.cuda.current_stream().cuda_stream),
               'cpe_detect_grid_batch_ex')
    return dict(xy=xy, id=ids, n=cnt, center=center, status=status, ws=ws)


def tables_of(det):
    """detect_grid_batch output -> GridTables (the N x 4 [x y col row] matrices of makePyGridPts.m:41)"""
    return GridTables(det['xy'], det['id'], det['n'])


def make_json(center, xy, ids):
    """the JSON string make_json returns (util_cylinder.py:1674-1727): indent 4, keys id/x/y in this order"""
    pts = [{"id": [int(c), int(r)], "x": float(x), "y": float(y)} for (x, y), (c, r) in zip(xy, ids)]
    return json.dumps({"center_point": [float(center[0]), float(center[1])], "points": pts}, indent=4, ensure_ascii=False)


def to_gray(input_img):
    """2-D grey array, or H x W x 3 BGR with identical channels (what cv2.imread gives for a mono camera)"""
    a = np.asarray(input_img)
    if a.dtype != np.uint8:
        raise TypeError('detect_grid expects uint8 images')
    if a.ndim == 2:
        return a
    if a.ndim == 3 and a.shape[2] == 3:
        if not (np.array_equal(a[..., 0], a[..., 1]) and np.array_equal(a[..., 1], a[..., 2])):
            raise NotImplementedError('colour frames are outside the scope of this build (grey / grey-replicated only)')
        return np.ascontiguousarray(a[..., 0])
    raise ValueError(f'Unexpected input dimensions: {a.ndim}')


def detect_grid(input_img, device='cuda:0', target='cylinder'):
    """detect_grid(input_img) -> (col_img, result_json, rows_updated, cols_updated)
    (python_grid_detection_cylinder.py:68-110; target='plane': python_grid_detection_plane.py:74-119, whose ids are
    (row, col)).  On a per-frame failure prints and returns None (:111-112)."""
    gray = to_gray(input_img)
    det = detect_grid_batch(torch.from_numpy(gray).to(device)[None], target=target)
    st = int(det['status'][0])
    if st != 0:
        print(f'Error in detect_grid: {STATUS_TEXT.get(st, st)}')
        return None
    m = int(det['n'][0])
    xy = det['xy'][0, :m].cpu().numpy(); ids = det['id'][0, :m].cpu().numpy(); center = det['center'][0].cpu().numpy()
    col_img = np.repeat(gray[..., None], 3, axis=2)
    for (x, y) in xy:                       # deterministic drawing (the reference's colours are random)
        xi, yi = int(x), int(y)
        col_img[max(yi - 2, 0):yi + 3, max(xi - 2, 0):xi + 3] = (0, 255, 0)
    cols = {}
    rows = {}
    for (x, y), (i0, i1) in zip(xy, ids):
        c, r = (i1, i0) if target == 'plane' else (i0, i1)
        cols.setdefault(f'col{int(c) + 1}', []).append((float(x), float(y)))
        rows.setdefault(f'row{int(r)}', []).append((float(x), float(y)))
    return col_img, make_json(center, xy, ids), {'points': rows, 'equations': {}}, {'points': cols, 'equations': {}}
