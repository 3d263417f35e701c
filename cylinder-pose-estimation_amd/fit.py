"""Host-side mirror of utils/fitSingleCylinder.m for a batch of stereo frames.

    [pts3, cylT, fvals, meanError] = fitSingleCylinder(i, gridPtsPair, cylRadius, ..., stereoParams, draw)

becomes fit_single_cylinder_batch(tables_left, tables_right, K1, K2, T_C2_C1, radius): all frames at
once, everything resident on the GPU, one wavefront per frame (csrc/fit.hip)."""
import ctypes as C
from dataclasses import dataclass

import torch

from . import lib as _lib

MAXP = _lib.MAXP
SEL_CHOOSE_IDX, SEL_THRESHOLD, SEL_JOIN = 0, 1, 2


@dataclass
class GridTables:
    """padded grid-point tables of n images: the reference's N x 4 [x y colIdx rowIdx] matrices"""
    xy: torch.Tensor    # f64 [n, MAXP, 2]
    id: torch.Tensor    # i32 [n, MAXP, 2]  (col, row)
    cnt: torch.Tensor   # i32 [n]

    @staticmethod
    def from_lists(mats, device):
        """mats: list of (N_i x 4) arrays [x y col row] (makePyGridPts.m:41)"""
        import numpy as np
        n = len(mats)
        xy = np.zeros((n, MAXP, 2)); ids = np.zeros((n, MAXP, 2), np.int32); cnt = np.zeros(n, np.int32)
        for i, m in enumerate(mats):
            m = np.asarray(m, dtype=np.float64).reshape(-1, 4)
            if len(m) > MAXP:
                raise ValueError(f'table {i} has {len(m)} points > MAXP={MAXP}')
            xy[i, :len(m)] = m[:, :2]; ids[i, :len(m)] = m[:, 2:4].astype(np.int32); cnt[i] = len(m)
        return GridTables(torch.from_numpy(xy).to(device), torch.from_numpy(ids).to(device),
                          torch.from_numpy(cnt).to(device))


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev3(a, device, shape):
    t = torch.as_tensor(a, dtype=torch.float64).reshape(shape).contiguous()
    return t.to(device)


def select_triangulate_batch(gp1: GridTables, gp2: GridTables, K1, K2, T21, selector=SEL_CHOOSE_IDX, patch=3, th=0.3):
    """chooseIdx / triangulateWithThreshold / findGridCorrespondences + triangulate (fitSingleCylinder.m:10-17)"""
    L = _lib.load()
    dev = gp1.xy.device
    n = gp1.cnt.shape[0]
    K1 = _dev3(K1, dev, (9,)); K2 = _dev3(K2, dev, (9,)); T21 = _dev3(T21, dev, (16,))
    ws_bytes = L.cpe_fit_workspace_bytes(n)
    ws = torch.empty(max(ws_bytes, 8), dtype=torch.uint8, device=dev)
    p1 = torch.zeros((n, MAXP, 2), dtype=torch.float64, device=dev); p2 = torch.zeros_like(p1)
    idx = torch.zeros((n, MAXP, 2), dtype=torch.int32, device=dev)
    X = torch.zeros((n, MAXP, 3), dtype=torch.float64, device=dev)
    err = torch.zeros((n, MAXP), dtype=torch.float64, device=dev)
    m = torch.zeros(n, dtype=torch.int32, device=dev); me = torch.zeros(n, dtype=torch.float64, device=dev)
    flags = torch.zeros(n, dtype=torch.int32, device=dev)
    _lib.check(L.cpe_select_triangulate_batch(gp1.xy.data_ptr(), gp1.id.data_ptr(), gp1.cnt.data_ptr(),
                                              gp2.xy.data_ptr(), gp2.id.data_ptr(), gp2.cnt.data_ptr(), n,
                                              K1.data_ptr(), K2.data_ptr(), T21.data_ptr(), selector, patch, th,
                                              ws.data_ptr(), ws_bytes, p1.data_ptr(), p2.data_ptr(), idx.data_ptr(),
                                              X.data_ptr(), err.data_ptr(), m.data_ptr(), me.data_ptr(),
                                              flags.data_ptr(), _stream()), 'cpe_select_triangulate_batch')
    return dict(p1=p1, p2=p2, idx=idx, pts3=X, err=err, m=m, mean_err=me, flags=flags, _ws=ws)


def choose_idx_batch(gp1: GridTables, gp2: GridTables, K1, K2, T21, patch=3, th=0.3):
    """[cgp1, cgp2] = chooseIdx(gp1, gp2, imgInfo, stereoParams, patch, th) (chooseIdx.m; fitSingleCylinder.m:12) for a batch:
    dict(p1, p2 f64[n,MAXP,2], idx i32[n,MAXP,2], m i32[n], flags i32[n])"""
    L = _lib.load()
    dev = gp1.xy.device
    n = gp1.cnt.shape[0]
    K1 = _dev3(K1, dev, (9,)); K2 = _dev3(K2, dev, (9,)); T21 = _dev3(T21, dev, (16,))
    ws_bytes = L.cpe_fit_workspace_bytes(n)
    ws = torch.empty(max(ws_bytes, 8), dtype=torch.uint8, device=dev)
    p1 = torch.zeros((n, MAXP, 2), dtype=torch.float64, device=dev); p2 = torch.zeros_like(p1)
    idx = torch.zeros((n, MAXP, 2), dtype=torch.int32, device=dev)
    m = torch.zeros(n, dtype=torch.int32, device=dev); flags = torch.zeros(n, dtype=torch.int32, device=dev)
    _lib.check(L.cpe_choose_idx_batch(gp1.xy.data_ptr(), gp1.id.data_ptr(), gp1.cnt.data_ptr(), gp2.xy.data_ptr(), gp2.id.data_ptr(),
                                      gp2.cnt.data_ptr(), n, K1.data_ptr(), K2.data_ptr(), T21.data_ptr(), patch, th, ws.data_ptr(),
                                      ws_bytes, p1.data_ptr(), p2.data_ptr(), idx.data_ptr(), m.data_ptr(), flags.data_ptr(), _stream()),
               'cpe_choose_idx_batch')
    return dict(p1=p1, p2=p2, idx=idx, m=m, flags=flags, _ws=ws)


def triangulate_batch(p1, p2, cnt, K1, K2, T21):
    """[worldPoints, reprojectionErrors] = triangulate(cgp1, cgp2, stereoParams) + meanError (fitSingleCylinder.m:15-17):
    p1, p2 f64[n,MAXP,2], cnt i32[n] -> dict(pts3 f64[n,MAXP,3], err f64[n,MAXP], mean_err f64[n])"""
    L = _lib.load()
    dev = p1.device
    n = cnt.shape[0]
    K1 = _dev3(K1, dev, (9,)); K2 = _dev3(K2, dev, (9,)); T21 = _dev3(T21, dev, (16,))
    X = torch.zeros((n, MAXP, 3), dtype=torch.float64, device=dev)
    err = torch.zeros((n, MAXP), dtype=torch.float64, device=dev); me = torch.zeros(n, dtype=torch.float64, device=dev)
    _lib.check(L.cpe_triangulate_batch(p1.contiguous().data_ptr(), p2.contiguous().data_ptr(), cnt.data_ptr(), n, K1.data_ptr(),
                                       K2.data_ptr(), T21.data_ptr(), X.data_ptr(), err.data_ptr(), me.data_ptr(), _stream()),
               'cpe_triangulate_batch')
    return dict(pts3=X, err=err, mean_err=me)


FIT_NELDER_MEAD, FIT_LM = 0, 1


def fit_cylinder_batch(pts3, cnt, radius, tol_x=1e-5, tol_f=1e-5, max_iter=100000, max_fun_evals=100000, mode=FIT_NELDER_MEAD):
    """fitCylinderWPts3 + applyCylParamsPrior + cylParams2T (fitSingleCylinder.m:20-25)"""
    L = _lib.load()
    dev = pts3.device
    n = cnt.shape[0]
    raw = torch.zeros((n, 2, 6), dtype=torch.float64, device=dev); cyl = torch.zeros_like(raw)
    T = torch.zeros((n, 4, 4), dtype=torch.float64, device=dev)
    fv = torch.zeros((n, 2), dtype=torch.float64, device=dev)
    it = torch.zeros((n, 2), dtype=torch.int32, device=dev); st = torch.zeros(n, dtype=torch.int32, device=dev)
    prm = _lib.CpeFitParams(tol_x, tol_f, max_iter, max_fun_evals, mode, 0)
    _lib.check(L.cpe_fit_cylinder_batch(pts3.data_ptr(), cnt.data_ptr(), n, float(radius), C.addressof(prm),
                                        raw.data_ptr(), cyl.data_ptr(), T.data_ptr(), fv.data_ptr(), it.data_ptr(),
                                        st.data_ptr(), _stream()), 'cpe_fit_cylinder_batch')
    return dict(cyl_raw=raw, cyl=cyl, T=T, fvals=fv, iters=it, status=st)


def fit_cylinder_ransac_batch(pts3, cnt, radius, hypotheses=64, sample=12, tau=0.5, seed=0, frame0=0, hyp_iters=8, tol_x=1e-5,
                              tol_f=1e-5, max_iter=100000, max_fun_evals=100000, mode=FIT_LM):
    """BUILD-DEFINED (BASELINE config 5; the reference has no RANSAC): the fit wrapped in a consensus search, see
    include/cpe.h cpe_fit_cylinder_ransac_batch.  Extra outputs: n_inliers i32[n], inlier_mask u8[n,MAXP]."""
    L = _lib.load()
    dev = pts3.device
    n = cnt.shape[0]
    raw = torch.zeros((n, 2, 6), dtype=torch.float64, device=dev); cyl = torch.zeros_like(raw)
    T = torch.zeros((n, 4, 4), dtype=torch.float64, device=dev)
    fv = torch.zeros((n, 2), dtype=torch.float64, device=dev)
    it = torch.zeros((n, 2), dtype=torch.int32, device=dev); st = torch.zeros(n, dtype=torch.int32, device=dev)
    ninl = torch.zeros(n, dtype=torch.int32, device=dev); mask = torch.zeros((n, _lib.MAXP), dtype=torch.uint8, device=dev)
    prm = _lib.CpeFitParams(tol_x, tol_f, max_iter, max_fun_evals, mode, 0)
    rp = _lib.CpeRansacParams(hypotheses, sample, tau, seed, frame0, hyp_iters, 0)
    _lib.check(L.cpe_fit_cylinder_ransac_batch(pts3.data_ptr(), cnt.data_ptr(), n, float(radius), C.addressof(prm), C.addressof(rp),
                                               raw.data_ptr(), cyl.data_ptr(), T.data_ptr(), fv.data_ptr(), it.data_ptr(),
                                               st.data_ptr(), ninl.data_ptr(), mask.data_ptr(), _stream()),
               'cpe_fit_cylinder_ransac_batch')
    return dict(cyl_raw=raw, cyl=cyl, T=T, fvals=fv, iters=it, status=st, n_inliers=ninl, inlier_mask=mask)


def fit_single_cylinder_batch(gp1: GridTables, gp2: GridTables, K1, K2, T21, radius, selector=SEL_CHOOSE_IDX,
                              patch=3, th=0.3, ransac=None, **fit_kw):
    """[pts3, cylT, fvals, meanError] = fitSingleCylinder(...) for every frame of the batch.
    ransac: None (reference behaviour) or a dict of fit_cylinder_ransac_batch keywords (build-defined config 5)."""
    sel = select_triangulate_batch(gp1, gp2, K1, K2, T21, selector, patch, th)
    if ransac is not None:
        fit = fit_cylinder_ransac_batch(sel['pts3'], sel['m'], radius, **ransac, **fit_kw)
    else:
        fit = fit_cylinder_batch(sel['pts3'], sel['m'], radius, **fit_kw)
    out = dict(sel)
    out.update(fit)
    return out
