"""The labelling pass on its own (cpe_debug_ccl, include/cpe.h): labels of {(img > thr) != invert} against
scipy.ndimage.label, which is what cv2.connectedComponents / the component front end of cv2.findContours compute
(util_cylinder.py:28,161,1817,1883,1968).  Widths that are multiples of 16 take the word-level kernels
(k_ccl_merge64), the others the byte-level ones."""
import ctypes as C

import numpy as np
import pytest
import torch
from scipy import ndimage


def _masks(rng, n, h, w):
    """grey frames with blobs, long thin lines, single-pixel noise and empty / full rows"""
    f = rng.integers(0, 256, size=(n, h, w), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    for i in range(n):
        g = f[i]
        g[(yy // 7 + xx // 11 + i) % 3 == 0] //= 3            # blocks of mostly-dark
        g[h // 3, :] = 255                                      # a full row: runs that cross every 64-pixel word
        g[:, w // 2] = 255
        g[(yy == xx % h)] = 250                                 # a diagonal (8- but not 4-connected)
        g[2 * h // 3:2 * h // 3 + 3, 5:w - 5] = 0
    f[n - 1] = 0
    f[n - 1, 1::2, 1::2] = 255                                  # isolated pixels everywhere
    return f


def _expected(mask, conn8):
    lab, k = ndimage.label(mask, structure=np.ones((3, 3)) if conn8 else None)
    idx = np.arange(mask.size, dtype=np.int64).reshape(mask.shape)
    out = np.full(mask.shape, -1, dtype=np.int64)
    if k:
        first = ndimage.minimum(idx, lab, index=np.arange(1, k + 1)).astype(np.int64)
        out[lab > 0] = first[lab[lab > 0] - 1]
    return out


@pytest.mark.gpu
@pytest.mark.parametrize('h,w', [(96, 192), (130, 201), (72, 1040), (64, 64)])
@pytest.mark.parametrize('thr,invert,conn8', [(128, 0, 1), (40, 1, 0), (200, 0, 1), (0, 0, 1), (254, 1, 0)])
def test_labels_match_scipy(cpe, gpu, h, w, thr, invert, conn8):
    from cpe_amd import api
    rng = np.random.default_rng(h * 1000 + w + thr)
    frames = _masks(rng, 3, h, w)
    g = torch.from_numpy(frames).to(gpu)
    n = g.shape[0]
    ws = api.DetectWorkspace(n, h, w, gpu)
    L = cpe.lib.load()
    cpe.lib.check(L.cpe_debug_ccl(g.data_ptr(), n, h, w, thr, invert, conn8, 0, 0, 1, ws.view.data_ptr(), ws.bytes,
                                  torch.cuda.current_stream().cuda_stream), 'cpe_debug_ccl')
    torch.cuda.synchronize()
    got = ws.plane('labels').cpu().numpy()
    for i in range(n):
        mask = (frames[i] > thr) != bool(invert)
        want = _expected(mask, conn8)
        assert np.array_equal(got[i][mask], want[mask]), (i, int((got[i][mask] != want[mask]).sum()))
        assert (got[i][~mask] == -1).all()
