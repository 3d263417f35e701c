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


def _nested_mask(rng, h, w, k):
    """rings (square and round) with dots and blocks inside them, some nested several levels deep, plus specks"""
    m = np.zeros((h, w), np.uint8)
    yy, xx = np.ogrid[:h, :w]
    for _ in range(k):
        cy, cx = int(rng.integers(8, h - 8)), int(rng.integers(8, w - 8))
        r = int(rng.integers(3, 40))
        d = np.maximum(np.abs(yy - cy), np.abs(xx - cx)) if rng.random() < 0.5 else np.hypot(yy - cy, xx - cx)
        m[(d <= r) & (d >= r - int(rng.integers(1, 3)))] = 255
        if rng.random() < 0.7:
            m[cy, cx] = 255
        if rng.random() < 0.3:
            m[max(cy - 1, 0):cy + 2, max(cx - 1, 0):cx + 2] = 255
    m[rng.random((h, w)) < 0.01] = 255
    return m


def _serpentine(h, w):
    """a closed box whose inside is a long meander of walls: the outer background is the frame around the box, everything inside
    is enclosed; the specks between the walls are nested.  A gap in the box lets the outside in: then the meander has to be
    flooded turn by turn (many sweeps)."""
    m = np.zeros((h, w), np.uint8)
    m[4, 4:w - 4] = 255; m[h - 5, 4:w - 4] = 255; m[4:h - 4, 4] = 255; m[4:h - 4, w - 5] = 255
    for k, y in enumerate(range(10, h - 10, 6)):
        if k % 2 == 0:
            m[y, 4:w - 12] = 255
        else:
            m[y, 12:w - 4] = 255
        m[y + 3, 20 + 7 * (k % 5)] = 255          # a speck in every corridor
    return m


@pytest.mark.gpu
def test_retr_external_rule_matches_literal_scanner(cpe, orc, gpu):
    """components inside a hole of another one are dropped exactly as cv2.findContours(RETR_EXTERNAL) drops them: the flood-based
    rule on the GPU vs the oracle's literal Suzuki-Abe scanner (icvFindNextContour's lnbd test)"""
    import ctypes as C
    from oracle import stages as S
    rng = np.random.default_rng(11)
    L = cpe.lib.load()
    cases = []
    for (h, w, n) in ((64, 64, 3), (97, 650, 2), (200, 1920, 2), (130, 257, 3)):
        frames = [_nested_mask(rng, h, w, int(rng.integers(2, 40))) for _ in range(n)]
        frames[0] = ((rng.random((h, w)) < 0.5) * 255).astype(np.uint8)            # dense noise: holes inside holes everywhere
        cases.append(np.stack(frames))
    closed = _serpentine(120, 200)
    opened = closed.copy(); opened[4, 100:103] = 0                                    # a door in the top wall
    cases.append(np.stack([closed, opened, np.zeros_like(closed), np.full_like(closed, 255)]))
    for batch in cases:
        n, h, w = batch.shape
        ws = cpe.api.DetectWorkspace(n, h, w, gpu)
        cap = 1 << 16
        first = torch.full((n, cap), -1, dtype=torch.int32, device=gpu)
        cnt = torch.zeros(n, dtype=torch.int32, device=gpu)
        d = torch.from_numpy(batch).to(gpu)
        cpe.lib.check(L.cpe_debug_external_components(d.data_ptr(), n, h, w, ws.view.data_ptr(), ws.bytes, first.data_ptr(), cap,
                                                      cnt.data_ptr(), torch.cuda.current_stream().cuda_stream), 'cpe_debug_external_components')
        torch.cuda.synchronize()
        for i in range(n):
            want = sorted(int(p[0][1]) * w + int(p[0][0]) for p, hole in S.find_contours(batch[i], 'external', 'simple'))
            got = sorted(first[i, :int(cnt[i])].cpu().tolist())
            assert got == want, (h, w, i, len(got), len(want))
            nall = len({int(p[0][1]) * w + int(p[0][0]) for p, hole in S.find_contours(batch[i], 'list', 'simple') if not hole})
            if i == 0 and n == 4 and h == 120:
                assert len(want) == 1 and nall > 10          # closed box: only the box itself is external
            if i == 1 and n == 4 and h == 120:
                assert len(want) == nall                     # door open: nothing is enclosed any more
