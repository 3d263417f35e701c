"""OpenCV pieces of the oracle that have a second implementation in numpy / scipy (cv2 itself is not installed here, so these
are checks of the restated ALGORITHM, not of OpenCV's bits):
  * findContours(RETR_EXTERNAL) on masks without nesting: one contour per 8-connected component (scipy.ndimage.label), each
    starting at its component's raster-first pixel; RETR_LIST adds one hole border per enclosed background component;
  * contour moments / contourArea: Green's theorem on the vertex polygon (exact in f64 for pixel coordinates);
  * convexHull: the vertex set of scipy.spatial.ConvexHull (Qhull);
  * fillPoly of a convex polygon: the pixels whose centre lies inside or on it;
  * GaussianBlur 19x19 / 7x7 (sigma from the kernel size, u8 fixed point): within 1 DN of the float filter."""
import numpy as np
import pytest
from scipy import ndimage
from scipy.spatial import ConvexHull

from oracle import stages


def _blobs(seed, h=120, w=160, n=14):
    rng = np.random.default_rng(seed)
    m = np.zeros((h, w), np.uint8)
    yy, xx = np.mgrid[:h, :w]
    for _ in range(n):
        cy, cx = rng.integers(8, h - 8), rng.integers(8, w - 8)
        a, b = rng.integers(2, 12), rng.integers(2, 9)
        m[((yy - cy) / a) ** 2 + ((xx - cx) / b) ** 2 <= 1] = 255
    return m


@pytest.mark.parametrize('seed', range(4))
def test_external_contours_are_the_8_connected_components(seed):
    m = _blobs(seed)
    m = ndimage.binary_fill_holes(m).astype(np.uint8) * 255                # no holes: nothing can be nested
    lab, k = ndimage.label(m, structure=np.ones((3, 3)))
    cs = stages.find_contours(m, 'external', 'none')
    assert len(cs) == k and not any(hole for _, hole in cs)
    firsts = sorted(int(np.flatnonzero(lab.ravel() == i)[0]) for i in range(1, k + 1))
    assert sorted(int(p[0, 1]) * m.shape[1] + int(p[0, 0]) for p, _ in cs) == firsts
    for p, _ in cs:                                                        # a border: every point set, 8-adjacent steps, closed
        assert (m[p[:, 1], p[:, 0]] > 0).all()
        d = np.abs(np.diff(np.r_[p, p[:1]], axis=0)).max(1)
        assert (d <= 1).all()


@pytest.mark.parametrize('seed', range(4))
def test_list_mode_adds_one_border_per_hole(seed):
    m = _blobs(seed, n=20)
    m[40:80, 50:110] = 255; m[50:60, 60:70] = 0; m[65:72, 90:100] = 0      # a block with two holes
    lab, k = ndimage.label(m, structure=np.ones((3, 3)))
    bg, kb = ndimage.label(m == 0)                                         # background: 4-connected
    edge = set(np.unique(np.r_[bg[0], bg[-1], bg[:, 0], bg[:, -1]])) - {0}
    holes = kb - len(edge)
    cs = stages.find_contours(m, 'list', 'none')
    assert sum(1 for _, hole in cs if not hole) == k
    assert sum(1 for _, hole in cs if hole) == holes >= 2


@pytest.mark.parametrize('seed', range(6))
def test_moments_area_and_hull(seed):
    rng = np.random.default_rng(seed)
    m = _blobs(seed, n=6)
    for p, _ in stages.find_contours(m, 'external', 'simple'):
        x, y = p[:, 0].astype(float), p[:, 1].astype(float)
        xn, yn = np.roll(x, -1), np.roll(y, -1)
        cr = x * yn - xn * y
        a = cr.sum() / 2
        m00, m10, m01 = stages.contour_moments(p)
        assert abs(stages.contour_area(p) - abs(a)) < 1e-9
        if a != 0:
            assert abs(m00 - abs(a)) < 1e-9
            assert abs(m10 / m00 - ((x + xn) * cr).sum() / (6 * a)) < 1e-9
            assert abs(m01 / m00 - ((y + yn) * cr).sum() / (6 * a)) < 1e-9
    pts = rng.integers(0, 200, (60, 2)).astype(np.int32)
    hull = stages.convex_hull(pts)
    want = pts[ConvexHull(pts.astype(float)).vertices]
    assert {tuple(v) for v in hull} == {tuple(v) for v in want}
    img = stages.fill_poly((200, 200), hull)                                # convex: inside or on the boundary
    yy, xx = np.mgrid[:200, :200]
    h2 = np.array(sorted({tuple(v) for v in want}), float)
    q = ConvexHull(h2)
    inside = np.all(q.equations[:, :2] @ np.stack([xx.ravel(), yy.ravel()]).astype(float) + q.equations[:, 2:3] <= 1e-9, axis=0).reshape(200, 200)
    strict = np.all(q.equations[:, :2] @ np.stack([xx.ravel(), yy.ravel()]).astype(float) + q.equations[:, 2:3] <= -0.75, axis=0).reshape(200, 200)
    assert (img[strict] == 255).all()                                       # everything well inside is filled
    assert not (img[~ndimage.binary_dilation(inside, iterations=1)] == 255).any()   # nothing a pixel beyond the hull


@pytest.mark.parametrize('k,fn', [(19, stages.blur19), (7, stages.blur7)])
def test_fixed_point_gaussians_track_the_float_filter(k, fn):
    rng = np.random.default_rng(k)
    g = (ndimage.gaussian_filter(rng.random((90, 140)), 2.0) * 900 % 256).astype(np.uint8)
    if k == 7:      # cv2.getGaussianKernel(7, sigma <= 0): the built-in small table, not the exponential
        t, tol = np.array([0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]), 0.5 + 1e-9
    else:           # sigma = 0.3 ((k - 1) / 2 - 1) + 0.8 = 3.2; the u8 path rounds the taps to 8 fractional bits
        sigma = 0.3 * ((k - 1) * 0.5 - 1) + 0.8
        t = np.exp(-(np.arange(k) - k // 2) ** 2 / (2 * sigma * sigma)); t /= t.sum()
        tol = 1.5
    want = ndimage.correlate1d(ndimage.correlate1d(g.astype(float), t, axis=1, mode='mirror'), t, axis=0, mode='mirror')
    got = fn(g).astype(float)
    assert np.abs(got - want).max() <= tol and np.abs(got - want).mean() < 0.3
