"""Stage a-1 of the oracle against independent evaluations (scipy.ndimage, a literal numpy restatement of OpenCV's box filter).

cv2.GaussianBlur(5x5, sigma 0) on u8 is the binomial (1 4 6 4 1)^2 / 256 rounded half up with BORDER_REFLECT_101.
cv2.boxFilter(15x15, normalize, BORDER_REPLICATE) on f64 = RowSum (one running sum along each whole row) then ColumnSum (one
running sum down each whole column).  The oracle runs the column sums exactly so; its row sums restart every ORC_BOX_BX = 8
columns (deviation 1: a whole-row running sum has no parallel evaluation).  `opencv_box_literal` below
is the literal form, implementation-independent (no blocks at all): the two must give the same Sauvola mask except for
pixels whose eigenvalue lies within rounding of the threshold; the tests count them on clean, noisy and ramp frames."""
import numpy as np
import pytest
from scipy import ndimage

import cpe_amd  # noqa: F401  (package alias)
from cpe_amd import synth
import oracle


def opencv_box_literal(a, k=15):
    """cv2.boxFilter(a, CV_64F, (k, k), normalize=True, borderType=BORDER_REPLICATE) as OpenCV 4.5.5 box_filter.simd.hpp runs
    it for double input: RowSum<double,double> (s = sum of the first k, then s += S[i + k] - S[i] along the whole row),
    ColumnSum<double,double> (SUM = first k - 1 rows added one by one to 0; per row s0 = SUM + Sp, D = s0 * scale,
    SUM = s0 - Sm).  numpy: the sequential recurrences run along one axis, vectorised over the other."""
    r = k // 2
    h, w = a.shape
    p = np.pad(a, r, mode='edge')                       # BORDER_REPLICATE
    rs = np.empty((h + 2 * r, w))
    s = np.zeros(h + 2 * r)
    for i in range(k):
        s = s + p[:, i]
    rs[:, 0] = s
    for x in range(w - 1):
        s = s + (p[:, x + k] - p[:, x])
        rs[:, x + 1] = s
    out = np.empty((h, w))
    SUM = np.zeros(w)
    for j in range(k - 1):
        SUM = SUM + rs[j]
    scale = 1.0 / (k * k)
    for y in range(h):
        s0 = SUM + rs[y + k - 1]
        out[y] = s0 * scale
        SUM = s0 - rs[y]
    return out


def sauvola_from_means(b, m, m2):
    var = np.maximum(m2 - m * m, 0)
    T = m * (1 + 0.5 * (np.sqrt(var) / 128 - 1))
    return (255 - (b > T).astype(np.uint8) * 255).astype(np.uint8)


@pytest.mark.parametrize('h,w,seed', [(480, 640, 5), (600, 960, 9)])
def test_blur_and_sauvola_match_scipy(h, w, seed):
    g = synth.render_batch(1, h, w, seed=seed, device='cpu', with_gt=False)['left'][0].numpy()
    k = np.array([1, 4, 6, 4, 1], float)
    want = np.floor(ndimage.correlate(g.astype(float), np.outer(k, k), mode='mirror') / 256 + 0.5).astype(np.uint8)
    blurred, mask, b = oracle.preprocess(g, want_b=True)
    assert np.array_equal(blurred, want)
    G = ndimage.gaussian_filter(blurred.astype(float) / 255, 3.0, mode='constant', cval=0, truncate=4.0)
    assert np.abs(G - oracle.gauss_sigma3(blurred)).max() < 1e-15          # bit-level pin: tests/golden/ridges.npz
    m = ndimage.uniform_filter(b, 15, mode='nearest'); m2 = ndimage.uniform_filter(b * b, 15, mode='nearest')
    want_mask = sauvola_from_means(b, m, m2)
    assert 0.2 < (mask > 0).mean() < 0.9
    assert (want_mask != mask).sum() <= 2


def _frames():
    """clean render, the same with sensor noise, an intensity ramp with noise, and a frame with flat saturated / black areas
    (where b, its means and the threshold are all exactly or nearly zero: the pixels a summation order can flip)"""
    rng = np.random.default_rng(11)
    g = synth.render_batch(1, 300, 420, seed=3, device='cpu', with_gt=False)['left'][0].numpy()
    noisy = np.clip(g.astype(int) + rng.integers(-9, 10, g.shape), 0, 255).astype(np.uint8)
    ramp = np.clip(np.linspace(0, 255, g.shape[1])[None, :] + rng.integers(-3, 4, g.shape), 0, 255).astype(np.uint8)
    flat = g.copy(); flat[:120, :150] = 255; flat[200:, 250:] = 0
    return dict(clean=g, noisy=noisy, ramp=ramp, flat=flat)


def test_mask_against_the_literal_opencv_box_filter():
    """the oracle's box sums (columns: OpenCV's own order; rows: restarted every 8 columns) against the literal whole-row /
    whole-column running sums.  Measured on these frames: no pixel differs on the clean, noisy and ramp frames; on the frame
    with perfectly flat areas every differing pixel has b == 0 exactly and a threshold of rounding-residue size
    (|T| < 1e-16; nearly all of them have an all-zero 15 x 15 box, whose true mean is exactly 0): what a running sum leaves
    behind after the non-zero values have been subtracted again (+-1e-18, sign by history) decides the bit, i.e. the mask
    is undefined there in OpenCV itself (it would change with its stripe / SIMD configuration).  Everywhere else the two
    evaluations must give the same bit."""
    undefined = 0
    for name, g in _frames().items():
        blurred, mask, b = oracle.preprocess(g, want_b=True)
        m = opencv_box_literal(b); m2 = opencv_box_literal(b * b)
        lit = sauvola_from_means(b, m, m2)
        diff = lit != mask
        if name != 'flat':
            assert not diff.any(), f'{name}: {diff.sum()} mask pixels differ'
            continue
        T = m * (1 + 0.5 * (np.sqrt(np.maximum(m2 - m * m, 0)) / 128 - 1))
        assert (b[diff] == 0).all() and np.abs(T[diff]).max() < 1e-16
        flat_box = ndimage.maximum_filter(np.abs(b), size=15, mode='nearest') == 0      # all 225 values of the box are 0
        assert (diff & ~flat_box).sum() <= 16                # the rim of the flat areas, where the residue has not died out yet
        undefined += int(diff.sum())
    assert undefined > 0        # the flat frame does exercise the undefined case (else this test checks less than it says)


def test_column_sums_are_opencvs_own():
    """with row sums taken out of the comparison (a 1-column-wide box has none to restart: k x 1 would need another entry
    point, so feed an image that is constant along x): the oracle's mean equals the literal ColumnSum bit for bit"""
    rng = np.random.default_rng(2)
    col = rng.normal(size=(200, 1)) * 1e-3
    b = np.repeat(col, 64, axis=1)
    # constant along x: every row sum is 15 * b (exactly the same value in both evaluations: 15 equal addends), so any
    # difference of the masks / means would come from the column pass
    lit = opencv_box_literal(b)
    mask = oracle.sauvola_mask(b)
    want = sauvola_from_means(b, lit, opencv_box_literal(b * b))
    assert np.array_equal(mask, want)
