"""Stage a-1 of the oracle against scipy.ndimage on rendered frames: cv2.GaussianBlur(5x5, sigma 0) on u8 is the binomial
(1 4 6 4 1)^2 / 256 rounded half up with BORDER_REFLECT_101; cv2.boxFilter(15x15, BORDER_REPLICATE) is a mean filter whose
running-sum order the oracle restarts every 8 columns / 4 rows (deviation 1) -- the Sauvola mask built from
scipy's uniform_filter (another summation order altogether) must come out the same but for pixels within rounding of the
threshold, of which these frames have none."""
import numpy as np
import pytest
from scipy import ndimage

import cpe_amd  # noqa: F401  (package alias)
from cpe_amd import synth
import oracle


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
    T = m * (1 + 0.5 * (np.sqrt(np.maximum(m2 - m * m, 0)) / 128 - 1))
    want_mask = 255 - (b > T).astype(np.uint8) * 255
    assert 0.2 < (mask > 0).mean() < 0.9
    assert (want_mask != mask).sum() <= 2
