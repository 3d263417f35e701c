"""The oracle's restatement of cv2.erode / dilate / morphologyEx with rectangular elements (util_cylinder.py:127, :150,
:1810-1814, :2004-2005) against scipy.ndimage -- an independent implementation of the same operators.  OpenCV's defaults:
anchor = the element's centre (k // 2, so a 20-tap window covers x - 10 .. x + 9), border pixels that fall outside the
image do not take part (erosion sees +inf there, dilation -inf)."""
import numpy as np
import pytest
from scipy import ndimage

from oracle import stages

SHAPES = [(20, 1), (1, 20), (3, 3), (5, 5), (7, 3)]          # (kw, kh): the reference's elements and two more odd ones


def _mask(seed, h=97, w=131, p=0.6):
    rng = np.random.default_rng(seed)
    m = (rng.random((h, w)) < p).astype(np.uint8) * 255
    m[30:60, 20:110] = 255                                        # a solid block so the 20-tap openings keep something
    m[:12, :] = 255; m[:, -9:] = 255                              # and pixels at the image border
    return m


@pytest.mark.parametrize('kw,kh', SHAPES)
@pytest.mark.parametrize('seed', [0, 1])
def test_rect_morphology_matches_scipy(kw, kh, seed):
    m = _mask(seed)
    ero = lambda a: ndimage.minimum_filter(a, size=(kh, kw), mode='constant', cval=255)   # window x - k//2 .. x + k - 1 - k//2
    dil = lambda a: ndimage.maximum_filter(a, size=(kh, kw), mode='constant', cval=0)     # the same window: cv2 does not reflect it
    assert np.array_equal(stages.erode_rect(m, kw, kh), ero(m))
    assert np.array_equal(stages.dilate_rect(m, kw, kh), dil(m))
    assert np.array_equal(stages.open_rect(m, kw, kh), dil(ero(m)))
    assert np.array_equal(stages.close_rect(m, kw, kh), ero(dil(m)))
    assert stages.open_rect(m, kw, kh).any()
