"""The oracle against golden vectors produced by the REAL reference functions
(tools/gen_golden.py, generated in the authoring container from /root/reference)."""
import json
import os

import numpy as np

from conftest import GOLDEN


def test_ridges_bit_identical_to_skimage(orc):
    """util_cylinder.detect_ridges (real skimage/scipy) vs oracle: bit-for-bit."""
    z = np.load(os.path.join(GOLDEN, 'ridges.npz'))
    for k in 'abcn':
        emax, emin = orc.detect_ridges(z['img_' + k])
        assert np.array_equal(emin, z['emin_' + k]), k
        assert np.array_equal(emax, z['emax_' + k]), k
