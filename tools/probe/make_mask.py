import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, cpe_amd
from cpe_amd import synth
from oracle import stages as S
b = synth.render_batch(1, 1200, 1920, seed=1000, with_gt=False)
g = b['left'][0].numpy()
kp, stats = S.simple_blob_detector(S.clahe(S.lab_l(g)))
ext = np.zeros_like(g)
for (x, y, sz) in kp:
    S.circle_fill(ext, int(x), int(y), int(sz / 2 + 4), 255)
with open(sys.argv[1], 'wb') as f:
    f.write(np.array(ext.shape, np.int32).tobytes()); f.write(ext.tobytes())
r = S.detect_grid(g, debug=True)
base = S.close_rect(r['roi_h'], 3, 3)
with open(sys.argv[2], 'wb') as f:
    f.write(np.array(base.shape, np.int32).tobytes()); f.write(base.tobytes())
