"""which capacity does a frame exceed?  python tools/debug_overflow.py  (GPU box): the noise frame of
tests/test_detect_gpu.py::test_detect_failure_statuses"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cpe_amd
from cpe_amd import api
from oracle import stages as S
rng = np.random.default_rng(3)
dark = np.full((480, 640), 7, np.uint8)
noise = rng.integers(0, 60, size=(480, 640), dtype=np.uint8)
det = api.detect_grid_batch(torch.from_numpy(np.stack([dark, noise])).cuda())
torch.cuda.synchronize()
st = det['ws'].state()
for i in range(2):
    print(i, int(det['status'][i]), {k: v for k, v in st[i].items() if k in ('overflow', 'n_joints', 'n_joints_all', 'n_kp', 'n_blobs', 'n_groups', 'n_roots', 'n_comps', 'status')})
ref = S.detect_grid(noise, debug=True)
print('oracle', ref['status'], ref['n_joints'], ref['n_cyl_joints'], ref['n_keypoints'], ref['rect'])
