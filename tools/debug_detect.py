import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cpe_amd
from cpe_amd import synth, api
from oracle import stages as S
h, w, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
b = synth.render_batch(2, h, w, seed=seed, with_gt=False)
frames = torch.cat([b['left'], b['right']])
det = api.detect_grid_batch(frames.cuda())
torch.cuda.synchronize()
st = det['ws'].state()
for i in range(frames.shape[0]):
    ref = S.detect_grid(frames[i].numpy())
    print(i, 'gpu n', int(det['n'][i]), 'status', int(det['status'][i]), 'ref n', len(ref['xy']), 'ref status', ref['status'], 'rows/cols ref', ref['n_rows'], ref['n_cols'])
    print('   ', {k: v for k, v in st[i].items() if k in ('status','n_rows','n_cols','n_joints','overflow','n_kp','n_groups','crect0','crect1','crect2','crect3','nrect0','nrect2','n_roots','dbg_max_fg','dbg_max_hole','dbg_sum_steps')})
