"""what the threshold sweep of the blob stage works on, per frame of the bench workload: the working rectangle, the pixels
per grey-level bucket (= pixels that join a forest per threshold), the component lists per threshold, the followed borders.
    python tools/sweep_stats.py [frames, default 4]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cpe_amd
from cpe_amd import synth, api

NTHR, NBK = 17, 18
SW_NH = 8; SW_NL = SW_NH + NTHR; SW_NB = SW_NL + NTHR; SW_ND = SW_NB + NTHR; SW_BS = SW_ND + NTHR
SW_BO = SW_BS + NBK; SW_BC = SW_BO + NBK; SW_NT = SW_BC + NBK; SW_NC = SW_NT + NTHR; SW_NA = SW_NC + NTHR

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
b = synth.render_batch(n, 1200, 1920, seed=1000, device='cuda', with_gt=False)
frames = torch.cat([b['left'], b['right']])
ws = api.DetectWorkspace(2 * n, 1200, 1920, frames.device)
det = api.detect_grid_batch(frames, ws)
torch.cuda.synchronize()
sw = ws.plane('sweep').cpu().numpy()
st = ws.state()
cl = ws.plane('clahe').cpu().numpy()
np.set_printoptions(linewidth=250)
for f in range(2 * n):
    S = sw[f]
    r = [st[f][k] for k in ('crect0', 'crect1', 'crect2', 'crect3')]
    area = (r[2] - r[0] + 1) * (r[3] - r[1] + 1)
    print(f'frame {f}: status {st[f]["status"]} crect {r} = {area} px ({100 * area / (1200 * 1920):.0f} % of the frame), '
          f'blobs {st[f]["n_blobs"]} groups {st[f]["n_groups"]}')
    print('  bucket px   ', S[SW_BS:SW_BS + NBK], 'sum', int(S[SW_BS + 1:SW_BS + NBK].sum()))
    print('  dark lists  ', S[SW_NH:SW_NH + NTHR], 'sum', int(S[SW_NH:SW_NH + NTHR].sum()))
    print('  bright lists', S[SW_NL:SW_NL + NTHR], 'sum', int(S[SW_NL:SW_NL + NTHR].sum()))
    print('  traced holes', S[SW_NT:SW_NT + NTHR], 'sum', int(S[SW_NT:SW_NT + NTHR].sum()))
    print('  blobs       ', S[SW_NB:SW_NB + NTHR], 'from holes', S[SW_NA:SW_NA + NTHR])
    print('  border dists', S[SW_ND:SW_ND + NTHR], 'chunks', S[SW_NC:SW_NC + NTHR])
    c = cl[f][r[1]:r[3] + 1, r[0]:r[2] + 1]
    print('  CLAHE levels in the rectangle: <=50: %d, >210: %d' % ((c <= 50).sum(), (c > 210).sum()))
