"""timing aid for the tile pass of the threshold sweep: k_sw_tile with CPE_SW_DBG = 0..5 (results are wrong for dbg != 0)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cpe_amd
from cpe_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
os.environ['CPE_SERIAL'] = '1'
b = synth.render_batch(n // 2, 1200, 1920, seed=1000, device='cuda', with_gt=False)
frames = torch.cat([b['left'], b['right']])
ws = api.DetectWorkspace(n, 1200, 1920, frames.device)
for dbg in (0, 4, 3):
    os.environ['CPE_SW_DBG'] = str(dbg)
    if dbg == 9:
        os.environ['CPE_SW_DBG'] = '4'
        frames = frames[:, ::-1].contiguous() if False else torch.zeros_like(frames) + 30
        frames[:, 100:900, 900:1400] = 200
    api.detect_grid_batch(frames, ws); torch.cuda.synchronize()
    cpe_amd.lib.profile(True)
    det = api.detect_grid_batch(frames, ws); torch.cuda.synchronize()
    rep = {r[0].split('::')[-1]: r for r in cpe_amd.lib.profile_report()}
    cpe_amd.lib.profile(False)
    st = det['ws'].state()
    print('dbg', dbg, {k: round(v[2], 2) for k, v in rep.items() if 'k_sw_' in k or 'k_hole' in k}, 'crect', [st[0][f'crect{i}'] for i in range(4)], flush=True)
