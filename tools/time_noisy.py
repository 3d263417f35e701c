"""what a frame with heavy sensor noise costs: the 10 frames of test_heavy_sensor_noise_at_full_size_is_not_a_capacity_overflow
(1920x1200, +-7 .. +-11 DN of uniform noise: up to 24 000 blobs per threshold, 22 000 blob groups), per-kernel event times"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cpe_amd
from cpe_amd import api, synth

dev = torch.device('cuda:0')
b = synth.render_batch(5, 1200, 1920, seed=77, device='cuda', with_gt=False)
fr = torch.cat([b['left'], b['right']]).cpu().numpy().astype(np.int32)
rng = np.random.default_rng(5)
for i in range(fr.shape[0]):
    a = 7 + i % 5
    fr[i] += rng.integers(-a, a + 1, size=fr[i].shape)
frames = torch.from_numpy(np.clip(fr, 0, 255).astype(np.uint8)).to(dev)
ws = api.DetectWorkspace(frames.shape[0], 1200, 1920, dev)
for it in range(2):
    torch.cuda.synchronize(); t = time.time()
    det = api.detect_grid_batch(frames, ws)
    torch.cuda.synchronize()
    print(f'detect of {frames.shape[0]} noisy frames: {1e3 * (time.time() - t):.1f} ms')
st = ws.state()
print('status', det['status'].tolist(), 'groups', [d['n_groups'] for d in st])
cpe_amd.lib.profile(True)
det = api.detect_grid_batch(frames, ws); torch.cuda.synchronize()
rep = cpe_amd.lib.profile_report(); cpe_amd.lib.profile(False)
print('event profile:', ' | '.join(f"{r[0].split('::')[-1]} x{r[1]} {r[2]:.1f}ms" for r in rep[:12]), '| total %.1f ms' % sum(r[2] for r in rep))
