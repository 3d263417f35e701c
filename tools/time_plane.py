"""row f-2: time the planar-target detector on near-planar synthetic frames (python tools/time_plane.py [stereo frames])"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cpe_amd
from cpe_amd import synth, api
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sc = synth.Scene(h=1200, w=1920, radius=5000.0, depth=(5340.0, 5400.0), tilt_deg=4.0)
b = synth.render_batch(n, 1200, 1920, seed=1, device='cuda', scene=sc, with_gt=False)
frames = torch.cat([b['left'], b['right']])
ws = api.DetectWorkspace(2 * n, 1200, 1920, frames.device)
for it in range(3):
    torch.cuda.synchronize(); t = time.time()
    det = api.detect_grid_batch(frames, ws, target='plane')
    torch.cuda.synchronize(); dt = time.time() - t
    print(f'iter {it}: planar detect {2*n} images {1e3*dt:.1f} ms ({1e3*dt/(2*n):.2f} ms/img), ok {(det["status"] == 0).float().mean().item():.2f}, points/img {det["n"].float().mean().item():.0f}')
os.environ['CPE_SERIAL'] = '1'
cpe_amd.lib.profile(True)
det = api.detect_grid_batch(frames, ws, target='plane'); torch.cuda.synchronize()
rep = cpe_amd.lib.profile_report(); cpe_amd.lib.profile(False)
print('event profile (one stream):', ' | '.join(f"{r[0].split('::')[-1]} x{r[1]} {r[2]:.1f}ms" for r in rep[:10]), '| total %.1f ms' % sum(r[2] for r in rep))
