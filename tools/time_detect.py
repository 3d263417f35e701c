"""quick timing of the detect + fit pipeline (run under rocprofv3 --stats for the per-kernel table)"""
import sys, time
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cpe_amd
from cpe_amd import synth, api, fit

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
b = synth.render_batch(n, 1200, 1920, seed=1, device='cuda', with_gt=False)
frames = torch.cat([b['left'], b['right']])
ws = api.DetectWorkspace(2 * n, 1200, 1920, frames.device)
for it in range(3):
    torch.cuda.synchronize(); t = time.time()
    det = api.detect_grid_batch(frames, ws)
    torch.cuda.synchronize(); t1 = time.time()
    g1 = fit.GridTables(det['xy'][:n], det['id'][:n], det['n'][:n]); g2 = fit.GridTables(det['xy'][n:], det['id'][n:], det['n'][n:])
    out = fit.fit_single_cylinder_batch(g1, g2, b['K1'], b['K2'], b['T21'], 45.0)
    torch.cuda.synchronize(); t2 = time.time()
    print(f'iter {it}: detect {2*n} images {1e3*(t1-t):.1f} ms ({1e3*(t1-t)/(2*n):.2f} ms/img), fit {n} frames {1e3*(t2-t1):.2f} ms')
print('status', det['status'].tolist(), 'npts', det['n'].tolist())
print('fit status', out['status'].tolist(), 'm', out['m'].tolist(), 'iters', out['iters'][:, 0].tolist(), 'flags', out['flags'].tolist())
print('fvals', out['fvals'].tolist())
cpe_amd.lib.profile(True)
det = api.detect_grid_batch(frames, ws); torch.cuda.synchronize()
rep = cpe_amd.lib.profile_report(); cpe_amd.lib.profile(False)
print('event profile:', ' | '.join(f"{r[0].split('::')[-1]} x{r[1]} {r[2]:.1f}ms" for r in rep[:int(os.environ.get("CPE_TOP", "10"))]), "| total %.1f ms" % sum(r[2] for r in rep))
