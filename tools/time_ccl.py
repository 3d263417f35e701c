"""per-pass cost of the run-based labelling on the CLAHE plane and on derived masks (128 images)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cpe_amd
from cpe_amd import synth, api
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
b = synth.render_batch(n // 2, 1200, 1920, seed=1, device='cuda', with_gt=False)
frames = torch.cat([b['left'], b['right']])
ws = api.DetectWorkspace(n, 1200, 1920, frames.device)
det = api.detect_grid_batch(frames, ws)
L = cpe_amd.lib.load()
planes = {k: ws.plane(k).clone() for k in ('clahe', 'exp_h', 'mask_contour', 'binary')}
def run(img, thr, inv, conn8, cm, bbox, roots):
    cpe_amd.lib.check(L.cpe_debug_ccl(img.data_ptr(), n, 1200, 1920, thr, inv, conn8, cm, bbox, roots, ws.view.data_ptr(), ws.bytes, torch.cuda.current_stream().cuda_stream), 'ccl')
for name, img, thr, inv, conn8, cm, bbox, roots in (('dark set of clahe <= 50 (4-conn, counts, rect)', planes['clahe'], 50, 1, 0, 1, 2, 1),
                                                      ('bright set of clahe > 50 (8-conn)', planes['clahe'], 50, 0, 1, 0, 2, 1),
                                                      ('exp_h mask', planes['exp_h'], 0, 0, 1, 0, 0, 1),
                                                      ('mask_contour (one big blob)', planes['mask_contour'], 0, 0, 1, 0, 0, 1),
                                                      ('binary ridge mask', planes['binary'], 0, 0, 1, 0, 0, 1)):
    run(img, thr, inv, conn8, cm, bbox, roots); torch.cuda.synchronize()
    cpe_amd.lib.profile(True)
    for _ in range(3): run(img, thr, inv, conn8, cm, bbox, roots)
    torch.cuda.synchronize()
    rep = cpe_amd.lib.profile_report(); cpe_amd.lib.profile(False)
    print(f'{name}: ' + ' '.join(f"{r[0].split('::')[-1][6:]}={r[2]/r[1]:.2f}ms" for r in rep))
