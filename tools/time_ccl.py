import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cpe_amd
from cpe_amd import synth, api
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
b = synth.render_batch(n // 2, 1200, 1920, seed=1, device='cuda', with_gt=False)
frames = torch.cat([b['left'], b['right']])
ws = api.DetectWorkspace(n, 1200, 1920, frames.device)
det = api.detect_grid_batch(frames, ws)
cl = ws.plane('clahe').clone()
L = cpe_amd.lib.load()
def run(thr, inv, conn8, cm, bbox, roots):
    cpe_amd.lib.check(L.cpe_debug_ccl(cl.data_ptr(), n, 1200, 1920, thr, inv, conn8, cm, bbox, roots, ws.view.data_ptr(), ws.bytes, torch.cuda.current_stream().cuda_stream), 'ccl')
print(ws.state()[0])
for thr in (50, 100):
    for cm, bbox, roots in ((0,0,0),(0,0,1),(2,0,0),(0,1,0),(2,1,1),(0,2,0),(2,3,1)):
        run(thr, 0, 1, cm, bbox, roots); torch.cuda.synchronize()
        cpe_amd.lib.profile(True)
        for _ in range(3): run(thr, 0, 1, cm, bbox, roots)
        torch.cuda.synchronize()
        rep = cpe_amd.lib.profile_report(); cpe_amd.lib.profile(False)
        st = ws.state()[0]
        print(f'thr {thr} count {cm} bbox/rect {bbox} roots {roots} n_roots {st["n_roots"]}: ' + ' '.join(f"{r[0].split('::')[-1][6:]}={r[2]/r[1]:.2f}ms" for r in rep))
