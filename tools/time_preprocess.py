import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cpe_amd
from cpe_amd import synth
n = 128
b = synth.render_batch(n // 2, 1200, 1920, seed=1, device='cuda', with_gt=False)
frames = torch.cat([b['left'], b['right']]); out = torch.empty_like(frames)
L = cpe_amd.lib.load()
def run(): cpe_amd.lib.check(L.cpe_preprocess_batch(frames.data_ptr(), n, 1200, 1920, out.data_ptr(), torch.cuda.current_stream().cuda_stream), 'pp')
for dbg in [0]:
    run(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f'dbg {dbg} k_preprocess: {ms:.2f} ms per launch of {n} images = {1e3*ms/n:.1f} us/img; algorithmic {2*n*1200*1920/ms/1e6:.1f} GB/s')
