"""time FramePipeline.run for different numbers of chunks in flight:  python tools/time_pipeline.py [frames] [chunk]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cpe_amd
from cpe_amd import synth, pipeline
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device('cuda:0')
b = synth.render_batch(min(F, 128), 1200, 1920, seed=1, device=dev, with_gt=False)
rep = (F + b['left'].shape[0] - 1) // b['left'].shape[0]
left = b['left'].repeat(rep, 1, 1)[:F].contiguous(); right = b['right'].repeat(rep, 1, 1)[:F].contiguous()
ref = None
for lanes in (1, 2, 3):
    pipe = pipeline.FramePipeline(1200, 1920, b['K1'], b['K2'], b['T21'], b['radius'], chunk=chunk, device=dev, lanes=lanes)
    rec = pipe.run(left, right); torch.cuda.synchronize()
    t = time.time(); rec = pipe.run(left, right); torch.cuda.synchronize(); dt = time.time() - t
    same = True if ref is None else bool(torch.equal(rec, ref))
    ref = rec if ref is None else ref
    print(f'lanes {lanes}: {F} frames in {dt*1e3:.1f} ms = {F/dt:.1f} frames/s ({1e3*dt/F/2:.3f} ms/img), records identical to lanes=1: {same}')
    del pipe; torch.cuda.empty_cache()
