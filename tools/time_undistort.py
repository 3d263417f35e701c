"""row f-3: time the remap kernel (HBM bound: 2 B/px per frame + 6 B/px of map per 8 frames)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cpe_amd
from cpe_amd import iotool
h, w, n = 1200, 1920, int(sys.argv[1]) if len(sys.argv) > 1 else 256
cam = dict(IntrinsicMatrix=[[1400.0, 0, 955.5], [0, 1398.0, 601.25], [0, 0, 1]], RadialDistortion=[-0.21, 0.07], TangentialDistortion=[0.001, -0.0007])
t = time.time(); und = iotool.Undistorter(cam, h, w, 'cuda:0'); torch.cuda.synchronize(); print(f'map build {1e3*(time.time()-t):.2f} ms (once per camera)')
src = torch.randint(0, 255, (n, h, w), dtype=torch.uint8, device='cuda'); dst = torch.empty_like(src)
for _ in range(2): und(src, dst)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): und(src, dst)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 10
by = (2.0 + 6.0 / 8) * n * h * w
print(f'remap {n} frames: {ms:.3f} ms = {ms*1e3/n:.2f} us/frame, {by/ms/1e6:.0f} GB/s algorithmic ({by/ms/1e6/8000:.3f} of 8 TB/s)')
