"""CPU-only sweep of the oracle under the sanitizer build (oracle/Makefile `asan`): degraded frames of several sizes
through detect_grid (cylinder, sub-pixel, planar) and the stereo fit.  Run as
    LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 ORACLE_ASAN=1 python tools/asan_oracle_sweep.py [cases]
A finding aborts the process (AddressSanitizer) or prints `runtime error` (UBSan, -fno-sanitize-recover: abort)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import cpe_amd  # noqa: F401
import oracle
from cpe_amd import synth
from oracle import stages as S

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sizes = [(480, 640), (483, 650), (602, 801), (1200, 1920)]
stat = {}
for c in range(cases):
    seed = 4000 + 11 * c
    rng = np.random.default_rng(seed)
    h, w = sizes[c % len(sizes)]
    b = synth.render_batch(1, h, w, seed=seed, with_gt=False)
    imgs = [synth.degrade(b[k][0].numpy(), rng)[0] for k in ('left', 'right')]
    refs = [S.detect_grid(g, lines=True, subpixel=bool(c % 2)) for g in imgs]
    for r in refs:
        stat[r['status']] = stat.get(r['status'], 0) + 1
    S.detect_grid_plane(imgs[0])
    if all(r['status'] == 0 for r in refs):
        oracle.fit_single_cylinder(np.concatenate([refs[0]['xy'], refs[0]['id']], 1), np.concatenate([refs[1]['xy'], refs[1]['id']], 1),
                                   b['K1'], b['K2'], b['T21'], b['radius'])
# the frames that exceed round 2's tables (tests/test_detect_gpu.py) and a frame of pure noise (capacity overflow path)
for seed in (4011, 9508):
    rng = np.random.default_rng(seed)
    b = synth.render_batch(1, 1200, 1920, seed=seed, with_gt=False)
    r = S.detect_grid(synth.degrade(b['left'][0].numpy(), rng)[0])
    stat[r['status']] = stat.get(r['status'], 0) + 1
r = S.detect_grid(np.random.default_rng(1).integers(0, 255, size=(600, 800), dtype=np.uint8))
stat[r['status']] = stat.get(r['status'], 0) + 1
print('statuses seen:', dict(sorted(stat.items())), '-- no sanitizer finding')
