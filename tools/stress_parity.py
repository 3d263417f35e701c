"""Differential stress run: randomly degraded synthetic frames through the GPU path and the CPU oracle.
    python tools/stress_parity.py [cases] [first seed] [cylinder|plane] [full] [subpixel]
(full: 1200x1920 frames only; subpixel: the optional grey-level line refinement, row f-4, on both sides)
For the cylinder script the stereo pair also goes through chooseIdx + triangulate + the Nelder-Mead fit on both sides
(cylinder parameters compared for equality).
Every frame must come out the same (status, centre, points, ids); a capacity overflow (status 6, build defined) is
reported separately.  Exit code 1 if any frame differs."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import cpe_amd
import oracle
from cpe_amd import api, fit, synth
from oracle import stages as S

SIZES = [(480, 640), (600, 800), (512, 768), (483, 650), (602, 801), (720, 1280)]


degrade = synth.degrade     # the degradations live beside the renderer: tests/test_detect_gpu.py replays single seeds


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    target = sys.argv[3] if len(sys.argv) > 3 else 'cylinder'
    subpixel = 'subpixel' in sys.argv[4:]
    ref_fn = S.detect_grid_plane if target == 'plane' else (lambda g: S.detect_grid(g, subpixel=subpixel))
    dev = torch.device('cuda:0')
    cpe_amd.lib.load(); oracle.build()
    bad = ovf = okf = fits = 0
    t0 = time.time()
    for c in range(cases):
        seed = seed0 + c
        rng = np.random.default_rng(seed)
        h, w = (1200, 1920) if 'full' in sys.argv[4:] else SIZES[seed % len(SIZES)]
        b = synth.render_batch(1, h, w, seed=seed, with_gt=False)
        frames, notes = [], []
        for img in (b['left'][0].numpy(), b['right'][0].numpy()):
            d, what = degrade(img, rng); frames.append(d); notes.append(what)
        frames = np.stack(frames)
        det = api.detect_grid_batch(torch.from_numpy(frames).to(dev), target=target, subpixel=subpixel)
        torch.cuda.synchronize()
        state = det['ws'].state()
        refs = []
        for i in range(2):
            ref = ref_fn(frames[i]); refs.append(ref)
            st = int(det['status'][i])
            if st == 6:
                ovf += 1
                print(f'seed {seed} frame {i} {h}x{w}: capacity overflow bits {state[i]["overflow"]} (oracle status {ref["status"]}) {notes[i]}')
                continue
            same = st == ref['status']
            if same and st == 0:
                m = int(det['n'][i])
                same = (m == len(ref['xy']) and np.array_equal(det['xy'][i, :m].cpu().numpy(), ref['xy'])
                        and np.array_equal(det['id'][i, :m].cpu().numpy(), ref['id'])
                        and np.array_equal(det['center'][i].cpu().numpy(), ref['center']))
                okf += 1
            if not same:
                bad += 1
                print(f'seed {seed} frame {i} {h}x{w}: MISMATCH gpu status {st} oracle {ref["status"]} {notes[i]}')
        if target == 'cylinder' and all(r['status'] == 0 for r in refs) and int(det['status'][0]) == 0 and int(det['status'][1]) == 0:
            g1 = fit.GridTables(det['xy'][:1], det['id'][:1], det['n'][:1]); g2 = fit.GridTables(det['xy'][1:], det['id'][1:], det['n'][1:])
            out = fit.fit_single_cylinder_batch(g1, g2, b['K1'], b['K2'], b['T21'], b['radius'])
            want = oracle.fit_single_cylinder(np.concatenate([refs[0]['xy'], refs[0]['id']], 1), np.concatenate([refs[1]['xy'], refs[1]['id']], 1),
                                              b['K1'], b['K2'], b['T21'], b['radius'])
            fits += 1
            if not np.array_equal(out['cyl'][0].cpu().numpy(), want['cyl']):
                bad += 1
                print(f'seed {seed}: FIT MISMATCH', out['cyl'][0].cpu().numpy().tolist(), want['cyl'].tolist())
    print(f'{fits} stereo pairs fitted on both sides')
    print(f'{2 * cases} frames in {time.time() - t0:.0f} s: {bad} mismatches, {ovf} capacity overflows, {okf} with points')
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
