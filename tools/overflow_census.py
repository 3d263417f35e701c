"""Which fixed capacity trips on which frames?  Runs the detector over clean synthetic 4K frames and over 1920x1200 frames with
heavy uniform noise, prints the status histogram, the OVF_* bits of every status-6 frame and the counters that matter
(components per labelling pass, blobs / groups, joints, lines, points).
    python tools/overflow_census.py [n4k] [nnoisy]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import cpe_amd
from cpe_amd import api, synth

OVF = ['ROOTS', 'LINES', 'TRACE', 'JOINTS', 'VERTS', 'SEGS', 'KERNEL', 'EXPAND', 'BLOBS', 'DISTS', 'GROUPS', 'SWEEP']


def bits(v):
    return '|'.join(n for k, n in enumerate(OVF) if v >> k & 1) or '-'


saved = [0]
save_dir = os.environ.get('CENSUS_SAVE')


def run(frames, tag, chunk=16):
    dev = torch.device('cuda:0')
    n = frames.shape[0]
    hist = {}
    for i0 in range(0, n, chunk):
        part = frames[i0:i0 + chunk].to(dev)
        det = api.detect_grid_batch(part)
        torch.cuda.synchronize()
        st = det['ws'].state()
        sw = det['ws'].plane('sweep').cpu().numpy()
        for i in range(part.shape[0]):
            s = int(det['status'][i])
            hist[s] = hist.get(s, 0) + 1
            d = st[i]
            line = (f'{tag} frame {i0 + i}: status {s} ovf {bits(d["overflow"])} pts {int(det["n"][i])} rows {d["n_rows"]} cols {d["n_cols"]} '
                    f'joints {d["n_joints"]}/{d["n_joints_all"]} groups {d["n_groups"]} kp {d["n_kp"]} roots {d["n_roots"]}/{d["n_roots_p"]}/{d["n_roots_s"]} '
                    f'seg {d["n_seg0"]},{d["n_seg1"]} dark max {sw[i, 8:25].max()} sum {sw[i, 8:25].sum()} bright max {sw[i, 25:42].max()} sum {sw[i, 25:42].sum()} trace sum {sw[i, 130:147].sum()} blobs max {sw[i, 42:59].max()}')
            if s == 6 or (i0 + i) < 2 or os.environ.get('CENSUS_ALL'):
                print(line)
            if s == 6 and save_dir and saved[0] < 3:
                os.makedirs(save_dir, exist_ok=True)
                np.save(os.path.join(save_dir, f'ovf_{tag.split()[0]}_{i0 + i}.npy'), frames[i0 + i].numpy())
                saved[0] += 1
    print(tag, 'status histogram', dict(sorted(hist.items())))


def main():
    n4k = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    nn = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    cpe_amd.lib.load()
    if n4k:
        b = synth.render_batch(n4k // 2, 2160, 3840, seed=1000, device='cuda', with_gt=False)
        run(torch.cat([b['left'], b['right']]).cpu(), '4K clean', 8)
    if nn:
        b = synth.render_batch(nn // 2, 1200, 1920, seed=77, device='cuda', with_gt=False)
        fr = torch.cat([b['left'], b['right']]).cpu().numpy().astype(np.int32)
        rng = np.random.default_rng(5)
        for i in range(fr.shape[0]):
            a = 7 + i % 5           # +-7 .. +-11 DN uniform noise
            fr[i] += rng.integers(-a, a + 1, size=fr[i].shape)
        run(torch.from_numpy(np.clip(fr, 0, 255).astype(np.uint8)), '1920x1200 noisy', 8)


if __name__ == '__main__':
    main()
