#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X: frames/sec for (grid-detect + cylinder fit) on a 1920x1200 batch.

    python bench.py --gpus N --steps K --warmup W [--frames F] [--chunk C]

A "step" is one pass of the whole hot path (detect_grid on both images of every stereo frame,
chooseIdx + triangulate, fitCylinderWPts3 Nelder-Mead, applyCylParamsPrior) over one batch of F synthetic
frames per GPU, inputs resident in HBM before the timed region.  Default F = 4096 = BASELINE.json
configs[2] ("4096-frame 1920x1200 batch, full detect + fitCylinderWPts3 solve, 1xMI355X"); with N > 1 every
rank processes its own F frames (weak scaling, no data-path collective) and one RCCL all-gather of the
128-byte pose records closes each step.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

H, W = 1200, 1920
BYTES_PER_FRAME = 2 * (H * W + 1024 * 24)          # 4 657 152 B per stereo frame
HBM_PEAK = 8.0e12                                   # MI355X_MICROARCH.md: 8 TB/s HBM3E
# algorithmic bytes of one launch (what the kernel must read + write once), DESIGN.md section 4.
#   streaming kernels: bytes per pixel of the frames (or of the blob detector's working rectangle) they walk;
#   list kernels: bytes of the records they consume / produce, from the per-frame counters of the same run.
ALGO_BYTES_PER_PX = {
    'k_preprocess': 2.0,        # u8 frame in, u8 mask out
    'k_ccl_init': 5.0,          # u8 image in, i32 label out (sparse passes write less)
    'k_ccl_merge': 1.0,         # u8 image in; unions touch few labels
    'k_ccl_finish': 5.0,        # u8 image + i32 labels of the set in
    'k_roi_base': 5.0,          # three u8 masks in, two out
    'k_open20_joints': 4.0,     # u8 mask in, three u8 masks out
    'k_ccl_init64': 1.0, 'k_ccl_merge64': 1.0, 'k_ccl_roots64': 1.0,   # u8 image in; labels touched only at run starts
    'k_bitplanes': 1.0 + 17.0 / 8, 'k_bitplanes64': 1.0 + 17.0 / 8, 'k_bk_pass': 1.0,
    'k_clahe_apply': 2.0, 'k_or_and': 4.0, 'k_blur_fused': 2.0,
}


def list_kernel_bytes(short, ws, n_img):
    """algorithmic bytes per launch of the kernels that walk component / blob lists (not pixels)"""
    sw = ws.plane('sweep').cpu().numpy().astype(np.int64)
    st = ws.state()
    dark, bright, blobs = sw[:, 8:25].sum(), sw[:, 25:42].sum(), sw[:, 42:59].sum()
    groups = sum(s_['n_groups'] for s_ in st)
    if short == 'k_blob_merge':      # 32-B blob records in, 1544-B group lists read + written once per touching blob
        return blobs * (32 + 2 * 1544)
    if short == 'k_blob_median':     # ~160 border points of 4 B per blob in, radius out
        return blobs * (160 * 4 + 8)
    if short.startswith('k_blob_trace'):   # a border step reads 3 x 8 B of the bit window, stores a 4-B point
        return blobs * 160 * 28
    if short.startswith('k_sw_'):    # 4-B parent + 1-B level of every pixel that joins, a few neighbours each
        return (dark + bright) * 8 * 5
    return None


def pmc_traffic(kernel, images_per_launch):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same
    pipeline (profiles/r01_pmc_summary.csv, bytes per launch at 64 images per launch; tools/pmc.sh), scaled to this
    launch size.  Byte-granular loads: FETCH_SIZE taken as reported (the guide's x2 applies to 16-B streaming reads)."""
    fn = os.path.join(ROOT, 'profiles', 'r01_pmc_summary.csv')
    if not os.path.exists(fn):
        return None
    import csv
    for r in csv.DictReader(open(fn)):
        if r['kernel'] == kernel:
            per64 = float(r['fetch_KiB_per_launch_raw']) * 1024 + float(r['write_bytes_per_launch'])
            return per64 * images_per_launch / float(r.get('images_per_launch', 64))
    return None


def cpu_baseline(left, right, K1, K2, T21, radius, gpu_rec=None, fit_mode=0, budget_s=20.0):
    """the oracle (single-thread C restatement of the reference path) timed on this host: kind = "port".
    gpu_rec: the GPU pose records of the same frames -> max |pose difference| (the second half of BASELINE's metric)"""
    import oracle
    from oracle import stages as S
    oracle.build()
    n = 0
    dmax = 0.0
    t0 = time.time()
    while n < left.shape[0]:
        a = S.detect_grid(left[n]); b = S.detect_grid(right[n])
        if a['status'] == 0 and b['status'] == 0:
            gp1 = np.concatenate([a['xy'], a['id']], 1); gp2 = np.concatenate([b['xy'], b['id']], 1)
            ref = oracle.fit_single_cylinder(gp1, gp2, K1, K2, T21, radius)
            if gpu_rec is not None and fit_mode == 0 and ref['status'] == 0:
                dmax = max(dmax, float(np.abs(gpu_rec[n, 0:12].reshape(2, 6) - ref['cyl']).max()))
        n += 1
        if time.time() - t0 > budget_s:
            break
    dt = time.time() - t0
    out = dict(value=n / dt, unit='frames/s', cores=1, kind='port',
               sample=f'{n} stereo frames {W}x{H} of the same synthetic workload, oracle detect_grid x2 + fitSingleCylinder, '
                      f'one thread, {dt:.1f} s')
    if gpu_rec is not None and fit_mode == 0:
        out['max_abs_dpose_gpu_vs_port'] = dmax     # cylinder origin / direction (2 x 6) of the same frames; 0.0 = bit-identical
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--frames', type=int, default=4096, help='stereo frames per GPU per step')
    ap.add_argument('--chunk', type=int, default=0,
                    help='stereo frames per kernel batch (workspace size); 0 = 256 at 1920x1200 (~160 GiB of workspace), 64 at 3840x2160')
    ap.add_argument('--unique', type=int, default=256, help='distinct rendered scenes per GPU (cycled with fresh noise)')
    ap.add_argument('--fit-mode', choices=['nm', 'lm'], default='nm',
                    help='nm = fminsearch clone (reference behaviour, default); lm = Levenberg-Marquardt fast mode')
    ap.add_argument('--ransac', type=int, default=0, metavar='H',
                    help='build-defined config 5: wrap the fit in a RANSAC with H hypotheses per frame (LM inside)')
    ap.add_argument('--size', choices=['1920x1200', '3840x2160'], default='1920x1200', help='frame size (config 5 uses 3840x2160)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    import cpe_amd
    from cpe_amd import synth, pipeline, dist as D
    rank, local, world = D.init_from_env('nccl')
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    dev = torch.device(f'cuda:{local}')
    torch.cuda.set_device(dev)
    cpe_amd.lib.load()                                   # no CPU fallback: fail loudly

    global H, W, BYTES_PER_FRAME
    W, H = (int(v) for v in args.size.split('x'))
    BYTES_PER_FRAME = 2 * (H * W + 1024 * 24)
    if args.chunk <= 0:
        args.chunk = 256 if H * W <= 1920 * 1200 else 64
    F = args.frames
    # ---- synthetic inputs, resident in HBM: `unique` rendered scenes, every frame gets its own sensor noise
    U = min(args.unique, F)
    base = synth.render_batch(U, H, W, seed=1000 + rank, device=dev, with_gt=False)
    K1, K2, T21, radius = base['K1'], base['K2'], base['T21'], base['radius']
    left = torch.empty((F, H, W), dtype=torch.uint8, device=dev)
    right = torch.empty((F, H, W), dtype=torch.uint8, device=dev)
    g = torch.Generator(device=dev); g.manual_seed(7 + rank)
    for i0 in range(0, F, U):
        k = min(U, F - i0)
        for src, dst in ((base['left'], left), (base['right'], right)):
            if i0 == 0:
                dst[:k] = src[:k]
            else:   # same scenes, new noise realisation (+-1 DN on ~1/3 of the pixels)
                nz = torch.randint(-1, 2, (k, H, W), generator=g, device=dev, dtype=torch.int16)
                nz = nz * (torch.randint(0, 3, (k, H, W), generator=g, device=dev, dtype=torch.int16) == 0)
                dst[i0:i0 + k] = (src[:k].to(torch.int16) + nz).clamp_(0, 255).to(torch.uint8)
    pipe = pipeline.FramePipeline(H, W, K1, K2, T21, radius, chunk=args.chunk, device=dev,
                                  fit_mode=1 if (args.fit_mode == 'lm' or args.ransac) else 0,
                                  ransac=dict(hypotheses=args.ransac, seed=2026, frame0=rank * F) if args.ransac else None)

    def step():
        rec = pipe.run(left, right)
        return D.gather_records(rec)

    for _ in range(args.warmup):
        allrec = step()
    D.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        allrec = step()
    torch.cuda.synchronize(); D.barrier()
    dt = time.perf_counter() - t0
    dt = D.max_over_ranks(dt, dev)

    # ---- per-kernel hipEvent timers on one more (untimed) pass over one chunk: the dominant kernel's roofline
    roof = None
    if rank == 0:
        # (the timed region overlaps three independent chains on three streams; for per-kernel durations the chains
        #  are put back on one stream, so a launch is timed alone on the GPU as rocprofv3 --kernel-trace would see it
        #  in a serial run)
        os.environ['CPE_SERIAL'] = '1'
        cpe_amd.lib.profile(True)
        c = min(args.chunk, F)
        reps = []
        for _ in range(2):   # twice, per-kernel minimum: an event pair also spans any host stall between record and launch
            pipe.run_chunk(left[:c], right[:c])
            torch.cuda.synchronize()
            reps.append({r[0]: r for r in cpe_amd.lib.profile_report()})   # (the report clears the timers)
        cpe_amd.lib.profile(False)
        rep = sorted(((k, v[1], min(v[2], reps[1].get(k, v)[2])) for k, v in reps[0].items()), key=lambda r: -r[2])
        os.environ.pop('CPE_SERIAL', None)
        tot = sum(r[2] for r in rep)
        name, calls, ms = rep[0]
        short = name.split('::')[-1]
        px_per_launch = 2 * c * H * W
        short = short.split('<')[0]
        bpp = ALGO_BYTES_PER_PX.get(short)
        algo = None
        if bpp is None:
            algo = list_kernel_bytes(short, pipe._ws(2 * c), 2 * c)
            if algo is not None:
                algo = float(algo) / calls
            else:            # other irregular kernels (border tracing ...): each frame's pixels at most once
                bpp = 1.0
        if algo is None:
            algo = bpp * px_per_launch
        avg_s = ms / calls / 1e3
        roof = dict(bound='hbm', kernel=short, calls_per_chunk=calls, avg_launch_ms=ms / calls, share_of_gpu_time=ms / tot,
                    algorithmic_bytes_per_launch=algo, achieved=algo / avg_s / 1e9, peak=HBM_PEAK / 1e9, unit='GB/s',
                    frac=(algo / avg_s) / HBM_PEAK, traffic=pmc_traffic(short, 2 * c),
                    top5=[dict(kernel=r[0].split('::')[-1], calls=r[1], ms=round(r[2], 3)) for r in rep[:5]])
        if short == 'k_preprocess':
            # what actually bounds this kernel: ~314 f64 operations per pixel, multiplies and adds kept
            # apart for bit parity with scipy, so the ceiling is half the 78.6 TFLOP/s FMA figure of MI355X_MICROARCH.md
            flops = 314.0 * px_per_launch
            roof['f64_valu'] = dict(ops_per_px=314, achieved=flops / avg_s / 1e12, peak=78.6 / 2, unit='Tflop/s (separate f64 add / mul)',
                                    frac=flops / avg_s / 1e12 / (78.6 / 2))

    if rank == 0:
        n_pts, iters, fit_st, dl, dr = pipeline.unpack_counters(allrec[:, 15])
        ok = ((fit_st == 0) & (dl == 0) & (dr == 0)).float().mean().item()
        total_frames = F * world * args.steps
        value = total_frames / dt
        out = dict(metric=f'frames/sec (grid-detect + cylinder fit) on {W}x{H} batch', value=value, unit='frames/s',
                   n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=1e3 * dt / args.steps,
                   higher_is_better=True, scaling='weak', vs_baseline=None, dtype='f64', data='synthetic',
                   config=dict(workload=f'{F}-frame {W}x{H} stereo batch per GPU, full detect (both images) + chooseIdx + '
                                        f'triangulate + ' + (f'RANSAC({args.ransac} hypotheses)-wrapped LM fit (build-defined, BASELINE.json configs[4])' if args.ransac
                                                          else f'fitCylinderWPts3 {"LM" if args.fit_mode == "lm" else "Nelder-Mead"} (BASELINE.json configs[2])'),
                               frames_per_gpu=F, chunk=args.chunk, unique_scenes=U, fit_mode=args.fit_mode,
                               parallelism=f'frames sharded x{world}, all_gather of 128-B pose records'),
                   frames_ok_fraction=ok, mean_points_per_frame=float(n_pts.float().mean().item()),
                   path_hbm_frac=value * BYTES_PER_FRAME / HBM_PEAK / world,
                   roofline=roof)
        if not args.no_cpu_baseline and not args.ransac:
            m = min(12, F)
            out['cpu_baseline'] = cpu_baseline(left[:m].cpu().numpy(), right[:m].cpu().numpy(), K1, K2, T21, radius,
                                               gpu_rec=allrec[:m].cpu().numpy(), fit_mode=1 if args.fit_mode == 'lm' else 0)
        print(json.dumps(out))
    D.barrier()


if __name__ == '__main__':
    main()
