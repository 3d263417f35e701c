#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X: frames/sec for (grid-detect + cylinder fit) on a 1920x1200 batch.

    python bench.py --gpus N --steps K --warmup W [--frames F] [--scaling strong|weak] [--chunk C]

A "step" is one pass of the whole hot path (detect_grid on both images of every stereo frame, chooseIdx +
triangulate, fitCylinderWPts3 Nelder-Mead, applyCylParamsPrior) over one batch of synthetic frames, inputs resident
in HBM before the timed region.

N = 1: F = 4096 frames = BASELINE.json configs[2] ("4096-frame 1920x1200 batch, full detect + fitCylinderWPts3
solve, 1xMI355X").
N > 1: one process per GPU.  Started by `python -m torch.distributed.run ... bench.py --gpus N` (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* in the environment) or, from a bare shell, by this script itself: with WORLD_SIZE unset the
parent starts N rank processes BEFORE it touches the GPU, relays rank 0's JSON line and exits non-zero if a rank
fails.  Default for N > 1 is --scaling strong = configs[3] ("4096-frame batch sharded 8x"): the SAME F-frame batch,
rank r owns the contiguous block dist.shard_range(F, r, N) (the loop of exp_gridDetection.m:78-81 cut in N pieces), no
data-path collective, one RCCL all-gather of the 128-byte pose records closes each step.  --scaling weak gives every
rank its own F frames.  Rank 0 prints ONE JSON line.
--stage detect = configs[1] ("256-frame 1920x1200 batch, grid detection + triangulation only"): the same path stopped
after chooseIdx + triangulate (fitSingleCylinder.m:12-17), F = 256 by default.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

H, W = 1200, 1920
HBM_PEAK = 8.0e12                                   # MI355X_MICROARCH.md: 8 TB/s HBM3E
# MI355X vector f64: 256 CUs x 4 SIMD16 x 2.4 GHz = 78.6e12 FMA-flop/s (128 lanes/clk/CU x 2); with -ffp-contract=off
# (multiplies and adds stay separate instructions for bit parity with scipy) a lane retires ONE f64 operation per
# issue, i.e. half of that.
F64_OPS_PEAK = 256 * 64 * 2.4e9                     # 39.3e12 separate f64 add / mul per second
# algorithmic f64 operations per pixel of k_preprocess = every filter output evaluated once per pixel:
# img_as_float 1 + Gaussian 25 taps along y 37 + along x 37 (centre product, 12 x (pair sum, product, accumulate)) + five
# np.gradient arrays 10 + eigenvalue 10 + 15x15 box of b and b^2 14 (square 1, row sums (14 + 7 x 2) / 8 per plane, cv2's
# running column sums 2 per plane, two scalings) + Sauvola threshold and compare 9.  (124 until round 3, when the column
# sums were direct 15-term sums restarted every 4 rows.)
PRE_ALGO_OPS_PER_PX = 118
# algorithmic bytes of one launch (what the kernel must read + write once), DESIGN.md section 3.
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


def bytes_per_frame():
    return 2 * (H * W + 1024 * 24)                  # 4 657 152 B per 1920x1200 stereo frame


# ---------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` from a bare shell
# ---------------------------------------------------------------------------------------------------------------
def _free_port():
    import socket
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def launch_ranks(argv, n, one_device_each=True, poll_s=0.2):
    """start n rank processes of this script (one per GPU), relay rank 0's stdout, fail FAST if any rank fails: the
    first non-zero exit terminates the other ranks (they would otherwise sit in a collective until the process-group
    timeout) and the launcher returns non-zero; children are also ended when the parent leaves for any other reason.
    The parent never makes a GPU call: children are ordinary child processes, nothing is re-executed.
    one_device_each: rank r sees only GPU r (HIP_VISIBLE_DEVICES = the r-th entry of the parent's list, or r) and uses
    cuda:0 -- a rank cannot touch another rank's card by mistake."""
    import threading
    port = _free_port()
    parent_list = [v for v in os.environ.get('HIP_VISIBLE_DEVICES', '').split(',') if v.strip() != '']
    procs, out0 = [], []
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), CPE_BENCH_CHILD='1')
            env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
            if one_device_each:
                env['HIP_VISIBLE_DEVICES'] = parent_list[r] if r < len(parent_list) else str(r)
                env['CPE_BENCH_DEVICE'] = '0'
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                          stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
        drain = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
        drain.start()
        bad = []
        while True:
            rcs = [p.poll() for p in procs]
            bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
            if bad or all(rc is not None for rc in rcs):
                break
            time.sleep(poll_s)
        if bad:
            sys.stderr.write(f'bench.py: rank(s) failed: {bad}; terminating the others\n')
        return 1 if bad else 0
    finally:
        for p in procs:                      # whatever happened: no rank outlives the launcher
            if p.poll() is None:
                p.terminate()
        t_end = time.time() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        if procs:
            try:
                drain.join(timeout=5)
            except Exception:
                pass
        if out0:
            sys.stdout.write(out0[0].decode())
            sys.stdout.flush()


# ---------------------------------------------------------------------------------------------------------------
# CPU baseline leg (rank 0, N = 1): the oracle = C port of the reference path, on the host cores of this box
# ---------------------------------------------------------------------------------------------------------------
def cpu_worker(path, lo, hi):
    """child process of the cpu_baseline leg: oracle detect_grid x2 + fitSingleCylinder on frames [lo, hi) of the
    sample file; prints one JSON line (seconds of pure compute, poses)"""
    import numpy as np
    import oracle
    from oracle import stages as S
    z = np.load(path, mmap_mode='r')
    left, right = z[0], z[1]
    meta = json.load(open(path + '.json'))
    K1, K2, T21 = (np.array(meta[k]) for k in ('K1', 'K2', 'T21'))
    oracle.lib()
    poses = []
    t0 = time.perf_counter()
    for i in range(lo, hi):
        a = S.detect_grid(np.ascontiguousarray(left[i])); b = S.detect_grid(np.ascontiguousarray(right[i]))
        cyl = None
        if a['status'] == 0 and b['status'] == 0:
            ref = oracle.fit_single_cylinder(np.concatenate([a['xy'], a['id']], 1), np.concatenate([b['xy'], b['id']], 1),
                                             K1, K2, T21, meta['radius'])
            if ref['status'] == 0:
                cyl = np.asarray(ref['cyl']).reshape(-1).tolist()
        poses.append(cyl)
    print(json.dumps(dict(lo=lo, hi=hi, seconds=time.perf_counter() - t0, poses=poses)))


def usable_cores():
    """CPUs this process may use: the affinity mask, capped by the cgroup's CPU quota (the GPU box gives a share)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return n


def physical_cores():
    """distinct (package, core) pairs among the CPUs this process may run on"""
    try:
        cpus = sorted(os.sched_getaffinity(0))
        seen = set()
        for c in cpus:
            base = f'/sys/devices/system/cpu/cpu{c}/topology/'
            seen.add((open(base + 'physical_package_id').read().strip(), open(base + 'core_id').read().strip()))
        return len(seen)
    except Exception:
        return usable_cores()


def cpu_baseline(left, right, K1, K2, T21, radius, gpu_rec=None, fit_mode=0, per_proc=4, single=6):
    """the oracle (plain-C restatement of the reference path, kind = "port") timed on this host twice:
      (i)  one process, one thread, `single` frames (how the reference's own loop runs: one image at a time);
      (ii) one process per physical core, frames sharded (`per_proc` each) -> the per-host rate.
    The reference itself (OpenCV + MATLAB) cannot run here.  gpu_rec: GPU pose records of the same frames ->
    max |pose difference| (second half of BASELINE's metric)."""
    import numpy as np
    import oracle
    oracle.build()
    procs = max(1, min(physical_cores(), usable_cores(), 64))
    n2 = min(left.shape[0], procs * per_proc)
    per = (n2 + procs - 1) // procs
    path = os.path.join('/dev/shm' if os.path.isdir('/dev/shm') else '/tmp', f'cpe_cpu_sample_{os.getpid()}.npy')
    np.save(path, np.stack([left[:n2], right[:n2]]))
    json.dump(dict(K1=np.asarray(K1).tolist(), K2=np.asarray(K2).tolist(), T21=np.asarray(T21).tolist(), radius=radius,
                   fit_mode=fit_mode), open(path + '.json', 'w'))
    env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1')

    def run(ranges):
        t0 = time.perf_counter()
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--cpu-worker', path, str(lo), str(hi)],
                               env=env, stdout=subprocess.PIPE) for lo, hi in ranges if hi > lo]
        outs = [json.loads(p.communicate()[0].decode().strip().splitlines()[-1]) for p in ps]
        return time.perf_counter() - t0, outs
    try:
        n1 = min(single, n2)
        _, o1 = run([(0, n1)])
        wall, on = run([(k * per, min(n2, (k + 1) * per)) for k in range(procs)])
    finally:
        for f in (path, path + '.json'):
            if os.path.exists(f):
                os.remove(f)
    dmax = None
    if gpu_rec is not None and fit_mode == 0:
        dmax = 0.0
        for o in on:
            for i, cyl in zip(range(o['lo'], o['hi']), o['poses']):
                if cyl is not None:
                    dmax = max(dmax, float(np.abs(gpu_rec[i, 0:12] - np.array(cyl)).max()))
    t1 = o1[0]['seconds']
    used = len(on)
    out = dict(value=n2 / wall, unit='frames/s', cores=used, kind='port',
               sample=f'{n2} stereo frames {W}x{H} of the same synthetic workload (oracle detect_grid x2 + fitSingleCylinder), '
                      f'{used} single-thread processes (one per physical core) x {per} frames, {wall:.1f} s wall incl. process start',
               single_thread=dict(value=n1 / t1, unit='frames/s', cores=1, sample=f'{n1} frames, {t1:.1f} s'),
               host=dict(os_cpu_count=os.cpu_count(), usable_cpus=usable_cores(), physical_cores=physical_cores()),
               note='the reference itself (OpenCV 4.5.5 + MATLAB) cannot run on this host; "port" = this repo\'s single-thread C '
                    'restatement of it, which is faster than the reference per core (no interpreter, endpoint-local dilation '
                    'instead of full-frame passes); a reported baseline, not a target')
    if dmax is not None:
        out['max_abs_dpose_gpu_vs_port'] = dmax     # cylinder origin / direction (2 x 6) of the same frames; 0.0 = bit-identical
    return out


# ---------------------------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------------------------
def list_kernel_bytes(short, ws, n_img):
    """algorithmic bytes per launch of the kernels that walk component / blob lists (not pixels)"""
    import numpy as np
    sw = ws.plane('sweep').cpu().numpy().astype(np.int64)
    dark, bright, blobs = sw[:, 8:25].sum(), sw[:, 25:42].sum(), sw[:, 42:59].sum()
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
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE summary of this
    pipeline (profiles/r*_pmc_summary.csv, tools/pmc.sh: separate passes), scaled to this launch size.  NOT measured in
    this run (PMC needs rocprofv3 around the process): the source file is named beside the figure."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_summary.csv')))
    for fn in reversed(files):
        for r in csv.DictReader(open(fn)):
            if r['kernel'] == kernel:
                # FETCH_SIZE is in KiB and under-reports reads by 2x on gfx950 (MI355X_MICROARCH.md; checked here on kernels that
                # read a u8 plane exactly once: k_clahe_hist, k_clahe_apply, k_bitplanes64 report half of it) -> corrected column
                per = float(r['fetch_bytes_x2_corrected']) + float(r['write_bytes_per_launch'])
                return per * images_per_launch / float(r.get('images_per_launch', 64)), os.path.basename(fn)
    return None, None


class StubPipeline:
    """CPU stand-in for FramePipeline used by tests/test_dist_cpu.py (--stub): exercises the launcher, the sharding and
    the gather of bench.py with gloo, no GPU and no kernels"""

    def __init__(self, lo):
        self.lo = lo

    def run(self, left, right):
        import torch
        rec = torch.zeros((left.shape[0], 16), dtype=torch.float64)
        rec[:, 0] = torch.arange(self.lo, self.lo + left.shape[0], dtype=torch.float64)
        rec[:, 1] = left.reshape(left.shape[0], -1).to(torch.float64).sum(1)
        return rec


def make_frames(synth, torch, lo, hi, U, dev):
    """frames [lo, hi) of THE batch (the same frames whatever the number of ranks): frame g is rendered scene g % U
    plus noise realisation g // U (realisation 0 = the rendering itself; others: +-1 DN on ~1/3 of the pixels)"""
    base = synth.render_batch(U, H, W, seed=1000, device=dev, with_gt=False)
    n = hi - lo
    stereo = torch.empty((n, 2, H, W), dtype=torch.uint8, device=dev)   # frame-major pairs: a chunk of frames is one
    left, right = stereo[:, 0], stereo[:, 1]                             # contiguous run of images (read in place)
    g = torch.Generator(device=dev)
    b0, b1 = lo // U, (hi + U - 1) // U
    for b in range(b0, b1):
        s0, s1 = max(lo, b * U), min(hi, (b + 1) * U)
        if s1 <= s0:
            continue
        k0, k1 = s0 - b * U, s1 - b * U
        for cam, (src, dst) in enumerate(((base['left'], left), (base['right'], right))):
            if b == 0:
                dst[s0 - lo:s1 - lo] = src[k0:k1]
                continue
            g.manual_seed(7 + 2 * b + cam)           # per (block, camera): the data do not depend on the sharding
            nz = torch.randint(-1, 2, (U, H, W), generator=g, device=dev, dtype=torch.int16)
            nz = nz * (torch.randint(0, 3, (U, H, W), generator=g, device=dev, dtype=torch.int16) == 0)
            dst[s0 - lo:s1 - lo] = (src[k0:k1].to(torch.int16) + nz[k0:k1]).clamp_(0, 255).to(torch.uint8)
            del nz
    return base, left, right


def run_rank(args):
    import numpy as np
    import torch
    import cpe_amd
    from cpe_amd import synth, pipeline, dist as D
    global H, W
    W, H = (int(v) for v in args.size.split('x'))
    backend = 'gloo' if args.stub else args.backend
    if args.stub and os.environ.get('CPE_BENCH_FAIL_RANK') == os.environ.get('RANK', '0'):
        raise SystemExit(3)                               # tests/test_dist_cpu.py: a rank that dies before the rendezvous
    # this rank's GPU: LOCAL_RANK under torch.distributed.run (every rank sees all cards); 0 when bench.py's own launcher
    # gave the rank one card (CPE_BENCH_DEVICE) or on the one-GPU rehearsal (--share-gpu)
    one_dev = os.environ.get('CPE_BENCH_DEVICE')
    dev_index = 0 if args.share_gpu else int(one_dev) if one_dev is not None else None
    rank, local, world = D.init_from_env(backend, device_index=None if args.stub else dev_index)
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    if args.stub:
        dev = torch.device('cpu')
    else:
        dev = torch.device(f'cuda:{local if dev_index is None else dev_index}')
        torch.cuda.set_device(dev)
        cpe_amd.lib.load()                               # no CPU fallback: fail loudly
    scaling = args.scaling or ('strong' if world > 1 else 'weak')
    auto_chunk = args.chunk <= 0
    if auto_chunk:
        args.chunk = 228 if H * W <= 1920 * 1200 else 56   # x lanes x 2 images x ~170 MB (1920x1200) / ~582 MB (4K) of workspace
    if scaling == 'strong':
        total = args.frames
        lo, hi = D.shard_range(total, rank, world)
    else:
        total = args.frames * world
        lo, hi = rank * args.frames, (rank + 1) * args.frames
    F = hi - lo
    if auto_chunk and F > 0:
        # equal chunks of at most 228 frames (round 3, workspace 170-195 MB per image: 208-240 frames per call is the sweet spot --
        # 4096 frames as chunks of 171: 2767 frames/s, 208: 2854, 224: 2867, 240: 2865, 300: 2753; calls of exactly 256 or 512
        # images are 6-10 % slower than their neighbours (128: 2478, 256: 2712) -- not an alignment effect: skewing the plane
        # offsets and the lanes' workspaces changed nothing; an odd number of chunks on two lanes costs nothing);
        # a shard that fits one call of 256 frames is not split
        nch = 1 if F <= 256 and H * W <= 1920 * 1200 else -(-F // args.chunk)
        args.chunk = -(-F // nch)
    # ---- synthetic inputs, resident in HBM before the timed region
    if args.stub:
        K1 = K2 = T21 = None; radius = 45.0
        left = (torch.arange(lo, hi, dtype=torch.int64) % 251).to(torch.uint8).view(F, 1, 1).expand(F, 4, 4).contiguous()
        right = left.clone()
        pipe = StubPipeline(lo)
        U = 0
    else:
        U = min(args.unique, total)
        base, left, right = make_frames(synth, torch, lo, hi, U, dev)
        K1, K2, T21, radius = base['K1'], base['K2'], base['T21'], base['radius']
        del base
        pipe = pipeline.FramePipeline(H, W, K1, K2, T21, radius, chunk=min(args.chunk, max(F, 1)), device=dev,
                                      fit_mode=1 if (args.fit_mode == 'lm' or args.ransac) else 0, lanes=args.lanes,
                                      ransac=dict(hypotheses=args.ransac, seed=2026, frame0=lo) if args.ransac else None,
                                      stage=args.stage)

    def sync():
        if dev.type == 'cuda':
            torch.cuda.synchronize()

    def step():
        rec = pipe.run(left, right)
        return rec, D.gather_records(rec, total)

    for _ in range(args.warmup):
        rec, allrec = step()
    D.barrier(); sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rec, allrec = step()
    sync(); D.barrier()
    dt = time.perf_counter() - t0
    dt = D.max_over_ranks(dt, dev)

    # ---- the closing collective alone (untimed extra): K all-gathers of the pose records
    t0 = time.perf_counter()
    for _ in range(5):
        D.gather_records(rec, total)
    sync()
    gather_ms = D.max_over_ranks((time.perf_counter() - t0) / 5 * 1e3, dev)

    # ---- per-kernel hipEvent timers on one more (untimed) pass over one chunk: the dominant kernel's roofline
    roof = None
    if rank == 0 and not args.stub:
        # (the timed region overlaps independent chains on several streams; for per-kernel durations the chains are put
        #  back on one stream, so a launch is timed alone on the GPU as rocprofv3 --kernel-trace sees it in a serial run)
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
        n_launch = sum(r[1] for r in rep)
        name, calls, ms = rep[0]
        short = name.split('::')[-1].split('<')[0]
        px_per_launch = 2 * c * H * W
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
        traffic, traffic_src = pmc_traffic(short, 2 * c)
        hbm = dict(algorithmic_bytes_per_launch=algo, achieved=algo / avg_s / 1e9, peak=HBM_PEAK / 1e9, unit='GB/s',
                   frac=(algo / avg_s) / HBM_PEAK)
        roof = dict(kernel=short, calls_per_chunk=calls, images_per_launch=2 * c, avg_launch_ms=ms / calls,
                    share_of_gpu_time=ms / tot, launches_per_chunk=n_launch, traffic=traffic,
                    traffic_source=(f'profiles/{traffic_src}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run, '
                                    f'(FETCH x 1024 x 2 on gfx950 + WRITE), scaled to this launch size; not measured in this run') if traffic_src else None,
                    top5=[dict(kernel=r[0].split('::')[-1], calls=r[1], ms=round(r[2], 3)) for r in rep[:5]])
        if short == 'k_preprocess':
            # what bounds this kernel: f64 VALU.  `achieved` counts ALGORITHMIC operations only (one
            # evaluation of every filter output per pixel; tile-halo recomputation is overhead, not work).
            ops = float(PRE_ALGO_OPS_PER_PX) * px_per_launch
            roof.update(bound='f64_valu', algorithmic_ops_per_px=PRE_ALGO_OPS_PER_PX, achieved=ops / avg_s / 1e12,
                        peak=F64_OPS_PEAK / 1e12, unit='Tflop/s (separate f64 add / mul, no FMA: bit parity with scipy)',
                        frac=ops / avg_s / F64_OPS_PEAK, hbm=hbm)
        else:
            roof.update(bound='hbm', **hbm)

    if rank == 0:
        out = dict(metric=f'frames/sec (grid-detect + cylinder fit) on {W}x{H} batch', unit='frames/s',
                   n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=1e3 * dt / args.steps,
                   higher_is_better=True, scaling=scaling, vs_baseline=None, dtype='f64', data='synthetic')
        total_frames = total * args.steps
        value = total_frames / dt
        out['value'] = value
        if args.stub:
            out['config'] = dict(workload='stub pipeline (launcher / shard / gather test)', frames_total=total, frames_per_gpu=F)
            out['checksum'] = float(allrec[:, 0].sum().item()); out['rows'] = int(allrec.shape[0])
            out['payload_sum'] = float(allrec[:, 1].sum().item())
        else:
            n_pts, iters, fit_st, dl, dr = pipeline.unpack_counters(allrec[:, 15])
            ok = ((fit_st == 0) & (dl == 0) & (dr == 0)).float().mean().item()
            fit_txt = ('nothing else: grid detection + triangulation only, no cylinder fit (BASELINE.json configs[1])' if args.stage == 'detect'
                       else f'RANSAC({args.ransac} hypotheses)-wrapped LM fit (build-defined, BASELINE.json configs[4])' if args.ransac
                       else f'fitCylinderWPts3 {"LM" if args.fit_mode == "lm" else "Nelder-Mead"} '
                            f'(BASELINE.json configs[{2 if world == 1 else 3}])')
            out['config'] = dict(workload=f'{total}-frame {W}x{H} stereo batch' + (f' sharded x{world}' if world > 1 else '') +
                                          f', full detect (both images) + chooseIdx + triangulate + ' + fit_txt, stage=args.stage,
                                 devices_visible=torch.cuda.device_count(), device=str(dev),
                                 frames_total=total, frames_per_gpu=F, chunk=min(args.chunk, F), lanes=args.lanes, unique_scenes=U, fit_mode=args.fit_mode,
                                 parallelism=f'contiguous frame shards x{world} (dist.shard_range), no data-path collective, '
                                             f'one all_gather of 128-B pose records per step', backend=backend, ranks=world)
            out.update(frames_ok_fraction=ok, mean_points_per_frame=float(n_pts.float().mean().item()),
                       path_hbm_frac=value * bytes_per_frame() / HBM_PEAK / world)
        out['all_gather_ms'] = gather_ms
        out['all_gather_bytes'] = int(total * 128)
        out['roofline'] = roof
        if not args.stub and not args.no_cpu_baseline and not args.ransac and world == 1 and args.stage == 'full':
            m = min(F, 64 * 4)
            out['cpu_baseline'] = cpu_baseline(left[:m].cpu().numpy(), right[:m].cpu().numpy(), K1, K2, T21, radius,
                                               gpu_rec=allrec[:m].cpu().numpy(), fit_mode=1 if args.fit_mode == 'lm' else 0)
        print(json.dumps(out), flush=True)
    D.barrier()
    D.shutdown()


def main():
    if len(sys.argv) >= 5 and sys.argv[1] == '--cpu-worker':
        return cpu_worker(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--frames', type=int, default=None,
                    help='stereo frames of the batch (strong scaling: in total, sharded over the ranks; weak: per GPU); '
                         'default 4096 (--stage full) / 256 (--stage detect)')
    ap.add_argument('--stage', choices=['full', 'detect'], default='full',
                    help='full = BASELINE.json configs[2] (detect + fit); detect = configs[1]: 256 frames, grid detection of both images + '
                         'chooseIdx + triangulate, stop before fitCylinderWPts3 (fitSingleCylinder.m:12-17)')
    ap.add_argument('--scaling', choices=['strong', 'weak'], default=None,
                    help='default: strong for N > 1 (BASELINE.json configs[3]: the 4096-frame batch sharded N x)')
    ap.add_argument('--chunk', type=int, default=0,
                    help='stereo frames per kernel batch (workspace size); 0 = equal chunks of at most 228 frames at 1920x1200 '
                         '(228 for a 4096-frame shard, 171 for a 512-frame one), 56 at 3840x2160 (x --lanes in flight)')
    ap.add_argument('--lanes', type=int, default=2,
                    help='chunks in flight, each on its own HIP stream with its own workspace: the narrow tail of one chunk (fragments, '
                         'lines: one workgroup per image) runs beside the wide kernels of the next')
    ap.add_argument('--unique', type=int, default=256, help='distinct rendered scenes (cycled with fresh noise)')
    ap.add_argument('--fit-mode', choices=['nm', 'lm'], default='nm',
                    help='nm = fminsearch clone (reference behaviour, default); lm = Levenberg-Marquardt fast mode')
    ap.add_argument('--ransac', type=int, default=0, metavar='H',
                    help='build-defined config 5: wrap the fit in a RANSAC with H hypotheses per frame (LM inside)')
    ap.add_argument('--size', choices=['1920x1200', '3840x2160'], default='1920x1200', help='frame size (config 5 uses 3840x2160)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--backend', choices=['nccl', 'gloo'], default='nccl', help='nccl = RCCL over xGMI (default); gloo: rehearsal')
    ap.add_argument('--share-gpu', action='store_true', help='rehearsal on a one-GPU box: every rank uses cuda:0 (needs --backend gloo)')
    ap.add_argument('--stub', action='store_true', help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.frames is None:
        args.frames = 256 if args.stage == 'detect' else 4096
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(sys.argv[1:], args.gpus, one_device_each=not (args.share_gpu or args.stub)))
    run_rank(args)


if __name__ == '__main__':
    main()
