"""N > 1 path on CPU: world_size-2 gloo processes shard frames and all-gather the pose records."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    import cpe_amd
    from cpe_amd import dist as D
    r, l, w = D.init_from_env('gloo')
    lo, hi = D.shard_range(total, r, w)
    rec = torch.zeros((hi - lo, 16), dtype=torch.float64)
    rec[:, 0] = torch.arange(lo, hi, dtype=torch.float64)          # frame id in slot 0
    rec[:, 15] = r
    D.barrier()
    allrec = D.gather_records(rec, total)
    t = D.max_over_ranks(float(r + 1), 'cpu')
    q.put((r, allrec[:, 0].tolist(), allrec[:, 15].tolist(), t))


def test_shard_and_gather_world2():
    sys.path.insert(0, ROOT)
    import cpe_amd
    from cpe_amd import dist as D
    assert D.shard_range(10, 0, 4) == (0, 3) and D.shard_range(10, 3, 4) == (9, 10)
    assert D.shard_range(4096, 7, 8) == (3584, 4096)
    assert D.shard_range(3, 3, 4) == (3, 3)                          # ragged: an empty shard
    total, world = 11, 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in ps: p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in ps: p.join(timeout=60)
    for r, ids, owners, t in res:
        assert ids == [float(i) for i in range(total)]
        assert owners == [0.0] * 6 + [1.0] * 5
        assert t == 2.0


def test_pack_counters_roundtrip():
    sys.path.insert(0, ROOT)
    import cpe_amd
    from cpe_amd import pipeline as P
    n = torch.tensor([0, 180, 2048]); it = torch.tensor([1, 283, 99999]); fs = torch.tensor([0, 5, 0])
    dl = torch.tensor([0, 2, 6]); dr = torch.tensor([1, 0, 3])
    got = P.unpack_counters(P.pack_counters(n, it, fs, dl, dr))
    for a, b in zip(got, (n, it, fs, dl, dr)):
        assert a.tolist() == b.tolist()


def _bench(*argv, env=None):
    import json
    import subprocess
    e = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *argv], env=e, capture_output=True, timeout=300)
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith('{')]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr.decode()


def test_bench_starts_its_own_ranks_strong_scaling():
    """`python bench.py --gpus 2` from a bare shell (no WORLD_SIZE): the parent starts two ranks, each owns its
    dist.shard_range block of THE batch, the records of all frames come back in order (gloo, stub pipeline)"""
    rc, out, err = _bench('--gpus', '2', '--stub', '--frames', '11', '--steps', '2', '--warmup', '1')
    assert rc == 0, err
    assert out['n_gpus'] == 2 and out['scaling'] == 'strong' and out['steps'] == 2
    assert out['config']['frames_total'] == 11 and out['config']['frames_per_gpu'] == 6
    assert out['rows'] == 11 and out['checksum'] == float(sum(range(11)))
    assert out['all_gather_bytes'] == 11 * 128
    rc1, one, err1 = _bench('--gpus', '1', '--stub', '--frames', '11', '--steps', '2', '--warmup', '1')
    assert rc1 == 0, err1
    assert one['payload_sum'] == out['payload_sum']          # same batch whatever the number of ranks
    assert one['scaling'] == 'weak' and one['n_gpus'] == 1


def test_bench_weak_scaling_and_rank_failure():
    rc, out, err = _bench('--gpus', '2', '--stub', '--frames', '5', '--scaling', 'weak', '--steps', '1', '--warmup', '0')
    assert rc == 0, err
    assert out['scaling'] == 'weak' and out['config']['frames_total'] == 10 and out['rows'] == 10
    # a world size that contradicts --gpus is refused (torchrun form with the wrong -nproc)
    rc, out, err = _bench('--gpus', '2', '--stub', env=dict(WORLD_SIZE='1', RANK='0', LOCAL_RANK='0'))
    assert rc != 0 and out is None


def test_launcher_fails_fast_when_a_rank_dies():
    """rank 1 exits before the rendezvous: the launcher ends rank 0 (which would wait for it for minutes) and returns
    non-zero quickly; no rank is left behind"""
    import time
    t0 = time.time()
    rc, out, err = _bench('--gpus', '2', '--stub', '--frames', '8', '--steps', '1', '--warmup', '0', env=dict(CPE_BENCH_FAIL_RANK='1'))
    assert rc != 0 and out is None
    assert 'rank(s) failed' in err and time.time() - t0 < 60


def test_launcher_gives_every_rank_one_device():
    """HIP_VISIBLE_DEVICES of rank r = the r-th entry of the parent's list (or r); the stub ranks keep the parent's view"""
    sys.path.insert(0, ROOT)
    import subprocess
    import bench
    seen = []
    real = subprocess.Popen

    class FakeProc:
        def __init__(self, cmd, env=None, stdout=None):
            seen.append((env.get('RANK'), env.get('HIP_VISIBLE_DEVICES'), env.get('CPE_BENCH_DEVICE')))
            self.stdout = open(os.devnull, 'rb')
        def poll(self): return 0
        def wait(self, timeout=None): return 0
        def terminate(self): pass
        def kill(self): pass
    old_env = os.environ.get('HIP_VISIBLE_DEVICES')
    try:
        subprocess.Popen = FakeProc
        os.environ['HIP_VISIBLE_DEVICES'] = '4,6'
        assert bench.launch_ranks([], 2) == 0
        os.environ.pop('HIP_VISIBLE_DEVICES')
        assert bench.launch_ranks([], 2) == 0
        assert bench.launch_ranks([], 2, one_device_each=False) == 0
    finally:
        subprocess.Popen = real
        if old_env is not None:
            os.environ['HIP_VISIBLE_DEVICES'] = old_env
    assert seen[:2] == [('0', '4', '0'), ('1', '6', '0')] and seen[2:4] == [('0', '0', '0'), ('1', '1', '0')]
    assert seen[4][2] is None and seen[5][2] is None


def test_bench_under_torchrun_form():
    """the driver's form: ranks started by torch.distributed.run, RANK / WORLD_SIZE in the environment"""
    import json
    import subprocess
    e = dict(os.environ)
    p = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
                        os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--stub', '--frames', '9', '--steps', '1', '--warmup', '0'],
                       env=e, capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    out = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith('{')][-1])
    assert out['n_gpus'] == 2 and out['rows'] == 9 and out['checksum'] == 36.0


def test_bench_batch_does_not_depend_on_the_sharding():
    """strong scaling shards THE batch: the frames a rank renders for its block are the frames of the one-rank run"""
    sys.path.insert(0, ROOT)
    import bench
    import cpe_amd
    from cpe_amd import synth
    old = bench.H, bench.W
    bench.H, bench.W = 96, 128
    try:
        full = bench.make_frames(synth, torch, 0, 20, 6, 'cpu')
        parts = [bench.make_frames(synth, torch, lo, hi, 6, 'cpu') for lo, hi in ((0, 7), (7, 14), (14, 20))]
        assert torch.equal(torch.cat([p[1] for p in parts]), full[1]) and torch.equal(torch.cat([p[2] for p in parts]), full[2])
        assert not torch.equal(full[1][6], full[1][0])            # frame 6 = scene 0 with its own noise
    finally:
        bench.H, bench.W = old
