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
    n = torch.tensor([0, 180, 1024]); it = torch.tensor([1, 283, 99999]); fs = torch.tensor([0, 5, 0])
    dl = torch.tensor([0, 2, 6]); dr = torch.tensor([1, 0, 3])
    got = P.unpack_counters(P.pack_counters(n, it, fs, dl, dr))
    for a, b in zip(got, (n, it, fs, dl, dr)):
        assert a.tolist() == b.tolist()
