"""Multi-GPU: one process per GPU, frames sharded in contiguous blocks, no data-path collective; a single
all_gather of the fixed-size pose records (128 B per frame) collects the result on every rank.
torch.distributed backend "nccl" is RCCL on ROCm (xGMI between the 8 GPUs of a node); "gloo" for CPU tests
and for rehearsals on a one-GPU box (records are staged through host memory there).
The gather moves 512 KiB for 4096 frames: latency-bound, far below a single xGMI link's bandwidth."""
import datetime
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, device_index=None):
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as set by torch.distributed.run (or by bench.py's own
    launcher).  device_index: GPU of this rank (default LOCAL_RANK)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local if device_index is None else device_index)   # the communicator binds to the current device
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(minutes=10))
    return rank, local, world


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()


def shard_range(total, rank, world):
    """contiguous block of ceil(total/world) frames per rank (the last ranks may get fewer)"""
    per = (total + world - 1) // world
    lo = min(total, rank * per)
    return lo, min(total, lo + per)


def _host_staged():
    """gloo moves host memory: device tensors are staged (rehearsal path; RCCL takes device pointers)"""
    return dist.get_backend() == 'gloo'


def gather_records(rec, total=None):
    """rec: [f_local, 16] f64 on this rank -> [total, 16] on every rank.  Shards of dist.shard_range are equal except
    for the tail, so every rank contributes ceil(total/world) rows (zero padded) and ONE all_gather_into_tensor moves
    them; without `total` the shard lengths are exchanged first."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return rec
    world = dist.get_world_size()
    dev = rec.device
    stage = _host_staged() and dev.type != 'cpu'
    if total is not None:
        per = (total + world - 1) // world
        counts = [max(0, min(total, (r + 1) * per) - min(total, r * per)) for r in range(world)]
        if rec.shape[0] != counts[dist.get_rank()]:       # not a shard_range block (weak scaling: equal blocks)
            counts = None
    else:
        counts = None
    if counts is None:
        n_local = torch.tensor([rec.shape[0]], dtype=torch.int64, device='cpu' if stage else dev)
        cl = [torch.zeros_like(n_local) for _ in range(world)]
        dist.all_gather(cl, n_local)
        counts = [int(c.item()) for c in cl]
        per = max(counts)
    src = rec.cpu() if stage else rec
    if src.shape[0] != per:
        pad = torch.zeros((per, rec.shape[1]), dtype=rec.dtype, device=src.device)
        pad[:src.shape[0]] = src
        src = pad
    out = torch.empty((world * per, rec.shape[1]), dtype=rec.dtype, device=src.device)
    dist.all_gather_into_tensor(out, src.contiguous())
    if all(c == per for c in counts):
        res = out
    else:
        res = torch.cat([out[r * per:r * per + counts[r]] for r in range(world)])
    if stage:
        res = res.to(dev)
    return res if total is None else res[:total]


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == 'nccl':
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def max_over_ranks(value, device):
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return float(value)
    stage = _host_staged()
    t = torch.tensor([value], dtype=torch.float64, device='cpu' if stage else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
