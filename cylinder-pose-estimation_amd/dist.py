"""Multi-GPU: one process per GPU, frames sharded in contiguous blocks, no data-path collective; a single
all_gather of the fixed-size pose records (128 B per frame) collects the result on every rank.
torch.distributed backend "nccl" is RCCL on ROCm (xGMI between the 8 GPUs of a node); "gloo" for CPU tests.
The gather moves 512 KiB for 4096 frames: latency-bound, far below a single xGMI link's bandwidth."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as set by torch.distributed.run"""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def shard_range(total, rank, world):
    """contiguous block of ceil(total/world) frames per rank (the last ranks may get fewer)"""
    per = (total + world - 1) // world
    lo = min(total, rank * per)
    return lo, min(total, lo + per)


def gather_records(rec, total=None):
    """rec: [f_local, 16] f64 on this rank -> [total, 16] on every rank (ranks padded to equal length)"""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return rec
    world = dist.get_world_size()
    n_local = torch.tensor([rec.shape[0]], dtype=torch.int64, device=rec.device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local)
    counts = [int(c.item()) for c in counts]
    per = max(counts)
    pad = torch.zeros((per, rec.shape[1]), dtype=rec.dtype, device=rec.device)
    pad[:rec.shape[0]] = rec
    out = torch.empty((world * per, rec.shape[1]), dtype=rec.dtype, device=rec.device)
    dist.all_gather_into_tensor(out, pad)
    parts = [out[r * per:r * per + counts[r]] for r in range(world)]
    res = torch.cat(parts)
    return res if total is None else res[:total]


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value, device):
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
