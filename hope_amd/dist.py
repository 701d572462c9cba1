"""Multi-GPU support for the scene-parallel path: one process per GPU, torch.distributed over RCCL
(backend "nccl" on ROCm; "gloo" in CPU tests).

The env step itself needs NO collective: independent scenes are split into contiguous blocks, one per
rank.  The reference has no distributed code at all; the only two exchange points a
data-parallel HOPE run needs are outside the step:
  * `allreduce_gradients` -- ONE fused flat bucket per optimiser step (actor 3.64 MB (+ critic): a ring
    all-reduce over point-to-point xGMI is latency-bound at this size, so fewer, larger messages win);
  * `gather_eval_stats`   -- one all_gather of 16 B per evaluated scene (status, steps, reward, path length),
    the bookkeeping of src/evaluation/eval_utils.py:57-84.
"""
import os

import torch
import torch.distributed as dist


def shard_range(n_total, rank, world):
    """contiguous block [lo, hi) of rank's scenes; blocks differ by at most one scene."""
    base, rem = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_from_env(backend=None):
    """reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run).  Returns (rank, world, local_rank)."""
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend)
    return rank, world, local_rank


def allreduce_gradients(params, average=True, group=None):
    """sum (or mean) the .grad of every parameter across ranks through ONE flat buffer."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat /= dist.get_world_size(group)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n
    return flat.numel() * flat.element_size()


def gather_eval_stats(status, steps, reward, path_len, group=None):
    """per-scene evaluation records from every rank, concatenated in rank order on every rank.
    status/steps: int32 [n_local]; reward/path_len: float32 [n_local] (n_local may differ by one)."""
    rec = torch.stack([status.to(torch.float32), steps.to(torch.float32), reward.to(torch.float32),
                       path_len.to(torch.float32)], dim=1).contiguous()
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return rec
    world = dist.get_world_size(group)
    n = torch.tensor([rec.shape[0]], device=rec.device, dtype=torch.int64)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    nmax = int(max(int(s.item()) for s in sizes))
    pad = torch.zeros((nmax, 4), device=rec.device, dtype=rec.dtype)
    pad[:rec.shape[0]] = rec
    out = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return torch.cat([o[:int(s.item())] for o, s in zip(out, sizes)], dim=0)


def success_rate(records):
    """fraction of ARRIVED (status 2) episodes, as eval_utils.py:75-77 reports."""
    return float((records[:, 0] == 2).float().mean().item()) if len(records) else 0.0
