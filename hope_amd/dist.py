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


def local_world_size():
    """ranks on THIS node (torch.distributed.run exports LOCAL_WORLD_SIZE; a single process counts as 1)"""
    return max(1, int(os.environ.get('LOCAL_WORLD_SIZE', os.environ.get('WORLD_SIZE', 1))))


def pin_rank_to_cores(local_rank=None, local_world=None):
    """Give this rank ITS share of the host: with one process per GPU every rank runs host-side helpers -- the native scene
    generator (hope_scenegen_generate), the pool refresher's thread, the OpenMP CPU baseline of bench.py -- whose default
    fan-out is "the CPUs I may run on".  Unpinned, 8 ranks x 256 hardware threads oversubscribe a 256-thread host eightfold.
    The rank's CPU affinity becomes a contiguous block of the CPUs the process may use now (NUMA-friendly for GPUs enumerated in
    socket order), OMP_NUM_THREADS / torch's intra-op pool follow.  HOPE_NO_PIN=1 disables it.  Returns the number of CPUs."""
    if local_rank is None:
        local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if local_world is None:
        local_world = local_world_size()
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:                      # not Linux
        return os.cpu_count() or 1
    if local_world <= 1 or os.environ.get('HOPE_NO_PIN') == '1' or len(cpus) < local_world:
        return len(cpus)
    per = len(cpus) // local_world
    mine = cpus[local_rank * per:(local_rank + 1) * per]
    os.sched_setaffinity(0, mine)
    os.environ['OMP_NUM_THREADS'] = str(len(mine))
    try:
        torch.set_num_threads(max(1, min(len(mine), 16)))
    except Exception:
        pass
    return len(mine)


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


_AR = {'calls': 0, 'bytes': 0, 'events': [], 'ms': 0.0}


def _single(group=None):
    """True when there is nobody to exchange with.  HOPE_DIST_FORCE=1 sends a one-rank group through the backend anyway (the
    1-GPU RCCL test: library, device kernels and IPC environment exercised without a second GPU)."""
    if not dist.is_initialized():
        return True
    return dist.get_world_size(group) == 1 and os.environ.get('HOPE_DIST_FORCE') != '1'


def allreduce_stats(reset=False):
    """fused gradient all-reduces since the last reset: {'calls', 'bytes', 'ms'} -- ms from device events around the collective
    (CUDA tensors; 0 for gloo / CPU), resolved here, not in the hot loop"""
    for a, b in _AR['events']:
        b.synchronize()
        _AR['ms'] += a.elapsed_time(b)
    _AR['events'].clear()
    timed = _AR.get('timed_calls', 0)
    out = {'calls': _AR['calls'], 'timed_calls': timed, 'bytes': _AR['bytes'], 'ms': _AR['ms'],
           'ms_per_call': _AR['ms'] / timed if timed else None, 'bytes_per_call': _AR['bytes'] / _AR['calls'] if _AR['calls'] else None}
    if reset:
        _AR.update(calls=0, timed_calls=0, bytes=0, ms=0.0)
    return out


def allreduce_gradients(params, average=True, group=None):
    """sum (or mean) the .grad of every parameter across ranks through ONE flat buffer."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads or _single(group):
        return 0
    flat = torch.cat([g.reshape(-1) for g in grads])
    ev = None
    if flat.is_cuda and len(_AR['events']) < 4096:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if ev is not None:
        ev[1].record()
        _AR['events'].append(ev)
        _AR['timed_calls'] = _AR.get('timed_calls', 0) + 1          # (ms_per_call divides by THIS count: untimed calls beyond the cap are not in the ms)
    _AR['calls'] += 1
    _AR['bytes'] += flat.numel() * flat.element_size()
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
    if _single(group):
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
