"""Multi-process helpers for the data-parallel hot path (one process per GPU, torch.distributed: "nccl" == RCCL on ROCm).

Inference / NMS / WBF shard by image with no data-path collective ("replicas only"): each rank takes a
contiguous slice of the global batch; the only communication is the barrier + max-over-ranks of the timed region that
bench.py's contract asks for.  The gradient all-reduce of the training path (train.py:208-209) is ddp.GradBuckets.
"""
import time

import torch


def shard_range(n_items, rank, world):
    """Contiguous [lo, hi) slice of n_items for this rank; remainders go to the lowest ranks (DistributedSampler-like
    coverage without padding: every item is processed exactly once)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def timed_steps(step, steps, dist=None, sync=None, device=None):
    """Runs `step()` exactly `steps` times between two (barrier + device sync) fences; returns the MAX over ranks of the
    wall time in seconds (bench.py contract)."""
    sync = sync or (lambda: None)
    sync()
    if dist is not None:
        dist.barrier()
    t0 = time.time()
    for _ in range(steps):
        step()
    sync()
    dt = time.time() - t0
    if dist is not None:
        dist.barrier()
        t = torch.tensor([dt], dtype=torch.float64, device=device or 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def whole_job_rate(units_per_rank_per_step, steps, world, seconds):
    """Aggregate throughput over all ranks (weak scaling: per-rank work fixed)."""
    return units_per_rank_per_step * steps * world / seconds
