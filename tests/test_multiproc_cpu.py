"""World-size-2 gloo test (CPU) of the N>1 path: batch sharding, barrier + max-over-ranks timing and the whole-job rate
bench.py reports.  The data path itself has no collective (replicas only)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, 'yolo-somi_amd'))
    from somi_amd.dist import shard_range, timed_steps, whole_job_rate
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    lo, hi = shard_range(11, rank, world)
    import time
    calls = []

    def step():
        calls.append(1)
        time.sleep(0.02 * (rank + 1))                      # rank 1 is the slow one

    dt = timed_steps(step, 3, dist=dist)
    rate = whole_job_rate(4, 3, world, dt)
    q.put((rank, lo, hi, len(calls), dt, rate))
    dist.destroy_process_group()


def test_two_rank_sharding_and_timing():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, lo0, hi0, n0, dt0, rate0), (_, lo1, hi1, n1, dt1, rate1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 6, 6, 11)             # every image exactly once, contiguous slices
    assert n0 == n1 == 3                                     # exactly K steps on every rank
    assert abs(dt0 - dt1) < 1e-9 and dt0 >= 3 * 0.04 - 1e-3  # MAX over ranks: both report the slow rank's time
    assert abs(rate0 - 4 * 3 * 2 / dt0) < 1e-9 and rate0 == rate1


def test_shard_range_covers_everything():
    sys.path.insert(0, os.path.join(ROOT, 'yolo-somi_amd'))
    from somi_amd.dist import shard_range
    for n in (0, 1, 7, 32, 257):
        for w in (1, 2, 3, 8):
            seen = []
            for r in range(w):
                lo, hi = shard_range(n, r, w)
                seen += list(range(lo, hi))
            assert seen == list(range(n))
