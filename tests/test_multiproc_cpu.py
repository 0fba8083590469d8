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


def _free_port():
    """A port nobody listens on right now (a fixed pid-derived port can meet another process's, or a socket still in TIME_WAIT)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def test_two_rank_sharding_and_timing():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
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


def _ddp_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, 'yolo-somi_amd'))
    from somi_amd.ddp import GradBuckets
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    # two "parameter groups"; layers own increasing ranges: layer 0 -> [0,300), 1 -> [300,700), 2 -> [700,1000) of buffer 0
    g0 = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    g1 = torch.ones(10) * (rank + 1)
    offs = [{0: 0, 1: 300, 2: 700}, {0: 0, 2: 4}]
    gb = GradBuckets([g0, g1], offs, dist=dist, bucket_bytes=256 * 4, use_streams=False)
    order = []
    for layer in (2, 1, 0):                                  # the reverse walk
        gb.layer_done(layer)
        order.append(len(gb.launched))
    gb.finish()
    stats = torch.arange(6, dtype=torch.float32) + 10 * rank  # "BatchNorm running statistics" that drifted apart between the ranks
    gb.broadcast_buffers(stats)                               # DDP's broadcast_buffers (train.py:208-209): rank 0's overwrite everyone's
    assert torch.equal(stats, torch.arange(6, dtype=torch.float32)), stats
    gb.broadcast_buffers(torch.empty(0))                      # a graph without buffers: no collective, no hang
    res = (rank, order, list(gb.launched), g0.tolist(), g1.tolist())   # plain lists: a tensor in the queue is fetched from THIS process, which may have exited by then
    secs, nbytes = gb.measure_exchange(iters=2)              # what bench.py reports as `allreduce` at N > 1 (collective)
    assert secs > 0 and nbytes == 1010 * 4 and float(g0.abs().sum()) == 0.0 and gb.launched == []
    q.put(res)
    dist.destroy_process_group()


def test_gradient_buckets_all_reduce_sum_two_ranks():
    """DDP semantics of train.py:208-209,266-267 (mean x WORLD_SIZE == SUM) with buckets launched as layers finish."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, order, launched, g0, g1 in res:
        assert g0 == (torch.arange(1000, dtype=torch.float32) * 3).tolist()      # (1 + 2) x the base gradient on every rank
        assert g1 == [3.0] * 10
        # buffer 0 buckets from the end: [744,1000) after layer 2; [488,744) after layer 1 (>= 300); [232,488) and [0,232) after layer 0
        b0 = [(s, e) for bi, s, e in launched if bi == 0]
        assert b0 == [(744, 1000), (488, 744), (232, 488), (0, 232)]
        assert order[0] >= 1 and order[1] > order[0] and order[2] > order[1]      # something new is launched after every layer
        covered = sorted(b0)
        assert covered[0][0] == 0 and all(covered[i][1] == covered[i + 1][0] for i in range(len(covered) - 1)) and covered[-1][1] == 1000
