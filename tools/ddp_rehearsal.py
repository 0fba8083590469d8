#!/usr/bin/env python3
"""Rehearsal of the data-parallel training step with N ranks sharing ONE GPU (gloo moves the buckets through the host): checks that
the bucketed, stream-overlapped gradient exchange of somi_amd.ddp.GradBuckets gives exactly the sum of the per-rank gradients and
that every rank ends a step with identical weights.  Launch:
  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/ddp_rehearsal.py
(the real multi-GPU run uses backend nccl = RCCL, one GPU per rank: bench.py --gpus N)."""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo-somi_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from somi_amd.configs import HYP_VISDRONE, SOMI_ANCHORS, fill_state, somi_cfg, synthetic_batch  # noqa: E402
from somi_amd.loss import ComputeLoss  # noqa: E402
from somi_amd.model import Model  # noqa: E402
from somi_amd.train import TrainStep  # noqa: E402


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo')
    dev = torch.device('cuda:0')
    cfg = somi_cfg(0.25, 0.33, anchors=SOMI_ANCHORS)
    model = fill_state(Model(cfg), 1 + rank).to(dev)             # different weights per rank: the broadcast must fix that
    shadow = copy.deepcopy(model)                                # plain per-rank replica for the reference gradients
    tr = TrainStep(model, dict(HYP_VISDRONE), 2, dist=dist, bucket_mb=1)     # 1 MB buckets: several per buffer at this size
    with torch.no_grad():                                        # the replica takes the broadcast weights too
        for p, q in zip(shadow.parameters(), model.parameters()):
            p.copy_(q)
        for p, q in zip(shadow.buffers(), model.buffers()):
            p.copy_(q)
    shadow.train()
    shadow.hyp = dict(HYP_VISDRONE)
    imgs, targets = synthetic_batch(2, 64, seed=10 + rank)
    imgs, targets = imgs.to(dev), targets.to(dev)

    # reference: local gradients of this rank's shard, summed over ranks with one blocking all-reduce per parameter
    loss_s, _ = ComputeLoss(shadow)(shadow(imgs), targets)
    (loss_s * world).backward()
    want = []
    for p in shadow.parameters():
        g = p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        want.append(g)

    # product path, stopped before the optimizer: forward, loss, reverse walk with bucketed exchange
    tr.buckets.reset()
    loss, _ = tr.compute_loss(model(imgs), targets)
    (loss * world).backward()
    tr.buckets.finish()
    torch.cuda.synchronize()
    worst = 0.0
    for (n, p), g in zip(model.named_parameters(), want):
        err = (p.grad - g).abs().max().item()
        worst = max(worst, err / (g.abs().max().item() * 1e-5 + 1e-7))
    nb = len(tr.buckets.launched)
    assert worst <= 1.0, f'rank {rank}: bucketed gradients differ from the per-parameter all-reduce ({worst:.1f} x tolerance)'
    assert nb > len(tr.optimizer.flat_grads), f'expected several buckets, got {nb}'
    tr.optimizer.zero_grad()

    # two full steps: identical weights on every rank afterwards
    for _ in range(2):
        loss, items = tr.step(imgs, targets)
    torch.cuda.synchronize()
    for buf in tr.optimizer.flat_params:
        mine = buf.detach().clone()
        ref = mine.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(mine, ref), f'rank {rank}: weights diverged from rank 0'
    assert torch.isfinite(loss).all()
    dist.barrier()
    if rank == 0:
        print(f'ddp rehearsal ok: world {world}, {nb} buckets, gradient error {worst:.2f} x tolerance, loss {float(loss):.4f}')
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
