#!/usr/bin/env python3
"""Rehearsal of the data-parallel training step with N ranks sharing ONE GPU (gloo moves the buckets through the host).

Checks, against torch DDP's semantics restated with plain collectives (train.py:208-209,266-267: the loss is multiplied by
WORLD_SIZE and DDP MEANs the gradients, so every rank steps on sum_r dL_r/dw):
  1. the bucketed, stream-overlapped exchange of somi_amd.ddp.GradBuckets leaves exactly  mean_r( d(WORLD_SIZE * L_r)/dw )  in
     every rank's gradient buffers;
  2. gradient accumulation over 2 micro-batches exchanges once and gives the sum of both micro-batches over all ranks;
  3. rank 0, alone, stepping on every rank's shard in turn (accumulate = world, no collectives) reaches the same gradients and
     the same weights after the optimizer step as the N-rank step did;
  4. every rank ends with bit-identical weights, and the BatchNorm running statistics start from rank 0's.
Launch:
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


def ddp_expected(shadow, batches, world):
    """What torch DDP leaves in .grad after the micro-batches `batches` (train.py:264-270): per micro-batch the local gradient of
    loss * WORLD_SIZE is added to .grad and the buffers are all-reduced with MEAN."""
    for p in shadow.parameters():
        p.grad = None
    crit = ComputeLoss(shadow)
    for imgs, targets in batches:
        for name, buf in shadow.named_buffers():                 # DDP's broadcast_buffers (default on, train.py:208-209): rank 0's running
            if buf.dtype.is_floating_point and 'running' in name:    # statistics before every forward - they are the pivot the batch statistics are
                dist.broadcast(buf, src=0)                           # summed around, so the last bits of the gradients depend on them
        loss, _ = crit(shadow(imgs), targets)
        (loss * world).backward()
        for p in shadow.parameters():
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            dist.all_reduce(p.grad, op=dist.ReduceOp.SUM)
            p.grad /= world
    return [p.grad.detach().clone() for p in shadow.parameters()]


def worst_error(model, want):
    worst = 0.0
    for p, g in zip(model.parameters(), want):
        worst = max(worst, (p.grad - g).abs().max().item() / (g.abs().max().item() * 2e-5 + 1e-7))
    return worst


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo')
    dev = torch.device('cuda:0')
    cfg = somi_cfg(0.25, 0.33, anchors=SOMI_ANCHORS)
    model = fill_state(Model(cfg), 1 + rank).to(dev)             # different weights AND statistics per rank: the broadcast must fix that
    tr = TrainStep(model, dict(HYP_VISDRONE), 2, dist=dist, bucket_mb=1)     # 1 MB buckets: several per buffer at this size
    for buf in (*tr.optimizer.flat_params, tr.optimizer.flat_buffers):       # 4. one set of weights and BN statistics after init
        ref = buf.detach().clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(buf, ref), f'rank {rank}: initial state differs from rank 0'
    start = copy.deepcopy(model.state_dict())
    shadow = Model(cfg).to(dev)                                  # plain per-rank replica for the expected gradients
    shadow.load_state_dict(start)
    shadow.train()
    shadow.hyp = dict(HYP_VISDRONE)
    shards = [[t.to(dev) for t in synthetic_batch(2, 64, seed=10 + r)] for r in range(world)]
    extra = [[t.to(dev) for t in synthetic_batch(2, 64, seed=50 + r)] for r in range(world)]
    imgs, targets = shards[rank]

    # 1. one micro-batch, stopped before the optimizer: forward, loss, reverse walk with the bucketed exchange
    want = ddp_expected(shadow, [shards[rank]], world)
    tr.buckets.reset()
    loss, _ = tr.compute_loss(model(imgs), targets)
    loss.backward()
    tr.buckets.finish()
    torch.cuda.synchronize()
    w1 = worst_error(model, want)
    nb = len(tr.buckets.launched)
    assert w1 <= 1.0, f'rank {rank}: bucketed gradients differ from DDP semantics ({w1:.1f} x tolerance)'
    assert nb > len(tr.optimizer.flat_grads), f'expected several buckets, got {nb}'
    tr.optimizer.zero_grad()

    # 2. accumulation over two micro-batches (running statistics restored first: step 1 advanced them)
    model.load_state_dict(start), shadow.load_state_dict(start)
    want2 = ddp_expected(shadow, [shards[rank], extra[rank]], world)
    tr.accumulate, tr._since_step = 2, 0
    hold = tr.optimizer.step
    tr.optimizer.step = lambda: None                             # keep the gradients for the comparison
    zg = tr.optimizer.zero_grad
    tr.optimizer.zero_grad = lambda *a, **k: None
    tr.step(*shards[rank])
    assert tr.buckets.launched == [], 'a non-stepping micro-batch must not exchange'
    tr.step(*extra[rank])
    torch.cuda.synchronize()
    w2 = worst_error(model, want2)
    assert w2 <= 1.0, f'rank {rank}: accumulate=2 gradients differ from DDP semantics ({w2:.1f} x tolerance)'
    tr.optimizer.step, tr.optimizer.zero_grad = hold, zg
    tr.optimizer.zero_grad()
    tr.accumulate, tr._since_step = 1, 0

    # 3. one full N-rank step from the common start, against rank 0 alone accumulating every shard
    model.load_state_dict(start)
    tr.optimizer.reset_ema()
    loss, items = tr.step(imgs, targets)
    torch.cuda.synchronize()
    after = [b.detach().clone() for b in tr.optimizer.flat_params]
    w3 = 0.0
    if rank == 0:
        solo = Model(cfg).to(dev)
        solo.load_state_dict(start)
        st = TrainStep(solo, dict(HYP_VISDRONE), 2, dist=None, accumulate=world)
        grads = None
        for i, (im, tg) in enumerate(shards):
            if i == world - 1:                                   # capture the accumulated gradients right before the step
                real = st.optimizer.step
                def spy():                                       # noqa: E306
                    nonlocal grads
                    grads = [g.detach().clone() for g in st.optimizer.flat_grads]
                    real()
                st.optimizer.step = spy
            st.step(im, tg)
        torch.cuda.synchronize()
        for a, b in zip(after, st.optimizer.flat_params):
            d = (a - b).abs()
            w3 = max(w3, torch.quantile(d[:: max(1, d.numel() // 1_000_000)].float(), 0.999).item() / (3e-4 * 1e-2))
        assert w3 <= 1.0, f'weights after one {world}-rank step differ from the single-process run on all shards ({w3:.1f} x tolerance)'
        assert grads is not None
    for buf in tr.optimizer.flat_params:                          # 4.
        ref = buf.detach().clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(buf, ref), f'rank {rank}: weights diverged from rank 0'
    assert torch.isfinite(loss).all()
    # 5. broadcast_buffers (DDP's default, train.py:208-209): step 3 ran on different shards, so the ranks' BatchNorm running statistics have
    #    drifted apart; the next step must start every rank from rank 0's.  Same batch on every rank -> same weights, same data, same starting
    #    statistics -> bit-identical statistics afterwards (without the broadcast they would keep their offset).
    before = tr.optimizer.flat_buffers.detach().clone()
    ref = before.clone()
    dist.broadcast(ref, src=0)
    drift = torch.tensor([0.0 if torch.equal(before, ref) else 1.0])
    dist.all_reduce(drift)
    assert drift.item() >= 1.0, 'the ranks were expected to hold different running statistics before the broadcast (the check would be vacuous)'
    tr.step(*shards[0])
    torch.cuda.synchronize()
    after_b = tr.optimizer.flat_buffers.detach().clone()
    ref = after_b.clone()
    dist.broadcast(ref, src=0)
    assert torch.equal(after_b, ref), f'rank {rank}: running statistics differ from rank 0 after a broadcast_buffers step'
    dist.barrier()
    if rank == 0:
        print(f'ddp rehearsal ok: world {world}, {nb} buckets, gradient error {w1:.2f} / accumulate-2 {w2:.2f} x tolerance, '
              f'weights vs single process {w3:.2f} x tolerance, loss {float(loss):.4f}')
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
