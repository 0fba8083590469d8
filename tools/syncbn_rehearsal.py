#!/usr/bin/env python3
"""--sync-bn rehearsal (reference train.py:165-167: torch.nn.SyncBatchNorm.convert_sync_batchnorm) with N ranks sharing ONE GPU (gloo).

SyncBatchNorm is BatchNorm over the concatenation of every rank's batch, so the checks compare the N-rank run on the shards of a
batch with a one-process run on the WHOLE batch:
  1. a Conv -> BN -> SiLU chain under ops.SYNC_BN: every rank's outputs / input gradients are its slice of the CPU oracle's
     (oracle/somi_ref blocks, plain torch) results on the whole batch, the running statistics are the whole batch's, and the
     parameter gradients summed over the ranks are the whole batch's;
  2. TrainStep(sync_bn=True) on the SOMI graph (SGD, so that the weight update is linear in the gradient): one step on the two
     half-batches leaves the weights and BatchNorm buffers that one process stepping on the whole batch gets (the halves carry the
     same label set, so that the per-rank loss normalisation - means over the rank's own targets - adds up to the whole batch's),
     bit-identical on all ranks.
Launch:
  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29551 tools/syncbn_rehearsal.py"""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo-somi_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn as nn  # noqa: E402


def nhwc(t):
    return t.detach().permute(0, 2, 3, 1).contiguous().float()


def close(a, b, what, rel=2e-4, atol=0.0):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    err, ref = (a - b).abs().max().item(), b.abs().max().item()
    assert err <= rel * ref + atol, f'{what}: max |diff| {err:.3e} vs scale {ref:.3e}'


def conv_chain(rank, world, dev):
    from oracle.somi_ref import blocks as OB
    from oracle.somi_ref.testing import fill_state
    from somi_amd import blocks as MB
    from somi_amd import ops
    g = torch.Generator().manual_seed(11)
    cfgs = [(16, 32, 3, 1), (32, 48, 3, 2), (48, 20, 1, 1)]
    ref = nn.Sequential(*[OB.Conv(*c) for c in cfgs])
    fill_state(ref, 4)
    OB.initialize_weights(ref)
    mine = nn.Sequential(*[MB.Conv(*c) for c in cfgs])
    mine.load_state_dict(ref.state_dict())
    for m in mine.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.eps, m.momentum = 1e-3, 0.03
    mine = mine.to(dev).train()
    ref.train()
    # the ranks' running means (the pivots their partial sums are taken around) have drifted apart, as they do after the first steps of a
    # real run: the exchanged records are pivot-free, so the statistics over all ranks must not care (rank 0 keeps the oracle's buffers)
    with torch.no_grad():
        for m in mine:
            m.bn.running_mean += 0.37 * rank
    per = 2
    x = torch.randn(per * world, 16, 14, 10, generator=g, requires_grad=True)
    y = ref(x)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    lo, hi = rank * per, (rank + 1) * per
    ops.SYNC_BN = dist
    try:
        a = MB.Act(nhwc(x[lo:hi]).to(dev))
        for m in mine:
            a = m(a)
        d = MB.Act(nhwc(dy[lo:hi]).to(dev))
        for m in reversed(list(mine)):
            d = m.backward(d)
    finally:
        ops.SYNC_BN = None
    close(a.t[..., :20], nhwc(y[lo:hi]), 'sync forward (own slice of the whole batch)')
    for m, r in zip(mine, ref):
        if rank == 0:
            close(m.bn.running_mean, r.bn.running_mean, 'running_mean over all ranks')
        close(m.bn.running_var, r.bn.running_var, 'running_var over all ranks (unbiased over the global count)')
    close(d.t, nhwc(x.grad[lo:hi]), 'sync dx (own slice)')
    for (n, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        if q.grad is None:
            continue
        tot = p.grad.detach().clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        close(tot, q.grad, f'sum over ranks of d{n}', atol=2e-5)


def whole_model(rank, world, dev):
    from somi_amd.configs import HYP_VISDRONE, SOMI_ANCHORS, fill_state, somi_cfg, synthetic_batch
    from somi_amd.model import Model
    from somi_amd.train import TrainStep
    cfg = somi_cfg(0.25, 0.33, anchors=SOMI_ANCHORS)
    per = 2
    imgs, targets = synthetic_batch(per * world, 96, seed=5)
    t0 = targets[targets[:, 0] < per]                             # the label set of the first shard, given to every shard
    targets = torch.cat([torch.cat([t0[:, :1] + r * per, t0[:, 1:]], 1) for r in range(world)])
    hyp = dict(HYP_VISDRONE)
    base = fill_state(Model(cfg), 2)
    one = TrainStep(copy.deepcopy(base).to(dev), hyp, per * world, adam=False)             # one process, whole batch
    w0 = [b.clone() for b in one.optimizer.flat_params]
    one.step(imgs.to(dev), targets.to(dev))
    shard = TrainStep(copy.deepcopy(base).to(dev), hyp, per, dist=dist, adam=False, sync_bn=True)
    mine_t = targets[(targets[:, 0] >= rank * per) & (targets[:, 0] < (rank + 1) * per)].clone()
    mine_t[:, 0] -= rank * per
    shard.step(imgs[rank * per:(rank + 1) * per].to(dev), mine_t.to(dev))
    for a, b, c in zip(shard.optimizer.flat_params, one.optimizer.flat_params, w0):
        close(a - c, b - c, 'weight update: 2 ranks with --sync-bn vs one process on the whole batch', rel=2e-3, atol=1e-9)
    close(shard.optimizer.flat_buffers, one.optimizer.flat_buffers, 'BatchNorm buffers', rel=1e-4)
    for buf in list(shard.optimizer.flat_params) + [shard.optimizer.flat_buffers]:
        got = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(got, buf)
        assert all(torch.equal(got[0], t) for t in got[1:]), 'ranks differ after a --sync-bn step'
    # without --sync-bn the running statistics are those of the rank's own shard: they must differ between ranks (the check has teeth)
    plain = TrainStep(copy.deepcopy(base).to(dev), hyp, per, dist=dist, adam=False)
    plain.step(imgs[rank * per:(rank + 1) * per].to(dev), mine_t.to(dev))
    got = [torch.empty_like(plain.optimizer.flat_buffers) for _ in range(world)]
    dist.all_gather(got, plain.optimizer.flat_buffers)
    assert not torch.equal(got[0], got[1]), 'rank-local statistics expected without --sync-bn'


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo')
    dev = torch.device('cuda:0')
    conv_chain(rank, world, dev)
    whole_model(rank, world, dev)
    dist.barrier()
    if rank == 0:
        print('sync-bn rehearsal ok')
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
