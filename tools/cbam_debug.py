#!/usr/bin/env python3
"""Stage-by-stage comparison of ONE C2fCBAM layer's backward against the fp64 oracle on tensors captured from a whole-graph pass (diagnostic;
the CPU oracle is the checker).  For every bottleneck: d(t*ca*sa) entering the attention pair, dca, the pooled-path vectors, the gradient leaving it.
usage: cbam_debug.py [--layer 2] [--size 320] [--batch 4] [--gamma 0.25] [--seed 2] [--dcn 0]"""
import argparse
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo-somi_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

from tests.isolate import capture, nhwc  # noqa: E402


def rel(a, b):
    b = b.double()
    return (a.double().cpu() - b).abs().max().item() / (b.abs().max().item() + 1e-300)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--layer', type=int, default=2)
    ap.add_argument('--size', type=int, default=320)
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--nc', type=int, default=10)
    ap.add_argument('--gamma', type=float, default=0.25)
    ap.add_argument('--seed', type=int, default=2)
    ap.add_argument('--dcn', type=int, default=0)
    ap.add_argument('--warm', type=int, default=0, help='1: a whole-graph HIP training pass first (recycled allocator blocks, stale contexts)')
    a = ap.parse_args()
    from oracle.somi_ref import Model as OModel
    from oracle.somi_ref.loss import ComputeLoss as OLoss
    from oracle.somi_ref.testing import SOMI_ANCHORS, fill_state, somi_cfg, synthetic_batch, HYP_VISDRONE
    from somi_amd import blocks as B
    from somi_amd import ops
    from somi_amd.model import Model
    torch.set_num_threads(16)
    cfg = somi_cfg(1.0, 1.0, nc=a.nc, anchors=SOMI_ANCHORS, dcn=bool(a.dcn))
    ref = fill_state(OModel(cfg), a.seed)
    with torch.no_grad():
        for m_ in ref.modules():
            if isinstance(m_, torch.nn.BatchNorm2d):
                m_.weight.mul_(a.gamma)
    mine = Model(cfg)
    mine.load_state_dict(ref.state_dict())
    ref.hyp = dict(HYP_VISDRONE)
    ref64 = copy.deepcopy(ref).double().train()
    imgs, targets = synthetic_batch(a.batch, a.size, nc=a.nc, seed=14)

    def run64():
        l64, _ = OLoss(ref64)(ref64(imgs.double() / 255), targets.double())
        l64.backward()
    rec = capture(ref64, run64, layers={a.layer})[a.layer]
    blk = ref64.model[a.layer]
    n = len(blk.m)
    keep = {}

    def keep_out(name):
        def f(m, inp, out):
            out.retain_grad()
            keep[name] = out
        return f

    def keep_in(name):
        def f(m, inp):
            inp[0].retain_grad()
            keep[name] = inp[0]
        return f
    hs = []
    for i, bt in enumerate(blk.m):
        hs.append(bt.cv1.register_forward_hook(keep_out(f't{i}')))
        hs.append(bt.channel_attention.register_forward_hook(keep_out(f'ca{i}')))
        hs.append(bt.spatial_attention.register_forward_hook(keep_out(f'sa{i}')))
        hs.append(bt.cv2.register_forward_pre_hook(keep_in(f't2_{i}')))

        def mlp_in(m, inp, i=i):
            inp[0].retain_grad()
            keep.setdefault(f'mlpin{i}', []).append(inp[0])
        hs.append(bt.channel_attention.shared_MLP.register_forward_pre_hook(mlp_in))
    for p in blk.parameters():
        p.grad = None
    x = rec['x'][0].double().requires_grad_(True)
    y = blk(x)
    y.backward(rec['dy'][0].double())
    for h in hs:
        h.remove()
    # HIP side, instrumented
    log = {'cbam': [], 'mlp': [], 'pool': [], 'bn_pooled': []}
    real_cbam, real_mlp, real_pool, real_bn = ops.cbam_backward, ops.attn_mlp_backward, ops.pool_backward_add_, ops.bn_act_backward

    def cbam(dt2, t, t_coff, c_, ca, sa, stats, w7, k_, dw7, db7, t_max=None):
        before = dt2.clone()
        ins = dict(ca=ca.clone(), sa=sa.clone(), stats=stats.clone(), w7=w7.clone())
        out = real_cbam(dt2, t, t_coff, c_, ca, sa, stats, w7, k_, dw7, db7, t_max=t_max)
        # the same chain with torch ops on the device, from the very tensors the kernels were given
        import torch.nn.functional as F
        v = t * ca[:, None, None, :]
        dlogit = (before * v).sum(-1) * sa * (1 - sa)
        amaxc = v.argmax(-1)
        wk = w7.permute(2, 0, 1).unsqueeze(0)                      # [k][k][2] -> (1,2,k,k): the forward conv's weight
        dstats = F.conv_transpose2d(dlogit.unsqueeze(1), wk, padding=k_ // 2)       # (B,2,H,W)
        dt1 = before * sa[..., None] + dstats[:, 0][..., None] / c_
        dt1.scatter_add_(3, amaxc[..., None], dstats[:, 1][..., None])
        dca_t = (dt1 * t).sum((1, 2))
        ins.update(dca_torch=dca_t, dt_torch=dt1 * ca[:, None, None, :])
        log['cbam'].append(dict(dt2=before, t=t.clone(), dca=out[0].clone(), amaxp=out[1].clone(), dt=dt2.clone(), **ins))
        return out

    def mlp(mode, dout, out, avg, mx, *r, **k):
        res = real_mlp(mode, dout, out, avg, mx, *r, **k)
        if mode == 0:
            log['mlp'].append(dict(davg=res[0].clone(), dmax=res[1].clone(), avg=avg.clone(), mx=mx.clone()))
        return res

    def pool(dt, *r, **k):
        res = real_pool(dt, *r, **k)
        log['pool'].append(dt.clone())
        return res
    ops.cbam_backward, ops.attn_mlp_backward, ops.pool_backward_add_ = cbam, mlp, pool
    mine = mine.cuda().train()
    if a.warm:
        from somi_amd.loss import ComputeLoss
        mine.hyp = dict(HYP_VISDRONE)
        ComputeLoss(mine)(mine(imgs.cuda()), targets.cuda())[0].backward()
        torch.cuda.synchronize()
        for p_ in mine.parameters():
            p_.grad = None
        log = {k_: [] for k_ in log}
    mb = mine.model[a.layer]
    mb(B.Act(nhwc(rec['x'][0]).cuda()))
    mb.backward(B.Act(nhwc(rec['dy'][0]).cuda()))
    torch.cuda.synchronize()
    ops.cbam_backward, ops.attn_mlp_backward, ops.pool_backward_add_ = real_cbam, real_mlp, real_pool
    print(f'layer {a.layer} ({n} bottlenecks), hidden width {blk.c}; relative errors against the fp64 oracle block (same fp32 inputs)')
    for j, i in enumerate(reversed(range(n))):                    # the backward walks the bottlenecks last to first
        c = log['cbam'][j]
        t64, ca64, t2 = keep[f't{i}'], keep[f'ca{i}'], keep[f't2_{i}']
        HW = t64.shape[2] * t64.shape[3]
        amax64 = t64.detach().reshape(t64.shape[0], t64.shape[1], -1).argmax(2)
        line = [f'm.{i}:', f't {rel(c["t"].permute(0, 3, 1, 2), t64.detach()):.1e}', f'd(t ca sa) {rel(c["dt2"].permute(0, 3, 1, 2), t2.grad):.1e}',
                f'dca {rel(c["dca"], ca64.grad.reshape(ca64.shape[0], -1)):.1e}',
                f'argmax_p differs at {(c["amaxp"].cpu().long() != amax64).sum().item()} of {amax64.numel()}']
        line.append(f'| inputs of the chain: ca {rel(c["ca"], ca64.detach().reshape(ca64.shape[0], -1)):.1e} sa {rel(c["sa"], keep[f"sa{i}"].detach()[:, 0]):.1e}'
                    f' | kernels vs torch ops on the same device tensors: dca {rel(c["dca"], c["dca_torch"].cpu()):.1e} dt {rel(c["dt"], c["dt_torch"].cpu()):.1e}')
        if j < len(log['mlp']):
            ml = log['mlp'][j]
            line.append(f'avg {rel(ml["avg"], t64.detach().mean((2, 3))):.1e} max {rel(ml["mx"], t64.detach().amax((2, 3))):.1e}')
            line.append(f'davg {rel(ml["davg"], keep[f"mlpin{i}"][0].grad):.1e} dmax {rel(ml["dmax"], keep[f"mlpin{i}"][1].grad):.1e}')
        if j < len(log['pool']):
            line.append(f'dt after the pooled add {rel(log["pool"][j].permute(0, 3, 1, 2), t64.grad):.1e}')
        print('  ' + '  '.join(line))
    # where the dca error of the worst bottleneck sits, and whether the inputs differ by a per-channel OFFSET (the sums over pixels see that)
    worst = max(range(n), key=lambda j_: rel(log['cbam'][j_]['dca'], keep[f'ca{n - 1 - j_}'].grad.reshape(log['cbam'][j_]['dca'].shape)))
    i = n - 1 - worst
    c = log['cbam'][worst]
    d64 = keep[f'ca{i}'].grad.reshape(c['dca'].shape)
    err = (c['dca'].cpu().double() - d64)
    top = err.abs().flatten().topk(5)
    print(f'  m.{i}: largest dca errors (b, c, hip, fp64): ' + ', '.join(
        f'({int(ix) // err.shape[1]}, {int(ix) % err.shape[1]}, {c["dca"].flatten()[int(ix)].item():+.4e}, {d64.flatten()[int(ix)].item():+.4e})' for ix in top.indices))
    t64 = keep[f't{i}'].detach()
    dt_off = (c['t'].permute(0, 3, 1, 2).cpu().double() - t64)
    g_off = (c['dt2'].permute(0, 3, 1, 2).cpu().double() - keep[f't2_{i}'].grad)
    print(f'  m.{i}: t: max |per-(b,c) mean of (hip - fp64)| {dt_off.mean((2, 3)).abs().max().item():.2e} against max |hip - fp64| {dt_off.abs().max().item():.2e}, max |t| {t64.abs().max().item():.2e}; '
          f'd(t ca sa): mean offset {g_off.mean((2, 3)).abs().max().item():.2e} against max diff {g_off.abs().max().item():.2e}, max |.| {keep[f"t2_{i}"].grad.abs().max().item():.2e}')
    sabs = (keep[f't{i}'].grad.abs() * t64.abs()).sum((2, 3))
    print(f'  m.{i}: dca is a sum of 6400 terms per (b,c): max |dca| {d64.abs().max().item():.3e}, typical sum |terms| {sabs.median().item():.3e}')
    g64 = {n_: p_.grad for n_, p_ in blk.named_parameters() if p_.grad is not None}
    bad = [(n_, f'{rel(p_.grad, g64[n_]):.1e}') for n_, p_ in mb.named_parameters() if p_.grad is not None and rel(p_.grad, g64[n_]) > 1e-4]
    print(f'  parameter gradients beyond 1e-4 relative: {bad}')


if __name__ == '__main__':
    main()
