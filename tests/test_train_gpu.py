"""Training-mode kernels and block backward passes on the MI355X against torch autograd on the CPU (the oracle blocks in
train mode).  Bar 1e-3 relative."""
import math

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ACTS = {'none': lambda v: v, 'silu': F.silu, 'gelu': F.gelu, 'relu': F.relu}


def rel_close(got, want, rel=1e-3, what='', atol=0.0):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = (got - want).abs().max().item()
    scale = want.abs().max().item() + 1e-12
    assert err <= rel * scale + atol, f'{what}: max err {err:.3e} vs scale {scale:.3e}'


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize('order,act,batch_stats', [(0, 'silu', True), (1, 'gelu', True), (0, 'silu', False), (0, 'none', True)])
def test_bn_act_forward_backward(order, act, batch_stats):
    from somi_amd import ops
    g = torch.Generator().manual_seed(order * 10 + len(act))
    B, C, H, W = 3, 24, 13, 9
    x = (torch.randn(B, C, H, W, generator=g) * 1.5 + 0.3).requires_grad_(True)
    bn = nn.BatchNorm2d(C, eps=1e-3, momentum=0.03)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g) * 0.2)
        bn.running_mean.copy_(torch.randn(C, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()
    bn.train(batch_stats)
    z = ACTS[act](bn(x)) if order == 0 else bn(ACTS[act](x))
    dz = torch.randn(z.shape, generator=g)
    z.backward(dz)
    d = torch.device('cuda')
    wide = torch.zeros(B, H, W, C + 8)
    wide[..., 4:4 + C] = nhwc(x.detach())
    xd = wide.to(d)
    gam, bet = bn.weight.detach().to(d), bn.bias.detach().to(d)
    if batch_stats:
        rm, rv = rm0.to(d), rv0.to(d)
        src = xd if order == 0 else ops.chan_affine_act(xd, C, 4, torch.ones(C, device=d), torch.zeros(C, device=d), act, 0,
                                                        torch.zeros_like(xd), 4)
        mean, rstd, scale, shift = ops.bn_stats(src, C, 4, gam, bet, 1e-3, 0.03, rm, rv)
        rel_close(rm, bn.running_mean, what='running_mean')
        rel_close(rv, bn.running_var, what='running_var')
    else:
        mean = rm0.to(d)
        rstd = (1.0 / torch.sqrt(rv0 + 1e-3)).to(d)
        scale = gam * rstd
        shift = bet - mean * scale
    out = torch.zeros(B, H, W, C, device=d)
    ops.chan_affine_act(xd, C, 4, scale, shift, act, order, out)
    rel_close(out, nhwc(z), what='forward')
    dgam, dbet = torch.zeros(C, device=d), torch.zeros(C, device=d)
    dx = torch.zeros(B, H, W, C, device=d)
    ops.bn_act_backward(nhwc(dz).to(d), 0, xd, 4, C, mean, rstd, scale, shift, act, order, batch_stats, dx, 0, dgam, dbet)
    rel_close(dx, nhwc(x.grad), what='dx')
    rel_close(dgam, bn.weight.grad, what='dgamma')
    rel_close(dbet, bn.bias.grad, what='dbeta')


@pytest.mark.parametrize('npix,C', [(1, 4), (7, 12), (353, 1028), (5000, 64), (70001, 256), (33, 2048)])
def test_elementwise_sweeps_cover_every_pixel_and_channel(npix, C):
    """The sweeps' thread -> (channel quad, pixel walk) mapping at awkward sizes: a single pixel, C/4 that does not divide 256, more channel quads
    than a workgroup has threads, pixel counts that leave a tail behind the four-deep groups, grids at the resident-round cap; in place too.
    Against the formulas in fp64."""
    from somi_amd import ops
    g = torch.Generator().manual_seed(npix + C)
    d = torch.device('cuda')
    x = torch.randn(1, 1, npix, C, generator=g)
    dz = torch.randn(1, 1, npix, C, generator=g)
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    res = torch.randn(1, 1, npix, C, generator=g)
    u = x.double() * sc.double() + sh.double()
    want = u * torch.sigmoid(u) + res.double()
    xd, resd = x.to(d), res.to(d)
    out = ops.chan_affine_act(xd, C, 0, sc.to(d), sh.to(d), 'silu', 0, torch.full_like(xd, 9.0), 0, residual=resd, res_coff=0)
    rel_close(out, want.float(), rel=1e-5, what='affine + silu + residual')
    inplace = xd.clone()
    ops.chan_affine_act(inplace, C, 0, sc.to(d), sh.to(d), 'silu', 0, inplace, 0)
    rel_close(inplace, (u * torch.sigmoid(u)).float(), rel=1e-5, what='affine + silu in place')
    relu_out = ops.chan_affine_act(xd, C, 0, sc.to(d), sh.to(d), 'relu', 1, torch.empty_like(xd), 0)          # the run-time switch form
    rel_close(relu_out, (torch.relu(x.double()) * sc.double() + sh.double()).float(), rel=1e-5, what='relu then affine')
    both = ops.add_(xd.clone(), 0, resd, 0, C)
    assert torch.equal(both.cpu(), x + res)
    # backward with frozen statistics: dx = scale * dz * silu'(u)
    mean, rstd = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    gam = torch.rand(C, generator=g) + 0.5
    scale = gam * rstd
    shift = -mean * scale
    u2 = x.double() * scale.double() + shift.double()
    sg = torch.sigmoid(u2)
    want_dx = scale.double() * dz.double() * (sg * (1 + u2 * (1 - sg)))
    dgam, dbet = torch.zeros(C, device=d), torch.zeros(C, device=d)
    dx = ops.bn_act_backward(dz.to(d), 0, xd, 0, C, mean.to(d), rstd.to(d), scale.to(d), shift.to(d), 'silu', 0, False,
                             torch.full_like(xd, 9.0), 0, dgam, dbet)
    rel_close(dx, want_dx.float(), rel=1e-5, what='frozen-statistics backward dx')
    dact = dz.double() * (sg * (1 + u2 * (1 - sg)))
    rel_close(dbet, dact.sum((0, 1, 2)).float(), rel=1e-4, what='dbeta')
    rel_close(dgam, (dact * (x.double() - mean.double()) * rstd.double()).sum((0, 1, 2)).float(), rel=1e-4, what='dgamma')


def _grads_close(mine, ref, what):
    for (n, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        if q.grad is None:
            continue
        assert p.grad is not None, f'{what}: {n} has no gradient'
        # atol: gradients that are analytically zero (e.g. a bias in front of a batch-norm) are rounding noise on both sides
        rel_close(p.grad, q.grad, what=f'{what}: d{n}', atol=2e-5)


def test_conv_block_train_forward_backward():
    """Conv (conv -> BN batch stats -> SiLU) chain in training mode: outputs, running stats, dx and all parameter gradients."""
    from oracle.somi_ref import blocks as OB
    from oracle.somi_ref.testing import fill_state
    from somi_amd import blocks as MB
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
    mine = mine.cuda().train()
    ref.train()
    x = torch.randn(3, 16, 14, 10, generator=g, requires_grad=True)
    y = ref(x)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    a = MB.Act(nhwc(x.detach()).cuda())
    for m in mine:
        a = m(a)
    rel_close(a.t[..., :20], nhwc(y), what='train forward')
    for m, r in zip(mine, ref):
        rel_close(m.bn.running_mean, r.bn.running_mean, what='running_mean')
        rel_close(m.bn.running_var, r.bn.running_var, what='running_var')
    d = MB.Act(nhwc(dy).cuda())
    for m in reversed(list(mine)):
        d = m.backward(d)
    rel_close(d.t, nhwc(x.grad), what='dx')
    _grads_close(mine, ref, 'conv chain')


def _run_block_train(mine, ref, x, seed, what, cin):
    from somi_amd import blocks as MB
    g = torch.Generator().manual_seed(seed)
    for m in mine.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.eps, m.momentum = 1e-3, 0.03
    mine = mine.cuda().train()
    ref.train()
    x = x.clone().requires_grad_(True)
    y = ref(x)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    out = mine(MB.Act(nhwc(x.detach()).cuda()))
    rel_close(out.t[..., out.coff:out.coff + out.c], nhwc(y), what=f'{what} forward')
    dx = mine.backward(MB.Act(nhwc(dy).cuda()))
    rel_close(dx.t[..., :cin], nhwc(x.grad), what=f'{what} dx')
    _grads_close(mine, ref, what)


@pytest.mark.parametrize('c1,c2,n,shortcut', [(32, 32, 2, True), (48, 32, 1, False), (32, 32, 3, False), (64, 32, 3, True)])
def test_c2fcbam_train_forward_backward(c1, c2, n, shortcut):
    from oracle.somi_ref import blocks as OB
    from oracle.somi_ref.testing import fill_state
    from somi_amd import blocks as MB
    ref = fill_state(OB.C2fCBAM(c1, c2, n, shortcut), 6)
    OB.initialize_weights(ref)
    mine = MB.C2fCBAM(c1, c2, n, shortcut)
    mine.load_state_dict(ref.state_dict())
    x = torch.randn(2, c1, 12, 10, generator=torch.Generator().manual_seed(c1))
    _run_block_train(mine, ref, x, 21, 'C2fCBAM', c1)


@pytest.mark.parametrize('c,n,B,H,W', [(64, 2, 3, 37, 29), (256, 1, 2, 20, 20), (32, 1, 5, 45, 31)])
def test_cbam_step_c_inside_the_batchnorm_backward_equals_the_three_pass_form(c, n, B, H, W, monkeypatch):
    """somi_cbam_bn_bwd_reduce_f32 / _apply_f32 (step C of the CBAM backward rebuilt in registers inside the first conv's BatchNorm + SiLU backward)
    against the form it replaces (somi_cbam_bwd_chan_f32 writing dt, then the pooled BatchNorm backward): same block, same tensors, every parameter
    gradient and the input gradient.  Ragged maps (chunks that end inside an image, arg-max pixels in the last chunk), hidden widths 32 / 128 / 16;
    both forms are separately held to the oracle by test_c2fcbam_train_forward_backward."""
    from somi_amd import blocks as MB, ops
    g = torch.Generator().manual_seed(c + H)
    blk = MB.C2fCBAM(c, c, n, True)
    with torch.no_grad():
        for p_ in blk.parameters():
            p_.copy_(torch.randn(p_.shape, generator=g) * (0.5 if p_.dim() < 2 else (2.0 / max(1, p_[0].numel())) ** 0.5))
        for m in blk.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.weight.copy_(0.5 + torch.rand(m.weight.shape, generator=g))
    blk = blk.cuda().train()
    x = torch.randn(B, H, W, c, generator=g).cuda()
    dy = torch.randn(B, H, W, c, generator=g).cuda()
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(ops, 'CBAM_FUSED_BN', fused)
        for p_ in blk.parameters():
            p_.grad = None
        blk(MB.Act(x.clone()))
        dx = blk.backward(MB.Act(dy.clone()))
        torch.cuda.synchronize()
        res[fused] = {'dx': dx.t[..., :c].clone(), **{n_: p_.grad.clone() for n_, p_ in blk.named_parameters()}}
    assert set(res[True]) == set(res[False])
    for k in res[True]:
        a, b = res[True][k].double(), res[False][k].double()
        err = (a - b).abs().max().item() / (b.abs().max().item() + 1e-30)
        assert err < 5e-5, f'{k}: fused vs three-pass {err:.2e}'


def test_sppf_seam_train_forward_backward():
    from oracle.somi_ref import blocks as OB
    from oracle.somi_ref.testing import fill_state
    from somi_amd import blocks as MB
    for name, ctor_o, ctor_m, cin in (('SPPF', lambda: OB.SPPF(32, 32, 5), lambda: MB.SPPF(32, 32, 5), 32),
                                      ('SEAM', lambda: OB.SEAM(32, 32, 1, 16), lambda: MB.SEAM(32, 32, 1, 16), 32)):
        ref = fill_state(ctor_o(), 8)
        OB.initialize_weights(ref)
        mine = ctor_m()
        mine.load_state_dict(ref.state_dict())
        x = torch.randn(2, cin, 11, 9, generator=torch.Generator().manual_seed(3))
        _run_block_train(mine, ref, x, 31, name, cin)


def test_bifpn_train_backward():
    from oracle.somi_ref import blocks as OB
    from somi_amd import blocks as MB
    g = torch.Generator().manual_seed(17)
    ref = OB.BiFPN(3)
    with torch.no_grad():
        ref.weight.copy_(torch.tensor([0.7, 1.3, 0.4]))
    mine = MB.BiFPN(3)
    mine.load_state_dict(ref.state_dict())
    mine = mine.cuda().train()
    lo = torch.randn(2, 16, 5, 6, generator=g, requires_grad=True)
    a = torch.randn(2, 16, 10, 12, generator=g, requires_grad=True)
    b = torch.randn(2, 16, 10, 12, generator=g, requires_grad=True)
    y = ref([F.interpolate(lo, scale_factor=2, mode='nearest'), a, b])
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    up = MB.Upsample(None, 2, 'nearest')
    out = mine([up(MB.Act(nhwc(lo.detach()).cuda())), MB.Act(nhwc(a.detach()).cuda()), MB.Act(nhwc(b.detach()).cuda())])
    rel_close(out.t, nhwc(y), what='bifpn forward')
    ds = mine.backward(MB.Act(nhwc(dy).cuda()))
    for d, r, nm in zip(ds, (lo, a, b), ('up', 'a', 'b')):
        rel_close(d.t, nhwc(r.grad), what=f'bifpn d{nm}')
    rel_close(mine.weight.grad, ref.weight.grad, what='bifpn dweight')


def test_decoupled_detect_train_backward():
    from oracle.somi_ref import blocks as OB
    from oracle.somi_ref.testing import fill_state
    from somi_amd import blocks as MB
    g = torch.Generator().manual_seed(23)
    anchors = [list(range(8))] * 2
    ref = fill_state(OB.DecoupledDetect(10, anchors, (64, 96)), 9)
    OB.initialize_weights(ref)
    mine = MB.DecoupledDetect(10, anchors, (64, 96))
    mine.load_state_dict(ref.state_dict())
    for m in mine.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.eps, m.momentum = 1e-3, 0.03
    mine = mine.cuda().train()
    ref.train()
    ref.stride = mine.stride = torch.tensor([8., 16.])
    xs = [torch.randn(2, 64, 8, 8, generator=g, requires_grad=True), torch.randn(2, 96, 4, 4, generator=g, requires_grad=True)]
    ys = ref(list(xs))
    dys = [torch.randn(y.shape, generator=g) for y in ys]
    torch.autograd.backward(ys, dys)
    outs = mine([MB.Act(nhwc(x.detach()).cuda()) for x in xs])
    for o, y in zip(outs, ys):
        rel_close(o, y, what='detect raw')
    dxs = mine.backward([d.cuda() for d in dys])
    for d, x in zip(dxs, xs):
        rel_close(d.t[..., :x.shape[1]], nhwc(x.grad), what='detect dx')
    _grads_close(mine, ref, 'DecoupledDetect')


def _train_cfg(odconv):
    from oracle.somi_ref.testing import somi_cfg, SOMI_ANCHORS
    cfg = somi_cfg(0.25, 0.33, anchors=SOMI_ANCHORS)
    if not odconv:                                           # the graph walk without the dynamic-conv layers
        for sect in ('backbone', 'head'):
            for l in cfg[sect]:
                if l[2] == 'ODConv_3rd':
                    l[2], l[3] = 'Conv', [l[3][0], 3, 2]
    return cfg


def _anchored_check(named_hip, named_cpu32, g64, sums, what, k_cpu=2.0):
    """Every parameter gradient against the fp64 oracle under the shared bar (oracle.somi_ref.testing.fp64_anchored_errors): BASELINE's flat 1e-3
    of the gradient's scale - or, where the fp32 CPU oracle itself is farther than that from fp64, k_cpu x the CPU path's own distance - plus a
    FIXED 16 fp32 roundings of the gradient's own terms (matters for cancelling sums only).  Nothing in the bar depends on the HIP path."""
    from oracle.somi_ref.testing import fp64_anchored_errors
    named_hip = list(named_hip)
    res = fp64_anchored_errors(named_hip, named_cpu32, g64, sums, k_cpu=k_cpu)
    missing = set(g64) - {t[0] for t in res}
    assert not missing and len(named_hip) == len(g64), f'{what}: parameters without a gradient / a measured term scale: {sorted(missing)[:6]}'
    bad = [(n, f'x{r:.1f}', f'hip {e:.2e}', f'cpu32 {ec:.2e}', f'scale {sc:.2e}') for n, r, e, ec, sc in res if not r <= 1.0]
    worst = max(res, key=lambda t: t[1])
    print(f'{what}: {len(res)} parameters, worst {worst[0]} at {worst[1]:.2f} of the bar (hip {worst[2]:.2e}, fp32 cpu {worst[3]:.2e}, scale {worst[4]:.2e})')
    assert not bad, f'{what}: {len(bad)} parameter gradients beyond max(1e-3 scale, {k_cpu} x fp32 CPU error) + 16 roundings of their terms: {bad[:8]}'


def _well_conditioned(model, gamma=0.25):
    """fill_state's BatchNorm scales are U(0.5, 1.5): with them the 38-layer graph is CHAOTIC - the fp32 CPU oracle's own gradients sit 2e-2 ...
    5e-2 (median, relative) from its fp64 run at any map size, a rounding is amplified ~10^6 x, and no per-parameter bar below O(1) means
    anything.  Scaling every BatchNorm gamma by 0.25 (pre-activations of +-0.3: SiLU in its near-linear range, contractive layers) puts the
    same graph in the regime a trained network lives in: fp32 CPU median 4e-6, 99 % of the parameters below 2e-4 of fp64 (measured, 320x320,
    batch 4).  Conv weights need no change - BatchNorm divides their scale out."""
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.weight.mul_(gamma)
    return model


@pytest.mark.parametrize('dcn', [False, True])
def test_full_width_well_conditioned_step_holds_flat_1e3(dcn):
    """one training step of the real yolov5l-SOMI widths (77.5 M parameters; head convs 256->177->98, 1024-wide SPPF, per-sample
    ODConv weights) at 320x320, batch 4 (ODConv's squeeze BatchNorm sees 4 samples), on a WELL-CONDITIONED fill (_well_conditioned), against
    the fp64 oracle.  The premise is asserted, not assumed: the fp32 CPU oracle must itself sit within 2e-4 of fp64 on 99 % of the parameters.
    Then the HIP path is held to BASELINE's flat 1e-3 on EVERY parameter (k_cpu = 0: no reference to any fp32 path's error) plus a fixed 16
    roundings of the gradient's own terms for the cancelling sums.  Both oracles are evaluated AT THE ARG-MAX DECISIONS OF THE HIP FORWARD
    (ForcedDecisions): this input has max decisions closer than fp32 rounding - the channel maximum of one pixel of model.2.m.1 (relative gap
    < 1e-6 of 25 600), SPPF windows at 2e-6 - where the gradient of the max jumps; the first version of this test met exactly those (HIP routed
    model.2.m.1's pixel to channel 38 where fp64 takes channel 20: every backbone parameter 2e-3 off, neck and head 1e-5) - tools/cbam_debug.py
    found it.  The decisions themselves are checked separately: the forward output and loss must match, and a decision may only differ from
    the oracle's own where the oracle's candidates are within 1e-4 of each other.  With the DCNv3 sites the offset branch's gradient is discontinuous
    wherever a sampling point crosses a pixel boundary (floor()), so ANY fp32 evaluation - the CPU oracle's too - is 1e-3 ... 4e-3 from fp64 on
    those parameters and the layers in front of them: there the bar is max(1e-3, 2 x the CPU oracle's own distance)."""
    import copy
    from oracle.somi_ref import Model as OModel
    from oracle.somi_ref.loss import ComputeLoss as OLoss
    from isolate import hip_decisions
    from oracle.somi_ref.testing import (SOMI_ANCHORS, AbsTermSums, ForcedDecisions, decision_disagreements, fill_state, max_decision_gaps,
                                         somi_cfg, synthetic_batch, HYP_VISDRONE)
    from somi_amd.loss import ComputeLoss
    from somi_amd.model import Model
    torch.set_num_threads(16)
    cfg = somi_cfg(1.0, 1.0, anchors=SOMI_ANCHORS, dcn=dcn)
    ref = _well_conditioned(fill_state(OModel(cfg), 2))
    mine = Model(cfg)
    mine.load_state_dict(ref.state_dict())
    ref.hyp = mine.hyp = dict(HYP_VISDRONE)
    ref64 = copy.deepcopy(ref).double()
    imgs, targets = synthetic_batch(4, 320, seed=14)
    ref.train(), ref64.train()
    mine = mine.cuda().train()
    pm = mine(imgs.cuda())
    table = hip_decisions(mine)                                  # which candidate every max-pool / arg-max of the HIP forward took
    lm, _ = ComputeLoss(mine)(pm, targets.cuda())
    lm.backward()
    torch.cuda.synchronize()
    gaps = max_decision_gaps(ref64, imgs.double() / 255)
    tight = sorted((v, k) for k, v in gaps.items() if v < 1e-5)
    ndiff, worst_gap = decision_disagreements(ref64, imgs.double() / 255, table)
    print(f'max decisions of this input closer than 1e-5 (relative) in the fp64 oracle: {[(k, f"{v:.1e}") for v, k in tight[:6]]} ({len(tight)} modules of '
          f'{len(gaps)}); the HIP forward decided {ndiff} of them differently, candidates at most {worst_gap:.1e} apart')
    assert worst_gap <= 1e-4, f'a HIP arg-max decision differs from the fp64 oracle\'s where the candidates are {worst_gap:.1e} apart: not a near-tie'
    with ForcedDecisions(ref64, table), AbsTermSums(ref64) as cond:
        l64, _ = OLoss(ref64)(ref64(imgs.double() / 255), targets.double())
        l64.backward()
    with ForcedDecisions(ref, table):
        l32, _ = OLoss(ref)(ref(imgs.float() / 255), targets)
        l32.backward()
    rel_close(lm, l64.detach().float(), rel=1e-5, what='loss')
    g64 = {n: p.grad for n, p in ref64.named_parameters() if p.grad is not None}
    g32 = [(n, p.grad) for n, p in ref.named_parameters() if p.grad is not None]
    rel_cpu = sorted((g.double() - g64[n]).abs().max().item() / (g64[n].abs().max().item() + 1e-300) for n, g in g32 if cond.sums[n].max() < 1e3 * g64[n].abs().max())
    q99 = rel_cpu[int(0.99 * len(rel_cpu))]
    print(f'well-conditioned fill, dcn={dcn}: fp32 CPU oracle vs fp64: median {rel_cpu[len(rel_cpu) // 2]:.2e} q99 {q99:.2e} max {rel_cpu[-1]:.2e} '
          f'({len(rel_cpu)} parameters that are not cancelling sums)')
    if not dcn:
        assert q99 <= 2e-4, f'the premise of this test - a fill on which the fp32 CPU path is within 2e-4 of fp64 - does not hold: q99 {q99:.2e}'
    _anchored_check(((n, p.grad) for n, p in mine.named_parameters() if p.grad is not None), g32, g64, cond.sums,
                    f'full width @320 batch 4, well-conditioned, dcn={dcn}', k_cpu=2.0 if dcn else 0.0)


@pytest.mark.parametrize('odconv', [False, True])
def test_whole_model_train_step_gradients(odconv):
    """loss.backward() through the whole SOMI graph on HIP: loss, train outputs and BN running statistics against the fp32 CPU oracle at 1e-3;
    every parameter gradient against the FP64 oracle under the shared bar: max(BASELINE's 1e-3 of the gradient's scale, 2 x the fp32 CPU
    oracle's own distance from fp64) + 16 roundings of the gradient's own terms - no absolute floor."""
    import copy
    from oracle.somi_ref import Model as OModel
    from oracle.somi_ref.loss import ComputeLoss as OLoss
    from oracle.somi_ref.testing import AbsTermSums, fill_state, synthetic_batch, HYP_VISDRONE
    from somi_amd.loss import ComputeLoss
    from somi_amd.model import Model
    cfg = _train_cfg(odconv)
    ref = fill_state(OModel(cfg), 2)
    mine = Model(cfg)
    mine.load_state_dict(ref.state_dict())
    ref.hyp = mine.hyp = dict(HYP_VISDRONE)
    ref64 = copy.deepcopy(ref).double().train()
    imgs, targets = synthetic_batch(2, 64, seed=1)
    ref.train()
    pr = ref(imgs.float() / 255)
    lr, ir = OLoss(ref)(pr, targets)
    lr.backward()
    with AbsTermSums(ref64) as cond:
        l64, _ = OLoss(ref64)(ref64(imgs.double() / 255), targets.double())
        l64.backward()
    mine = mine.cuda().train()
    pm = mine(imgs.cuda())
    for a, b in zip(pm, pr):
        rel_close(a, b, what='train outputs')
    lm, im = ComputeLoss(mine)(pm, targets.cuda())
    rel_close(lm, lr, rel=1e-4, what='loss')
    lm.backward()
    _anchored_check(((n, p.grad) for n, p in mine.named_parameters() if p.grad is not None),
                    [(n, p.grad) for n, p in ref.named_parameters() if p.grad is not None],
                    {n: p.grad for n, p in ref64.named_parameters() if p.grad is not None}, cond.sums, f'small SOMI graph, odconv={odconv}')
    for (n, p), (_, q) in zip(mine.named_buffers(), ref.named_buffers()):
        if 'running' in n:
            rel_close(p, q, what=n)


@pytest.mark.parametrize('B', [3, 1])
def test_odconv_train_forward_backward(B):
    from oracle.somi_ref import blocks as OB
    from oracle.somi_ref.testing import fill_state
    from somi_amd import blocks as MB
    ref = fill_state(OB.ODConv_3rd(16, 32, 3, 2, 4), 12)
    OB.initialize_weights(ref)
    mine = MB.ODConv_3rd(16, 32, 3, 2, 4)
    mine.load_state_dict(ref.state_dict())
    x = torch.randn(B, 16, 12, 12, generator=torch.Generator().manual_seed(B))
    _run_block_train(mine, ref, x, 41 + B, f'ODConv B={B}', 16)


def test_fused_sgd_nesterov_ema_matches_torch():
    """The reference's other optimizer branch (train.py:138: SGD(momentum, nesterov=True), same three parameter groups) fused with the
    EMA update: four steps with a changing lr and momentum (the warm-up of train.py:250-256 ramps both) vs torch.optim.SGD."""
    import copy
    from somi_amd.optim import FusedAdamEMA, reference_param_groups
    g = torch.Generator().manual_seed(6)
    net = nn.Sequential(nn.Conv2d(4, 8, 3, bias=False), nn.BatchNorm2d(8), nn.Conv2d(8, 6, 1), nn.Linear(6, 5))
    ref = copy.deepcopy(net)
    ema_ref = copy.deepcopy(net)
    g0, g1, g2 = reference_param_groups(ref)
    opt = torch.optim.SGD(g0, lr=0.0032, momentum=0.843, nesterov=True)
    opt.add_param_group({'params': g1, 'weight_decay': 0.00036})
    opt.add_param_group({'params': g2})
    net = net.cuda()
    mine = FusedAdamEMA(net, lr=0.0032, betas=(0.843, 0.999), weight_decay=0.00036, sgd=True)
    assert all(pg['momentum'] == 0.843 and pg['nesterov'] for pg in mine.param_groups)
    for step in range(1, 5):
        for pg, qg in zip(opt.param_groups, mine.param_groups):
            pg['lr'] = qg['lr'] = 0.0032 * step
            pg['momentum'] = qg['momentum'] = 0.5 + 0.08 * step
        grads = [torch.randn(p.shape, generator=g) for p in ref.parameters()]
        for p, q, gr in zip(ref.parameters(), net.parameters(), grads):
            p.grad = gr.clone()
            q.grad.copy_(gr)
        opt.step()
        mine.step()
        d = 0.9999 * (1 - math.exp(-step / 2000))
        with torch.no_grad():
            msd = ref.state_dict()
            for k, v in ema_ref.state_dict().items():
                if v.dtype.is_floating_point:
                    v *= d
                    v += (1 - d) * msd[k].detach()
    for (n, p), (_, q) in zip(ref.named_parameters(), net.named_parameters()):
        rel_close(q, p, rel=1e-5, what=f'param {n}')
    esd = mine.ema_state_dict()
    for k, v in ema_ref.state_dict().items():
        if v.dtype.is_floating_point:
            rel_close(esd[k], v, rel=1e-5, what=f'ema {k}')


def test_fused_adam_ema_matches_torch():
    """Three optimizer steps of the fused Adam+EMA kernel vs torch.optim.Adam with the reference's parameter groups and
    the reference's ModelEMA rule (utils/torch_utils.py:331-345)."""
    import copy
    from somi_amd.optim import FusedAdamEMA, reference_param_groups
    g = torch.Generator().manual_seed(5)
    net = nn.Sequential(nn.Conv2d(4, 8, 3, bias=False), nn.BatchNorm2d(8), nn.Conv2d(8, 6, 1), nn.Linear(6, 5))
    ref = copy.deepcopy(net)
    ema_ref = copy.deepcopy(net)
    g0, g1, g2 = reference_param_groups(ref)
    opt = torch.optim.Adam(g0, lr=3e-4, betas=(0.843, 0.999))
    opt.add_param_group({'params': g1, 'weight_decay': 0.00036})
    opt.add_param_group({'params': g2})
    net = net.cuda()
    mine = FusedAdamEMA(net, lr=3e-4, betas=(0.843, 0.999), weight_decay=0.00036)
    for step in range(1, 4):
        for pg in opt.param_groups:
            pg['lr'] = 3e-4 * step
        for pg in mine.param_groups:
            pg['lr'] = 3e-4 * step
        grads = [torch.randn(p.shape, generator=g) for p in ref.parameters()]
        for p, q, gr in zip(ref.parameters(), net.parameters(), grads):
            p.grad = gr.clone()
            q.grad.copy_(gr)
        with torch.no_grad():
            ref[1].running_mean += 0.1 * step
            net[1].running_mean += 0.1 * step
        opt.step()
        mine.step()
        d = 0.9999 * (1 - math.exp(-step / 2000))
        with torch.no_grad():
            msd = ref.state_dict()
            for k, v in ema_ref.state_dict().items():
                if v.dtype.is_floating_point:
                    v *= d
                    v += (1 - d) * msd[k].detach()
    for (n, p), (_, q) in zip(ref.named_parameters(), net.named_parameters()):
        rel_close(q, p, rel=1e-5, what=f'param {n}')
    esd = mine.ema_state_dict()
    for k, v in ema_ref.state_dict().items():
        if v.dtype.is_floating_point:
            rel_close(esd[k], v, rel=1e-5, what=f'ema {k}')
    mine.zero_grad()
    assert all(float(q.grad.abs().max()) == 0.0 for q in net.parameters())


def test_fused_adam_packed_conv_masters():
    """Conv+BN blocks keep their weight master in the kernels' packing inside the flat optimizer buffer; the module parameter is
    a strided view of it.  Adam on that storage must equal torch.optim.Adam on the reference layout, pads must stay zero, and
    the EMA / state_dict round trip must come back in the reference layout."""
    import copy
    from somi_amd.blocks import Conv
    from somi_amd.optim import FusedAdamEMA, reference_param_groups
    from somi_amd.pack import pack_conv_weight, pad4
    g = torch.Generator().manual_seed(11)
    net = nn.Sequential(Conv(3, 8, 3), Conv(8, 6, 1), Conv(6, 12, 3, 2))       # Cin 3 -> 4, Cout 6 -> 8: both kinds of padding
    ref = copy.deepcopy(net)
    g0, g1, g2 = reference_param_groups(ref)
    opt = torch.optim.Adam(g0, lr=1e-3, betas=(0.9, 0.999))
    opt.add_param_group({'params': g1, 'weight_decay': 0.01})
    opt.add_param_group({'params': g2})
    net = net.cuda()
    mine = FusedAdamEMA(net, lr=1e-3, betas=(0.9, 0.999), weight_decay=0.01)
    assert all('_master' in m.__dict__ for m in net)
    for _ in range(3):
        grads = [torch.randn(p.shape, generator=g) for p in ref.parameters()]
        for p, q, gr in zip(ref.parameters(), net.parameters(), grads):
            p.grad = gr.clone()
            q.grad.copy_(gr)
        opt.step()
        mine.step()
    for (n, p), (_, q) in zip(ref.named_parameters(), net.named_parameters()):
        rel_close(q, p, rel=1e-5, what=f'param {n}')
    for m in net:
        c2, c1, k = m.conv.weight.shape[:3]
        want = pack_conv_weight(m.conv.weight.detach().cpu().contiguous(), cout_pad=pad4(c2))
        assert torch.equal(m.__dict__['_master'][0].cpu(), want), 'master storage is not the packed weight (or a pad moved)'
    sd = {k_: v.cpu() for k_, v in net.state_dict().items()}
    for k_, v in ref.state_dict().items():
        if v.dtype.is_floating_point:
            rel_close(sd[k_], v, rel=1e-5, what=f'state_dict {k_}')
    esd = mine.ema_state_dict()
    assert esd['0.conv.weight'].shape == ref[0].conv.weight.shape
    mine.zero_grad()
    assert all(float(q.grad.abs().max()) == 0.0 for q in net.parameters())


def test_train_step_with_optimizer_matches_oracle_adam():
    """Two whole TrainStep.step() calls on the small SOMI graph (tiny layers: single-split weight gradients accumulated into the
    packed gradient masters) against the CPU oracle driven by torch.optim.Adam with the reference's parameter groups.
    Adam's first updates are lr*sign(g) for every weight, so a weight whose gradient is rounding noise may move the other
    way: the bar is 2 steps x 2 lr absolute plus 1e-3 relative - layout or packing mistakes are orders of magnitude above it."""
    from oracle.somi_ref import Model as OModel
    from oracle.somi_ref.loss import ComputeLoss as OLoss
    from oracle.somi_ref.testing import fill_state, synthetic_batch, HYP_VISDRONE
    from somi_amd.model import Model
    from somi_amd.optim import reference_param_groups
    from somi_amd.train import TrainStep
    cfg = _train_cfg(False)
    ref = fill_state(OModel(cfg), 4)
    mine = Model(cfg)
    mine.load_state_dict(ref.state_dict())
    hyp = dict(HYP_VISDRONE)
    ref.hyp = hyp
    ref.train()
    imgs, targets = synthetic_batch(2, 64, seed=5)
    g0, g1, g2 = reference_param_groups(ref)
    wd = hyp['weight_decay'] * 2 * 32 / 64                       # train.py:121-123 at batch 2: accumulate = 32
    opt = torch.optim.Adam(g0, lr=3e-4, betas=(hyp['momentum'], 0.999))
    opt.add_param_group({'params': g1, 'weight_decay': wd})
    opt.add_param_group({'params': g2})
    tr = TrainStep(mine.cuda(), hyp, 2)
    crit = OLoss(ref)
    losses = []
    for _ in range(2):
        lr_, _ = crit(ref(imgs.float() / 255), targets)
        opt.zero_grad()
        lr_.backward()
        opt.step()
        lm, _ = tr.step(imgs.cuda(), targets.cuda())
        losses.append((float(lm), float(lr_.detach())))
    assert abs(losses[0][0] - losses[0][1]) <= 1e-4 * abs(losses[0][1]), losses
    assert abs(losses[1][0] - losses[1][1]) <= 1e-2 * abs(losses[1][1]), losses
    bad = []
    for (n, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        err = (p.detach().cpu() - q.detach()).abs().max().item()
        if err > 4 * 3e-4 + 1e-3 * q.detach().abs().max().item():
            bad.append((n, err))
    assert not bad, bad[:8]
    for (n, p), (_, q) in zip(mine.named_buffers(), ref.named_buffers()):
        if 'running' in n:
            rel_close(p, q, rel=2e-2, what=n)


def test_data_parallel_rehearsal_two_ranks_one_gpu():
    """tools/ddp_rehearsal.py: two ranks share this GPU (gloo), bucketed + stream-overlapped gradient exchange must equal the
    per-parameter all-reduce of the rank-local gradients and leave identical weights on both ranks."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', '29541', os.path.join(root, 'tools', 'ddp_rehearsal.py')], capture_output=True, text=True,
                       timeout=420, env=env, cwd=root)
    assert r.returncode == 0 and 'ddp rehearsal ok' in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_sync_bn_two_ranks_one_gpu():
    """tools/syncbn_rehearsal.py (--sync-bn, train.py:165-167): BatchNorm statistics over the batches of both ranks - a Conv/BN chain
    against the CPU oracle on the whole batch, and one TrainStep(sync_bn=True) on two half-batches against one process on the whole."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', '29551', os.path.join(root, 'tools', 'syncbn_rehearsal.py')], capture_output=True, text=True,
                       timeout=420, env=env, cwd=root)
    assert r.returncode == 0 and 'sync-bn rehearsal ok' in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_train_step_gradient_accumulation():
    """accumulate=2 (train.py:121,272): the first batch only accumulates gradients, the second steps the optimizer on the sum."""
    from somi_amd.configs import HYP_VISDRONE, SOMI_ANCHORS, fill_state, somi_cfg, synthetic_batch
    from somi_amd.model import Model
    from somi_amd.train import TrainStep
    model = fill_state(Model(somi_cfg(0.25, 0.33, anchors=SOMI_ANCHORS)), 3).cuda()
    tr = TrainStep(model, dict(HYP_VISDRONE), 2, accumulate=2)
    imgs, targets = synthetic_batch(2, 64, seed=8)
    w0 = [b.clone() for b in tr.optimizer.flat_params]
    tr.step(imgs.cuda(), targets.cuda())
    assert all(torch.equal(a, b) for a, b in zip(w0, tr.optimizer.flat_params)), 'weights moved before the accumulation was complete'
    g1 = [g.clone() for g in tr.optimizer.flat_grads]
    assert any(float(g.abs().max()) > 0 for g in g1)
    tr.step(imgs.cuda(), targets.cuda())
    assert any(not torch.equal(a, b) for a, b in zip(w0, tr.optimizer.flat_params)), 'no optimizer step after two batches'
    assert all(float(g.abs().max()) == 0 for g in tr.optimizer.flat_grads)


def test_end_to_end_training_reaches_map_and_matches_oracle():
    """tests/e2e_map.py: 300 product-path training steps on the synthetic rectangles task must actually learn it (mAP@0.5 > 0.8),
    and the product validation pipeline (HIP forward, NMS, matching, AP) must give the same mAP as the CPU oracle evaluating the
    same trained weights (BASELINE.json: mAP@0.5 parity, +-0.1 points)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('e2e_map', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'e2e_map.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.main(300)
    assert res['loss_last'] < 0.2 * res['loss_first'], res
    assert res['product']['mAP50'] > 0.8, res
    assert res['abs_diff_mAP50'] <= 1e-3 and res['abs_diff_mAP50_95'] <= 1e-3, res
    assert abs(res['product']['P'] - res['oracle']['P']) <= 1e-3 and abs(res['product']['R'] - res['oracle']['R']) <= 1e-3, res
    assert res['confusion_diagonal_share_of_labels'] > 0.8, res     # val.py:186's confusion matrix, filled on the device by V.run


def test_bn_statistics_survive_a_large_channel_mean():
    """Channels whose mean dwarfs their spread (mean 300, std 0.05): sum_sq/n - mean^2 in fp32 would be pure rounding noise; the
    kernels take the sums around the running mean (the pivot), so variance, the normalised output and the backward pass stay
    accurate once the running statistics have caught up with the data."""
    from somi_amd import ops
    g = torch.Generator().manual_seed(21)
    B, H, W, C = 4, 24, 20, 16
    x = 300.0 + 0.05 * torch.randn(B, C, H, W, generator=g, dtype=torch.float64)
    x32 = x.float()
    bn = nn.BatchNorm2d(C, eps=1e-3, momentum=0.03).double()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g, dtype=torch.float64) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g, dtype=torch.float64))
        bn.running_mean.fill_(299.9)                             # the running mean has (nearly) caught up
    xr = x32.double().requires_grad_(True)
    z = F.silu(bn(xr))
    dz = torch.randn(z.shape, generator=g, dtype=torch.float64)
    z.backward(dz)
    d = torch.device('cuda')
    xd = nhwc(x32).to(d)
    rm, rv = torch.full((C,), 299.9, device=d), torch.ones(C, device=d)
    gam, bet = bn.weight.detach().float().to(d), bn.bias.detach().float().to(d)
    mean, rstd, scale, shift = ops.bn_stats(xd, C, 0, gam, bet, 1e-3, 0.03, rm, rv)
    want_var = x32.double().var((0, 2, 3), unbiased=False)
    rel_close(1.0 / rstd.double().cpu() ** 2 - 1e-3, want_var, rel=1e-3, what='batch variance')
    rel_close(rm, bn.running_mean, rel=1e-6, what='running mean')
    out = ops.chan_affine_act(xd, C, 0, scale, shift, 'silu', 0, torch.empty_like(xd))
    rel_close(out, nhwc(z), rel=2e-2, what='normalised output (x - mean is only ~8 bits of a float at mean 300)')
    dx, dg, db = torch.empty_like(xd), torch.zeros(C, device=d), torch.zeros(C, device=d)
    ops.bn_act_backward(nhwc(dz).float().to(d), 0, xd, 0, C, mean, rstd, scale, shift, 'silu', 0, True, dx, 0, dg, db)
    rel_close(dg, bn.weight.grad, rel=2e-2, what='dgamma')
    rel_close(db, bn.bias.grad, rel=1e-3, what='dbeta')


def test_rccl_call_pattern_single_rank_rehearsal():
    """SOMI_DDP_SINGLE_RANK=1: bench.py's N > 1 path (RCCL init bound to the device, weight broadcast, bucketed asynchronous
    all-reduces on the side stream, barriers, MAX over ranks, the `allreduce` report) on a one-rank `nccl` group - the call
    pattern the driver's 8-GPU run uses, on the one GPU this box has."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:   # a port nobody listens on (a fixed one can meet a socket in TIME_WAIT)
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, SOMI_DDP_SINGLE_RANK='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '2', '--warmup', '1', '--batch', '4', '--size', '256',
                        '--model', 'somi', '--no-cpu-baseline', '--no-infer'], capture_output=True, text=True, timeout=420, env=env, cwd=root)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 1 and line['value'] > 0
    ar = line['allreduce']
    assert ar['backend'] == 'nccl' and ar['ranks'] == 1 and ar['bytes'] == 311747792 and ar['buckets'] >= 7 and ar['ms'] > 0


def test_bench_two_ranks_end_to_end_on_one_gpu():
    """bench.py's N > 1 path end to end exactly as the driver launches it (python -m torch.distributed.run --nproc-per-node 2 ...
    bench.py --gpus 2), with gloo standing in for RCCL because both ranks share this box's one GPU: the printed line must say
    n_gpus 2, dp2, carry the DCNv3 roofline object and the all-reduce report of a 2-rank group."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SOMI_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', '29571', os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '2',
                        '--size', '256', '--no-infer'], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['config']['parallelism'] == 'dp2' and line['scaling'] == 'weak' and line['value'] > 0
    assert line['allreduce']['ranks'] == 2 and line['allreduce']['backend'] == 'gloo' and 'cpu_baseline' not in line
    assert line['roofline_dcnv3']['bound'] == 'hbm' and set(line['roofline_dcnv3']['kernels']) == {'dcnv3_fwd_kernel', 'dcnv3_bwd_kernel'}


def _optimizer_snapshot(opt):
    st = [{k: (v.data.clone() if hasattr(v, 'data') and not torch.is_tensor(v) else (v.clone() if torch.is_tensor(v) else v))
           for k, v in d.items()} for d in opt._flat]
    return dict(flat=st, bufs=opt._bflat.data.clone(), buf_ema=None if opt._buf_ema is None else opt._buf_ema.clone(), steps=opt.steps,
                updates=opt.updates)


def _optimizer_restore(opt, snap, model):
    with torch.no_grad():
        for d, s in zip(opt._flat, snap['flat']):
            d['p'].data.copy_(s['p'])
            d['g'].data.copy_(s['g'])
            d['m'].copy_(s['m'])
            d['v'].copy_(s['v'])
            if d['ema'] is not None:
                d['ema'].copy_(s['ema'])
        opt._bflat.data.copy_(snap['bufs'])
        if opt._buf_ema is not None:
            opt._buf_ema.copy_(snap['buf_ema'])
    opt.steps, opt.updates = snap['steps'], snap['updates']
    for m in model.modules():                                    # BatchNorm step counters live outside the flat buffers
        if isinstance(m, nn.BatchNorm2d):
            m.num_batches_tracked.zero_()
    model.invalidate()


@pytest.mark.parametrize('graph', ['somi_w025', 'somi_full', 'yolov5s', 'somi_dcn_w025'])
def test_training_step_is_bit_reproducible(graph):
    """The same TrainStep from the same state, twice: gradients (captured right before the optimizer), BatchNorm statistics, weights,
    Adam moments and the EMA shadow must be bit-identical - every reduction on the path has a fixed order (stream-K fix-up, wgrad
    splits, BN partial sums, the loss's per-cell entry lists, attention / ODConv parameter gradients, SPPF routing).  A difference
    names the first tensor that moved, i.e. the kernel to look at; run-to-run noise can then no longer hide a race."""
    from somi_amd.configs import HYP_VISDRONE, SOMI_ANCHORS, fill_state, somi_cfg, synthetic_batch, yolov5_cfg
    from somi_amd.model import Model
    from somi_amd.train import TrainStep
    cfg, B, S, nc = {'somi_w025': (somi_cfg(0.25, 0.33, anchors=SOMI_ANCHORS), 4, 128, 10),
                     'somi_full': (somi_cfg(1.0, 1.0, anchors=SOMI_ANCHORS), 4, 160, 10),
                     'yolov5s': (yolov5_cfg(), 4, 256, 80),
                     'somi_dcn_w025': (somi_cfg(0.25, 0.33, anchors=SOMI_ANCHORS, dcn=True), 4, 128, 10)}[graph]
    model = fill_state(Model(cfg), 5)
    if graph == 'somi_dcn_w025':
        # fill_state's offset branch already throws taps beyond the 2-pixel window slack; scale it up further (x3: offsets of several
        # pixels, more than the 300-step soak run reached) - the near pass of the DCNv3 backward adds those in a fixed order, so the step
        # stays bit-reproducible.  The assertion on the far-tap counter below proves that no float atomic ran.
        with torch.no_grad():
            for m_ in model.modules():
                if type(m_).__name__ == 'DCNv3':
                    m_.offset.weight.mul_(3.0), m_.offset.bias.mul_(3.0)
    model = model.cuda()
    from somi_amd import ops as _ops
    _ops.reset_dcn_overflow_taps()                               # the far-tap total counts from here (earlier tests throw taps +-20 px on purpose)
    tr = TrainStep(model, dict(HYP_VISDRONE), B)
    imgs, targets = synthetic_batch(B, S, nc=nc, seed=8)
    targets = torch.cat([targets, targets[:7]])                  # duplicate targets: several entries per loss cell
    imgs, targets = imgs.cuda(), targets.cuda()
    tr.step(imgs, targets)                                       # one step first, so the moments / EMA / statistics are non-trivial
    torch.cuda.synchronize()
    snap = _optimizer_snapshot(tr.optimizer)
    names = {}
    for st in tr.optimizer._flat:
        for p, o in zip(st['p'].tensors, st['p'].offsets):
            names[(id(st), o)] = next(n for n, q in model.named_parameters() if q is p)
    runs = []
    for _ in range(2):
        _optimizer_restore(tr.optimizer, snap, model)
        grads = []
        real = tr.optimizer.step

        def spy(real=real, grads=grads):
            grads.extend(g.clone() for g in tr.optimizer.flat_grads)
            real()
        tr.optimizer.step = spy
        loss, items = tr.step(imgs, targets)
        tr.optimizer.step = real
        torch.cuda.synchronize()
        runs.append(dict(loss=loss.clone(), items=items.clone(), grads=grads, params=[p.clone() for p in tr.optimizer.flat_params],
                         bufs=tr.optimizer.flat_buffers.clone(), m=[st['m'].clone() for st in tr.optimizer._flat],
                         v=[st['v'].clone() for st in tr.optimizer._flat], ema=[st['ema'].clone() for st in tr.optimizer._flat]))
    a, b = runs
    if graph == 'somi_dcn_w025':
        from somi_amd import ops
        # summed over EVERY DCNv3 backward of both runs (two sites per step share one workspace whose counter word each call zeroes again)
        assert ops.dcn_overflow_taps(total=True) == 0, 'sampling taps went farther than the near pass reaches: this run exercises the atomic fallback, not the claim'

    def first_difference(x, y, st):
        idx = int((x != y).nonzero()[0])
        offs = sorted(o for (sid, o) in names if sid == id(st))
        owner = max(o for o in offs if o <= idx)
        return f'{names[(id(st), owner)]} (+{idx - owner}): {float(x[idx])!r} vs {float(y[idx])!r}, {int((x != y).sum())} elements differ'
    assert torch.equal(a['loss'], b['loss']) and torch.equal(a['items'], b['items']), 'loss differs between two identical runs'
    for gi, st in enumerate(tr.optimizer._flat):
        assert torch.equal(a['grads'][gi], b['grads'][gi]), 'gradient of ' + first_difference(a['grads'][gi], b['grads'][gi], st)
    assert torch.equal(a['bufs'], b['bufs']), 'BatchNorm running statistics differ'
    for key in ('params', 'm', 'v', 'ema'):
        for gi, st in enumerate(tr.optimizer._flat):
            assert torch.equal(a[key][gi], b[key][gi]), f'{key} of ' + first_difference(a[key][gi], b[key][gi], st)


# layers of the 1280x1280 DCN graph that are re-run ALONE on the tensors captured in the whole-graph fp64 pass: the
# blocks whose parameters stood out in round 3's full-size run - 30 and 36 (C2fCBAM, n = 3, 160^2 / 40^2), 27 (C2fCBAM n = 3 at 320^2), the SEAM
# blocks 22 / 26 (160^2 / 320^2), the DCNv3 sites 11 / 13 (320^2 / 160^2) - and the block types that had no full-shape isolated test: ODConv
# 64 -> 128 stride 2 at 640 -> 320 (layer 1) and the decoupled head 256 -> 177 -> 98 at 320^2 ... 40^2 (layer 37)
ISOLATED_1280 = (1, 11, 13, 22, 26, 27, 30, 36, 37)


@pytest.fixture(scope='module')
def uavdt1280():
    """BASELINE configs[3] per-GPU shape - full-width SOMI WITH its DCNv3 sites, nc 3, 1280x1280 (grids 320 / 160 / 80 / 40), batch 2 - one
    training step on the HIP path, on the fp64 CPU oracle (measuring Q = sqrt(sum terms^2) of every parameter gradient and capturing the inputs /
    output gradients of the ISOLATED_1280 layers) and on the fp32 CPU oracle.  Shared by the tests below; ~140 s of CPU work."""
    import copy
    from isolate import capture
    from oracle.somi_ref import Model as OModel
    from oracle.somi_ref.loss import ComputeLoss as OLoss
    from oracle.somi_ref.testing import SOMI_ANCHORS, AbsTermSums, fill_state, somi_cfg, synthetic_batch, HYP_VISDRONE
    from somi_amd.loss import ComputeLoss
    from somi_amd.model import Model
    torch.set_num_threads(16)
    cfg = somi_cfg(1.0, 1.0, nc=3, anchors=SOMI_ANCHORS, dcn=True)
    ref = fill_state(OModel(cfg), 6)
    mine = Model(cfg)
    mine.load_state_dict(ref.state_dict())
    ref.hyp = mine.hyp = dict(HYP_VISDRONE)
    ref64 = copy.deepcopy(ref).double()
    imgs, targets = synthetic_batch(2, 1280, nc=3, seed=14)
    mine = mine.cuda().train()
    pm = mine(imgs.cuda())
    lm, _ = ComputeLoss(mine)(pm, targets.cuda())
    lm.backward()
    torch.cuda.synchronize()
    st = dict(mine=mine, ref=ref, ref64=ref64, lm=lm.detach().cpu(), pm=[p.detach().cpu() for p in pm],
              hip=[(n, p.grad.detach().cpu()) for n, p in mine.named_parameters() if p.grad is not None])
    ref64.train(), ref.train()

    def run64():
        with AbsTermSums(ref64, sums=False) as cond:
            p64 = ref64(imgs.double() / 255)
            l64, _ = OLoss(ref64)(p64, targets.double())
            l64.backward()
        st.update(rss=cond.rss, l64=l64.detach(), p64=[p.detach() for p in p64])
    st['cap'] = capture(ref64, run64, layers=set(ISOLATED_1280))
    st['g64'] = {n: p.grad.detach().clone() for n, p in ref64.named_parameters() if p.grad is not None}
    p32 = ref(imgs.float() / 255)
    l32, _ = OLoss(ref)(p32, targets)
    l32.backward()
    st.update(p32=[p.detach() for p in p32], cpu=[(n, p.grad.detach().clone()) for n, p in ref.named_parameters() if p.grad is not None])
    yield st
    st.clear()
    torch.cuda.empty_cache()


def test_uavdt_1280_nc3_training_step_gradients(uavdt1280):
    """The whole-graph step at configs[3]'s per-GPU shape against the fp64 oracle: loss at 1e-4, train-mode outputs within 4x of the fp32 CPU
    oracle's own distance from fp64, and the POPULATION of parameter gradients.  With fill_state's weights this graph is chaotic (the fp32 CPU
    oracle itself is a median 1.3e-2, q90 2.2e-2 away from its own fp64 run in units of |g| + Q, Q = the root-sum-square of the gradient's terms
    measured in the fp64 pass; one fp32 rounding of the weights moves single parameters of the CPU ORACLE by up to tens of those q90 -
    profiles/r04_twin_distribution_1280.txt), so no per-parameter bar below O(1) means anything HERE; per-parameter guarantees come from the two
    tests that can give them: every block of this very graph re-run alone on the tensors captured in this pass
    (test_uavdt_1280_block_isolated_on_captured_tensors) and the well-conditioned full-width step at a flat 1e-3
    (test_full_width_well_conditioned_step_holds_flat_1e3).  The bar uses nothing measured on the HIP path: median and 90th
    percentile of r(HIP) within a FIXED 4x of the fp32 CPU oracle's, and no O(1) error anywhere (r <= 0.5).  What the 2.6x between the two
    fp32 paths is: on identical inputs a HIP conv layer is 1.2e-6 ... 1.5e-6 from fp64 where MKLDNN is 3e-7 (tools/block_isolate.py: the MFMA sums
    K = 9 Cin terms as ONE fp32 fmaf chain, MKLDNN in blocked partial sums), and the graph amplifies either by the same ~10^5."""
    from oracle.somi_ref.testing import noise_scaled_errors
    st = uavdt1280
    assert [tuple(p.shape) for p in st['pm']] == [(2, 4, g, g, 8) for g in (320, 160, 80, 40)]
    rel_close(st['lm'], st['l64'].float(), rel=1e-4, what='loss @1280 nc=3')
    for a, b32, b64 in zip(st['pm'], st['p32'], st['p64']):
        scale = b64.abs().max().item()
        e_mine = (a.double() - b64).abs().max().item() / scale
        e_o32 = (b32.double() - b64).abs().max().item() / scale
        assert e_mine <= max(4 * e_o32, 1e-3), f'train outputs @1280: HIP {e_mine:.2e} vs fp32 CPU {e_o32:.2e} (relative to fp64)'
    g64, rss = st['g64'], st['rss']
    assert {n for n, _ in st['hip']} == set(g64), 'gradient presence differs from the oracle'
    hip = noise_scaled_errors(st['hip'], g64, rss)
    cpu = noise_scaled_errors(st['cpu'], g64, rss)
    assert len(hip) == len(cpu) == len(g64), f'{len(g64) - len(hip)} parameters without a measured term scale'
    rh, rc = torch.tensor([t[1] for t in hip]), torch.tensor([t[1] for t in cpu])
    print(f'1280 nc=3 with DCNv3 sites: noise-scaled error r  HIP median {float(rh.median()):.2e} q90 {float(rh.quantile(0.9)):.2e} max {float(rh.max()):.2e} | '
          f'fp32 CPU median {float(rc.median()):.2e} q90 {float(rc.quantile(0.9)):.2e} max {float(rc.max()):.2e}')
    gross = [(n, f'r {r:.2e}') for n, r, _ in hip if r > 0.5]
    assert not gross, f'O(1) gradient errors: {gross[:8]}'
    assert rh.median() <= 4.0 * float(rc.median()), (float(rh.median()), float(rc.median()))
    assert rh.quantile(0.9) <= 4.0 * float(rc.quantile(0.9)), (float(rh.quantile(0.9)), float(rc.quantile(0.9)))
    # every parameter within a FIXED 12 x max(u90, the CPU oracle's own distance there).  Where the 12 comes from (profiles/r04_twin_distribution_1280.txt,
    # 16 one-rounding twins of the fp32 CPU ORACLE, nothing of the HIP path in it): one fp32 rounding of the weights moves single parameters of the
    # oracle itself by up to 10.05 u90 (11 of 16 twins move some parameter by >= 7.1 - the same 1.59e-1 each time: a pooling / ReLU switch upstream
    # of model.30.m.2.cv1 that fp32 evaluations take either way), none of the 16 x 734 responses is beyond 12, and the HIP path's largest
    # per-parameter distance is 7.7 u90 - 1.5 x that parameter's own largest CPU-twin response
    u90 = float(rc.quantile(0.9))
    cpu_r = {t[0]: t[1] for t in cpu}
    far = [(n, f'r {r:.2e}', f'cpu {cpu_r[n]:.2e}', f'x{r / max(u90, cpu_r[n]):.1f}') for n, r, _ in hip if r > 12.0 * max(u90, cpu_r[n])]
    assert not far, f'{len(far)} parameters beyond 12 x max(u90 = {u90:.2e}, the fp32 CPU oracle\'s own distance): {far[:8]}'


@pytest.mark.parametrize('layer', ISOLATED_1280)
def test_uavdt_1280_block_isolated_on_captured_tensors(uavdt1280, layer):
    """Capture-and-isolate: layer `layer` of the 1280x1280 graph alone, on the inputs and output gradients the
    whole-graph fp64 pass produced for it, rounded to fp32 - the fp64 oracle block and the HIP block see bit-identical tensors and weights, so
    nothing arrives from upstream.  Output and input gradient at 1e-3 (1e-5 observed); EVERY parameter gradient at BASELINE's flat 1e-3 of its
    scale plus a fixed 16 fp32 roundings of its own terms.  Only the DCNv3 blocks also run the fp32 CPU oracle block: the offset branch's
    gradient is discontinuous where a sampling point crosses a pixel boundary, which puts ANY fp32 evaluation 2e-3 ... 7e-3 from fp64 on the
    offset / depthwise parameters (HIP and MKLDNN agree there to three digits); those get max(1e-3, 2 x the CPU oracle's own distance)."""
    from isolate import hip_alone, oracle_alone
    st = uavdt1280
    rec = st['cap'][layer]
    assert all(d is not None for d in rec['dy']), 'no output gradient captured'
    blk64, blk32, mine = st['ref64'].model[layer], st['ref'].model[layer], st['mine'].model[layer]
    table = {}
    oh, dxh, gh = hip_alone(mine, rec, decisions=table)
    # the oracle differentiates the block AT the arg-max decisions the HIP forward took (ForcedDecisions: a max's gradient jumps where two
    # candidates tie to fp32 rounding; with ~10^6 decisions per block such a near-tie is met now and then, by any fp32 implementation)
    o64, dx64, g64, sums = oracle_alone(blk64, rec, torch.float64, with_sums=True, forced=table)
    is_dcn = blk64.type == 'DCNv3_YOLO'
    g32 = list(oracle_alone(blk32, rec, torch.float32, forced=table)[2].items()) if is_dcn else None
    for got, want in zip(oh, o64):
        rel_close(got, want, what=f'layer {layer} ({blk64.type}) output')
    if layer != 0:
        for got, want in zip(dxh, dx64):
            # the DCNv3 input gradient inherits the offset branch's discontinuity (both fp32 paths: 8e-3 ... 2e-2 of the largest element)
            rel_close(got, want, rel=5e-2 if is_dcn else 1e-3, what=f'layer {layer} ({blk64.type}) input gradient')
    _anchored_check(gh.items(), g32, g64, sums, f'layer {layer} ({blk64.type}) alone on captured tensors', k_cpu=2.0 if is_dcn else 0.0)


@pytest.mark.parametrize('amp,loss_rel,grad_med,grad_q90,cos_min', [('bf16x3', 2e-5, 2e-3, 1e-2, 0.99999), ('bf16', 1e-2, None, None, 0.8)])
def test_amp_training_step_stays_in_its_band(amp, loss_rel, grad_med, grad_q90, cos_min):
    """Opt-in reduced precision (TrainStep(amp=...), train.py:263 autocast): the whole small SOMI graph, one forward + loss + backward with the
    conv family's products on the bf16 matrix instructions, against the exact fp32 path on the same weights and batch.  The bands are
    stated here.  bf16x3 (two bf16 values per operand, three products): loss to 2e-5, median parameter-gradient error 2e-3 of the
    gradient's scale (measured 4.6e-4: the kernels are 1e-5-class, the 38-layer graph with random weights amplifies it), cosine of the
    whole gradient vector >= 0.99999.  Plain bf16 (autocast's arithmetic, 8 mantissa bits): loss to 1e-2 (measured 1.8e-4); per-parameter
    gradients of this randomly initialised deep graph are NOT close element-wise under it (measured median 0.33 of the scale - what
    autocast does to such a graph on any hardware), so the band is on the direction of the whole gradient (cosine >= 0.8).  Everything
    that is not a conv product stays fp32."""
    from somi_amd import ops
    from somi_amd.configs import HYP_VISDRONE, SOMI_ANCHORS, fill_state, somi_cfg, synthetic_batch
    from somi_amd.loss import ComputeLoss
    from somi_amd.model import Model
    cfg = somi_cfg(0.5, 0.33, anchors=SOMI_ANCHORS, dcn=True)     # width 0.5: most layers have Cin % 32 == 0 and take the bf16 kernels
    model = fill_state(Model(cfg), 4).cuda().train()
    model.hyp = dict(HYP_VISDRONE)
    imgs, targets = synthetic_batch(4, 128, seed=9)
    imgs, targets = imgs.cuda(), targets.cuda()

    def run(prec):
        for p in model.parameters():
            p.grad = None
        ops.CONV_PREC = ops.PREC[prec]
        try:
            loss, _ = ComputeLoss(model)(model(imgs), targets)
            loss.backward()
        finally:
            ops.CONV_PREC = 0
        torch.cuda.synchronize()
        return loss.detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    bufs = {n: b.clone() for n, b in model.named_buffers()}
    l0, g0 = run(None)
    with torch.no_grad():                                        # same BatchNorm running statistics for the second pass
        for n, b in model.named_buffers():
            b.copy_(bufs[n])
    l1, g1 = run(amp)
    assert abs(l1.item() - l0.item()) <= loss_rel * abs(l0.item()), f'{amp}: loss {l1.item()} vs fp32 {l0.item()}'
    assert l1.item() != l0.item(), f'{amp}: loss bit-identical to fp32 - the reduced-precision kernels did not run'
    rel = []
    for n, g in g0.items():
        scale = g.abs().max().item()
        if scale > 1e-6:
            rel.append((g1[n] - g).abs().max().item() / scale)
    rel = torch.tensor(rel)
    print(f'{amp}: loss rel {abs(l1.item() - l0.item()) / abs(l0.item()):.2e}; gradient error median {float(rel.median()):.2e} '
          f'q90 {float(rel.quantile(0.9)):.2e} max {float(rel.max()):.2e}')
    v0, v1 = torch.cat([g.flatten().double() for g in g0.values()]), torch.cat([g1[n].flatten().double() for n in g0])
    cos = float((v0 * v1).sum() / (v0.norm() * v1.norm()))
    print(f'{amp}: cosine of the whole gradient vector vs fp32: {cos:.8f}')
    assert cos >= cos_min, cos
    if grad_med is not None:
        assert rel.median() <= grad_med and rel.quantile(0.9) <= grad_q90, (float(rel.median()), float(rel.quantile(0.9)))


def test_multi_scale_step_matches_oracle():
    """`--multi-scale` (train.py:257-262): the batch is resized with F.interpolate(bilinear, align_corners=False) to a size on the stride
    grid before the forward pass.  One TrainStep at an explicitly given size (an upscale and a downscale, non-square) against the CPU
    oracle fed the same interpolation: loss and parameter gradients; and the drawn sizes follow the reference's formula."""
    import random
    from oracle.somi_ref import Model as OModel
    from oracle.somi_ref.loss import ComputeLoss as OLoss
    from oracle.somi_ref.testing import SOMI_ANCHORS, fill_state, somi_cfg, synthetic_batch, HYP_VISDRONE
    from somi_amd.model import Model
    from somi_amd.train import TrainStep, multi_scale_size
    rng = random.Random(3)
    for _ in range(20):
        ns = multi_scale_size((96, 128), 128, 32, rng)
        assert ns is None or (ns[0] % 32 == 0 and ns[1] % 32 == 0 and 64 <= max(ns) <= 192 + 32)
    cfg = somi_cfg(0.25, 0.33, anchors=SOMI_ANCHORS)
    for size in ((160, 128), (64, 96)):
        ref = fill_state(OModel(cfg), 7)
        mine = Model(cfg)
        mine.load_state_dict(ref.state_dict())
        ref.hyp = dict(HYP_VISDRONE)
        imgs, targets = synthetic_batch(2, 128, seed=12)
        imgs = imgs[:, :, :96]                                       # a non-square batch (96 x 128)
        ref.train()
        x = F.interpolate(imgs.float() / 255, size=size, mode='bilinear', align_corners=False)
        lr, _ = OLoss(ref)(ref(x), targets)
        lr.backward()
        tr = TrainStep(mine.cuda(), dict(HYP_VISDRONE), 2)
        grads = []
        real = tr.optimizer.step

        def spy():
            grads.extend(g.clone() for g in tr.optimizer.flat_grads)
            real()
        tr.optimizer.step = spy
        lm, _ = tr.step(imgs.contiguous().cuda(), targets.cuda(), size=size)
        rel_close(lm, lr, rel=1e-4, what=f'multi-scale loss at {size}')
        # gradients live in the flat buffers at optimizer.step time: compare a few parameters through their views
        tr.optimizer.step = real
        got = {n: None for n, _ in mine.named_parameters()}
        for st, flat in zip(tr.optimizer._flat, grads):
            for p, o in zip(st['p'].tensors, st['p'].offsets):
                name = next(n for n, q in mine.named_parameters() if q is p)
                got[name] = (flat, o)
        checked = 0
        for n, q in ref.named_parameters():
            if q.grad is None or q.dim() != 1 or got.get(n) is None:     # 1-D parameters (BN weights, biases) sit unpacked in the flat buffers
                continue
            flat, o = got[n]
            g = flat[o:o + q.numel()].cpu()
            scale = q.grad.abs().max().item()
            assert (g - q.grad).abs().max().item() <= 5e-3 * scale + 5e-6, f"multi-scale d{n} at {size}"     # (the attention-MLP biases: ReLU gates)
            checked += 1
        assert checked > 50
