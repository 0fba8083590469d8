"""Stock YOLOv5 module set on the MI355X (north_star "CSP/Darknet backbone, PANet/FPN neck, anchor-based detection head";
BASELINE configs[0]): Bottleneck, C3, SPP, Focus, Concat, Detect and whole yolov5 graphs through libsomi_hip.so against
(a) vectors produced by the reference's own classes (tests/golden/block_*.npz, model_yolov5_*.npz) and (b) torch autograd on the
CPU oracle.  Bar 1e-3 relative (BASELINE)."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def rel_close(got, want, rel=1e-3, what='', atol=0.0):
    got, want = got.detach().cpu().double(), torch.as_tensor(want).detach().cpu().double()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = (got - want).abs().max().item()
    scale = want.abs().max().item() + 1e-12
    assert err <= rel * scale + atol, f'{what}: max err {err:.3e} vs scale {scale:.3e}'


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def _pair(tag):
    from oracle.somi_ref import blocks as OB
    from somi_amd import blocks as MB
    mk = {'bottleneck_sc': lambda M: M.Bottleneck(16, 16, True, 1, k=((1, 1), (3, 3)), e=1.0),
          'bottleneck_nosc': lambda M: M.Bottleneck(16, 24, True), 'c3_sc': lambda M: M.C3(32, 32, 2, True),
          'c3_nosc': lambda M: M.C3(48, 32, 1, False), 'spp': lambda M: M.SPP(32, 32, (5, 9, 13)), 'focus': lambda M: M.Focus(3, 16, 3),
          'conv6x6_s2': lambda M: M.Conv(3, 16, 6, 2, 2)}[tag]
    return mk(OB), mk(MB)


def _bn_hyper(mod):
    for m in mod.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.eps, m.momentum = 1e-3, 0.03
    return mod


BLOCKS = ['bottleneck_sc', 'bottleneck_nosc', 'c3_sc', 'c3_nosc', 'spp', 'focus', 'conv6x6_s2']


@pytest.mark.parametrize('tag', BLOCKS)
def test_stock_blocks_match_reference_vectors(golden, tag):
    """eval (BN folded) and train-mode (batch statistics) forward against the outputs of the reference's own classes."""
    from oracle.somi_ref.testing import fill_state
    from somi_amd import blocks as MB
    g = golden('block_' + tag)
    ref, mine = _pair(tag)
    fill_state(ref, 0)
    mine.load_state_dict(ref.state_dict())
    mine = _bn_hyper(mine).cuda()
    x = MB.Act(nhwc(T(g['in0'])).cuda())
    if x.t.shape[3] % 4:                                             # the 3-channel image: padded to 4 like the ingest kernel does
        x = MB.Act(torch.nn.functional.pad(x.t, (0, 4 - x.t.shape[3] % 4)).contiguous(), 0, x.t.shape[3])
    for mode in ('eval', 'train'):
        mine.train(mode == 'train')
        with torch.no_grad():
            out = mine(x)
        want = nhwc(T(g[f'out_{mode}']))
        rel_close(out.t[..., out.coff:out.coff + out.c], want, what=f'{tag} {mode}')


@pytest.mark.parametrize('tag,cin,shape', [('bottleneck_sc', 16, (3, 16, 9, 11)), ('bottleneck_nosc', 16, (2, 16, 8, 8)),
                                           ('c3_sc', 32, (2, 32, 10, 10)), ('c3_nosc', 48, (2, 48, 7, 9)), ('spp', 32, (2, 32, 11, 11))])
def test_stock_blocks_train_forward_backward(tag, cin, shape):
    """Training forward + hand-written backward against torch autograd on the oracle block: output, dx, every parameter gradient."""
    from oracle.somi_ref import blocks as OB
    from oracle.somi_ref.testing import fill_state
    from somi_amd import blocks as MB
    ref, mine = _pair(tag)
    fill_state(ref, 5)
    OB.initialize_weights(ref)
    mine.load_state_dict(ref.state_dict())
    mine = _bn_hyper(mine).cuda().train()
    ref.train()
    g = torch.Generator().manual_seed(len(tag))
    x = torch.randn(*shape, generator=g, requires_grad=True)
    y = ref(x)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    out = mine(MB.Act(nhwc(x.detach()).cuda()))
    rel_close(out.t[..., out.coff:out.coff + out.c], nhwc(y), what=f'{tag} forward')
    dx = mine.backward(MB.Act(nhwc(dy).cuda()))
    rel_close(dx.t[..., dx.coff:dx.coff + cin], nhwc(x.grad), what=f'{tag} dx')
    for (n, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        if q.grad is not None:
            assert p.grad is not None, n
            rel_close(p.grad, q.grad, what=f'{tag}: d{n}', atol=2e-5)
    for (n, p), (_, q) in zip(mine.named_buffers(), ref.named_buffers()):
        if 'running' in n:
            rel_close(p, q, what=f'{tag}: {n}')


def test_focus_and_6x6_stem_backward_inside_a_graph():
    """Focus and the 6x6 / stride-2 stem placed BEHIND another layer, so their data gradients (inverse space-to-depth; the strided
    dgrad of a 6x6 kernel) are exercised too - as a first layer they need none."""
    from oracle.somi_ref import blocks as OB
    from oracle.somi_ref.testing import fill_state
    from somi_amd import blocks as MB
    for name, mk in (('focus', lambda M: nn.Sequential(M.Conv(8, 8, 3, 1), M.Focus(8, 16, 3))),
                     ('stem6', lambda M: nn.Sequential(M.Conv(8, 8, 1, 1), M.Conv(8, 16, 6, 2, 2)))):
        ref, mine = mk(OB), mk(MB)
        fill_state(ref, 6)
        OB.initialize_weights(ref)
        mine.load_state_dict(ref.state_dict())
        mine = _bn_hyper(mine).cuda().train()
        ref.train()
        g = torch.Generator().manual_seed(3)
        x = torch.randn(2, 8, 12, 16, generator=g, requires_grad=True)
        y = ref(x)
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy)
        a = MB.Act(nhwc(x.detach()).cuda())
        for m in mine:
            a = m(a)
        rel_close(a.t[..., :16], nhwc(y), what=f'{name} forward')
        d = MB.Act(nhwc(dy).cuda())
        for m in reversed(list(mine)):
            d = m.backward(d)
        rel_close(d.t[..., :8], nhwc(x.grad), what=f'{name} dx')
        for (n, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
            rel_close(p.grad, q.grad, what=f'{name}: d{n}', atol=2e-5)


def test_concat_with_upsampled_input_forward_backward(golden):
    from somi_amd import blocks as MB
    g = golden('block_concat')
    cat = MB.Concat(1).train()
    a, b = MB.Act(nhwc(T(g['in0'])).cuda()), MB.Act(nhwc(T(g['in1'])).cuda())
    out = cat([a, b])
    assert torch.equal(out.t.cpu(), nhwc(T(g['out_eval'])))                  # a copy: bit-exact against the reference's torch.cat
    d = cat.backward(MB.Act(torch.arange(out.t.numel(), dtype=torch.float32, device='cuda').view_as(out.t)))
    assert (d[0].coff, d[0].c, d[1].coff, d[1].c) == (0, 8, 8, 12) and d[0].t is d[1].t
    # nn.Upsample(None, 2, 'nearest') feeding the concat (the PANet top-down path): the view flag is expanded by the copy,
    # the gradient of the upsampled input is the 2x2 block sum
    gen = torch.Generator().manual_seed(9)
    lo = torch.randn(2, 8, 3, 5, generator=gen, requires_grad=True)
    hi = torch.randn(2, 12, 6, 10, generator=gen, requires_grad=True)
    y = torch.cat([nn.functional.interpolate(lo, scale_factor=2, mode='nearest'), hi], 1)
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy)
    up = MB.Upsample(None, 2, 'nearest')
    out = cat([up(MB.Act(nhwc(lo.detach()).cuda())), MB.Act(nhwc(hi.detach()).cuda())])
    assert torch.equal(out.t.cpu(), nhwc(y.detach()))
    dlo, dhi = cat.backward(MB.Act(nhwc(dy).cuda()))
    rel_close(dlo.t, nhwc(lo.grad), rel=1e-6, what='d upsampled input')
    assert torch.equal(dhi.t[..., dhi.coff:dhi.coff + dhi.c].cpu(), nhwc(hi.grad))


def test_plain_detect_matches_reference_vectors_and_autograd(golden):
    from oracle.somi_ref import blocks as OB
    from oracle.somi_ref.testing import fill_state
    from somi_amd import blocks as MB
    g = golden('block_detect')
    anchors = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119]]
    ref = fill_state(OB.Detect(7, anchors, [16, 24]), 3)
    mine = MB.Detect(7, anchors, [16, 24])
    mine.load_state_dict(ref.state_dict())
    ref.stride = mine.stride = torch.tensor([8., 16.])
    ref.anchors /= ref.stride.view(-1, 1, 1)
    mine.anchors /= mine.stride.view(-1, 1, 1)
    mine = mine.cuda().eval()
    xs = [T(g['in0']), T(g['in1'])]
    with torch.no_grad():
        z, raw = mine([MB.Act(nhwc(x).cuda()) for x in xs])
    rel_close(z, T(g['z']), what='z')
    rel_close(raw[0], T(g['raw0']), what='raw0')
    rel_close(raw[1], T(g['raw1']), what='raw1')
    # training: raw outputs + backward against autograd
    ref.train(), mine.train()
    gen = torch.Generator().manual_seed(4)
    xs = [x.clone().requires_grad_(True) for x in xs]
    ys = ref(list(xs))
    dys = [torch.randn(y.shape, generator=gen) for y in ys]
    torch.autograd.backward(ys, dys)
    outs = mine([MB.Act(nhwc(x.detach()).cuda()) for x in xs])
    for o, y in zip(outs, ys):
        rel_close(o, y, what='detect raw (train)')
    dxs = mine.backward([d.cuda() for d in dys])
    for d, x in zip(dxs, xs):
        rel_close(d.t[..., :x.shape[1]], nhwc(x.grad), what='detect dx')
    for (n, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        rel_close(p.grad, q.grad, what=f'Detect: d{n}', atol=2e-5)


@pytest.mark.parametrize('tag,version', [('yolov5_v6', '6.0'), ('yolov5_v5', '5.0')])
def test_stock_yolov5_model_matches_reference_vectors(golden, tag, version):
    """Whole stock graphs against the numbers of the reference's own Model (width 0.25, 80 classes): eval and fused-eval
    predictions, train-mode outputs, the loss of the reference's ComputeLoss and parameter gradients from its autograd."""
    from oracle.somi_ref.testing import HYP_VISDRONE, fill_state, yolov5_cfg
    from oracle.somi_ref import Model as OModel
    from somi_amd.loss import ComputeLoss
    from somi_amd.model import Model
    g = golden('model_' + tag)
    cfg = yolov5_cfg(0.25, 0.33, nc=80, version=version)
    ref = fill_state(OModel(cfg), 1)
    mine = Model(cfg)
    assert sum(p.numel() for p in mine.parameters()) == int(g['nparams'])
    mine.load_state_dict(ref.state_dict())
    rel_close(mine.stride, T(g['stride']), what='stride')
    rel_close(mine.model[-1].anchors, T(g['anchors']), what='anchors')
    mine = mine.cuda().eval()
    x = T(g['x']).cuda()
    with torch.no_grad():
        z, raw = mine(x)
    rel_close(z, T(g['z']), what='z')
    rel_close(z, T(g['z_fused']), what='z vs Model.fuse() output')
    for i, r in enumerate(raw):
        rel_close(r, T(g[f'raw{i}']), what=f'raw{i}')
    mine.train()
    mine.hyp = dict(HYP_VISDRONE)
    tr = mine(x)
    for i, r in enumerate(tr):
        rel_close(r, T(g[f'train{i}']), what=f'train{i}')
    loss, items = ComputeLoss(mine)(tr, T(g['targets']).cuda())
    rel_close(loss, T(g['loss']).reshape(1), rel=1e-4, what='loss')
    rel_close(items, T(g['loss_items']), rel=1e-4, what='loss items')
    loss.backward()
    pd = dict(mine.named_parameters())
    for i, n in enumerate(g['grad_names']):
        want = T(g[f'grad{i}'])
        rel_close(pd[str(n)].grad, want, rel=2e-3, atol=2e-6, what=f'd{n}')


def test_yolov5s_coco_training_step_and_nms():
    """BASELINE configs[0]: yolov5s (7,235,389 parameters, 80 classes) at 640x640, batch 2 - one whole training step (loss, every
    parameter gradient, BN statistics) against the CPU oracle, then eval predictions and NMS (selection bit-exact)."""
    from oracle.somi_ref import Model as OModel
    from oracle.somi_ref.loss import ComputeLoss as OLoss
    from oracle.somi_ref.nms import non_max_suppression as oracle_nms
    from oracle.somi_ref.testing import HYP_VISDRONE, fill_state, synthetic_batch, yolov5_cfg
    from somi_amd.loss import ComputeLoss
    from somi_amd.model import Model
    from somi_amd.nms import non_max_suppression
    cfg = yolov5_cfg()
    ref = fill_state(OModel(cfg), 3)
    mine = Model(cfg)
    assert sum(p.numel() for p in mine.parameters()) == 7235389
    mine.load_state_dict(ref.state_dict())
    ref.hyp = mine.hyp = dict(HYP_VISDRONE)
    imgs, targets = synthetic_batch(2, 640, nc=80, seed=2)
    ref.train()
    pr = ref(imgs.float() / 255)
    lr, ir = OLoss(ref)(pr, targets)
    lr.backward()
    mine = mine.cuda().train()
    pm = mine(imgs.cuda())
    for a, b in zip(pm, pr):
        rel_close(a, b, what='train outputs')
    lm, im = ComputeLoss(mine)(pm, targets.cuda())
    rel_close(lm, lr, rel=1e-4, what='loss')
    rel_close(im, ir, rel=1e-4, what='loss items')
    lm.backward()
    bad = []
    for (n, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        if q.grad is None:
            continue
        assert p.grad is not None, n
        err = (p.grad.cpu().double() - q.grad.double()).abs().max().item()
        scale = q.grad.double().abs().max().item() + 1e-9
        if err > 2e-3 * scale + 2e-6:
            bad.append((n, err, scale))
    assert not bad, bad[:8]
    for (n, p), (_, q) in zip(mine.named_buffers(), ref.named_buffers()):
        if 'running' in n:
            rel_close(p, q, what=n)
    ref.eval(), mine.eval()
    with torch.no_grad():
        zr, _ = ref(imgs.float() / 255)
        z, _ = mine(imgs.cuda())
    assert z.shape == (2, 25200, 85)
    rel_close(z, zr, what='z')
    det = non_max_suppression(z, 0.001, 0.6, multi_label=True)
    want = oracle_nms(z.cpu(), 0.001, 0.6, multi_label=True)
    for a, b in zip(det, want):
        assert a.shape == b.shape and torch.equal(a.cpu(), b)

