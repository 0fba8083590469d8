"""DCNv3 HIP kernels (through the C ABI and the host mirror of the reference extension) against
(1) the golden vectors produced by the reference's dcnv3_core_pytorch and (2) the CPU oracle at larger sizes.
Bars: the reference's own test uses rtol 1e-2 / atol 1e-3 in fp32 (models/ops_dcnv3/test.py:85,134-148);
BASELINE.json asks 1e-3 relative - that is what is asserted here."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def rel_close(got, want, rel=1e-3, what=''):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = (got - want).abs().max().item()
    scale = want.abs().max().item() + 1e-30
    assert err <= rel * scale, f'{what}: max err {err:.3e} vs scale {scale:.3e}'


GOLD = ['testpy_f32', 'testpy_f64', 'bwd_D1', 'bwd_D16', 'bwd_D30', 'bwd_D32', 'bwd_D64', 'bwd_D71', 's2_p1', 'd2_p2',
        'k5_p2', 'p0']


@pytest.mark.parametrize('tag', GOLD)
def test_dcnv3_against_reference_vectors(golden, tag):
    from somi_amd.dcnv3 import dcnv3_forward, dcnv3_backward
    g = golden('dcnv3_' + tag)
    N, H, W, G, Gc, k, s, p, d = (int(v) for v in g['params'])
    osc = float(g['offset_scale'])
    dev = torch.device('cuda:0')
    x, off, m = (T(g[n]).float().contiguous().to(dev) for n in ('input', 'offset', 'mask'))
    out = dcnv3_forward(x, off, m, k, k, s, s, p, p, d, d, G, Gc, osc, 256)
    rel_close(out, T(g['output']), what='forward')
    go = T(g['grad_output']).float().contiguous().to(dev)
    gi, goff, gm = dcnv3_backward(x, off, m, k, k, s, s, p, p, d, d, G, Gc, osc, go, 256)
    rel_close(gi, T(g['grad_input']), what='grad_input')
    rel_close(goff, T(g['grad_offset']), what='grad_offset')
    rel_close(gm, T(g['grad_mask']), what='grad_mask')


@pytest.mark.parametrize('N,H,W,G,Gc,k,s,p,d', [(2, 8, 8, 2, 1025, 3, 1, 1, 1),      # the D=1025 case of test.py:257
                                                (2, 40, 40, 8, 32, 3, 1, 1, 1),       # SOMI-like C=256 G=8
                                                (3, 21, 17, 4, 24, 3, 2, 1, 1),       # lanes/group = 6 (not a power of 2)
                                                (1, 16, 16, 16, 4, 3, 1, 1, 2)])
def test_dcnv3_against_oracle(N, H, W, G, Gc, k, s, p, d):
    from oracle.somi_ref import dcnv3 as O
    from somi_amd.dcnv3 import DCNv3Function
    g = torch.Generator().manual_seed(N * 100 + Gc)
    Ho, Wo = O.dcnv3_out_size(H, k, s, p, d), O.dcnv3_out_size(W, k, s, p, d)
    K = k * k
    x = torch.randn(N, H, W, G * Gc, generator=g)
    off = (torch.rand(N, Ho, Wo, G * K * 2, generator=g) - 0.5) * 6
    m = torch.softmax(torch.randn(N, Ho, Wo, G, K, generator=g), -1).reshape(N, Ho, Wo, G * K)
    go = torch.randn(N, Ho, Wo, G * Gc, generator=g)
    want = O.dcnv3_core(x, off, m, k, k, s, s, p, p, d, d, G, Gc, 1.3)
    wgi, wgo, wgm = O.dcnv3_backward(x, off, m, k, k, s, s, p, p, d, d, G, Gc, 1.3, go, 256)
    dev = torch.device('cuda:0')
    xd, od, md = (t.to(dev).requires_grad_(True) for t in (x, off, m))
    out = DCNv3Function.apply(xd, od, md, k, k, s, s, p, p, d, d, G, Gc, 1.3, 256)
    rel_close(out, want, what='forward')
    out.backward(go.to(dev))
    rel_close(xd.grad, wgi, what='grad_input')
    rel_close(od.grad, wgo, what='grad_offset')
    rel_close(md.grad, wgm, what='grad_mask')


def test_dcnv3_argument_errors():
    """Error behaviour of the reference host launcher (dcnv3_cuda.cu:29-53) surfaces as RuntimeError."""
    from somi_amd.dcnv3 import dcnv3_forward
    dev = torch.device('cuda:0')
    x = torch.zeros(3, 8, 8, 16, device=dev)
    off = torch.zeros(3, 8, 8, 4 * 9 * 2, device=dev)
    m = torch.zeros(3, 8, 8, 4 * 9, device=dev)
    with pytest.raises(RuntimeError, match='must divide im2col_step'):
        dcnv3_forward(x, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 4, 4, 1.0, 2)
    with pytest.raises(RuntimeError, match='wont match'):
        dcnv3_forward(x, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 4, 8, 1.0, 256)
    with pytest.raises(RuntimeError, match='contiguous'):
        dcnv3_forward(x.transpose(1, 2), off, m, 3, 3, 1, 1, 1, 1, 1, 1, 4, 4, 1.0, 256)
    with pytest.raises(RuntimeError, match='CUDA tensor'):
        dcnv3_forward(x.cpu(), off, m, 3, 3, 1, 1, 1, 1, 1, 1, 4, 4, 1.0, 256)


def test_dcnv3_module_matches_oracle():
    from oracle.somi_ref.dcnv3 import DCNv3 as ODCN
    from oracle.somi_ref.testing import fill_state
    from somi_amd.dcnv3 import DCNv3
    ref = fill_state(ODCN(64, 3, group=4, offset_scale=2.0, center_feature_scale=True), 3).eval()
    mod = DCNv3(64, 3, group=4, offset_scale=2.0, center_feature_scale=True)
    mod.load_state_dict(ref.state_dict())
    mod = mod.cuda().eval()
    x = torch.randn(2, 12, 10, 64, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        rel_close(mod(x.cuda()), ref(x), what='DCNv3 module')


def test_dcnv3_module_helpers():
    import torch.nn.functional as F
    from somi_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 9, 7, 64, generator=g)
    gm, bt = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
    rel_close(ops.layernorm_act(x.cuda(), gm.cuda(), bt.cuda(), 1e-6, 'gelu'), F.gelu(F.layer_norm(x, (64,), gm, bt, 1e-6)),
              rel=1e-5, what='layernorm+gelu')
    m = torch.randn(2, 9, 7, 4 * 9, generator=g)
    rel_close(ops.group_softmax(m.cuda(), 9), torch.softmax(m.view(2, 9, 7, 4, 9), -1).view(2, 9, 7, 36), rel=1e-5, what='softmax')
    a, b, lg = torch.randn(2, 9, 7, 64, generator=g), torch.randn(2, 9, 7, 64, generator=g), torch.randn(2, 9, 7, 4, generator=g)
    s = torch.sigmoid(lg)[..., None].repeat(1, 1, 1, 1, 16).flatten(-2)
    rel_close(ops.cfs_blend(a.cuda(), b.cuda(), lg.cuda(), 4, 16), a * (1 - s) + b * s, rel=1e-5, what='cfs blend')


@pytest.mark.parametrize('N,H,W,G,Gc,k,s,p,d,spread', [(2, 40, 40, 8, 32, 3, 1, 1, 1, 1.0),      # SOMI-like
                                                       (2, 37, 29, 4, 16, 3, 1, 1, 1, 12.0),     # offsets far beyond the kernel footprint / the image
                                                       (3, 21, 17, 4, 24, 3, 2, 1, 1, 3.0),      # stride 2, ragged tiles
                                                       (1, 16, 16, 16, 4, 3, 1, 2, 2, 2.0),      # dilation 2, one lane per group
                                                       (2, 19, 23, 2, 64, 5, 1, 2, 1, 2.5)])     # 5x5
def test_dcnv3_forward_wide_offsets(N, H, W, G, Gc, k, s, p, d, spread):
    """Forward against the CPU oracle at 1e-5 with offsets from well inside one pixel to far outside the image (most taps
    clipped), stride 2 with ragged tiles, dilation 2 with one lane per group, and a 5x5 kernel."""
    from oracle.somi_ref import dcnv3 as O
    from somi_amd.dcnv3 import dcnv3_forward
    g = torch.Generator().manual_seed(H * 31 + W)
    Ho, Wo = O.dcnv3_out_size(H, k, s, p, d), O.dcnv3_out_size(W, k, s, p, d)
    K = k * k
    x = torch.randn(N, H, W, G * Gc, generator=g)
    off = torch.randn(N, Ho, Wo, G * K * 2, generator=g) * spread
    m = torch.softmax(torch.randn(N, Ho, Wo, G, K, generator=g), -1).reshape(N, Ho, Wo, G * K)
    want = O.dcnv3_core(x, off, m, k, k, s, s, p, p, d, d, G, Gc, 1.7)
    dev = torch.device('cuda:0')
    got = dcnv3_forward(x.to(dev), off.to(dev), m.to(dev), k, k, s, s, p, p, d, d, G, Gc, 1.7, 256)
    rel_close(got, want, rel=1e-5, what='forward')


@pytest.mark.parametrize('cfs,C,G', [(True, 64, 4), (False, 64, 4), (False, 256, 8), (True, 512, 8)])   # 256 / 512: the register-resident LN+GELU backward
def test_dcnv3_module_backward_matches_oracle(cfs, C, G):
    """Training through the whole DCNv3 module (modules/dcnv3.py:222-379): input gradient and every parameter gradient from the
    HIP autograd node against CPU autograd of the oracle module."""
    from oracle.somi_ref.dcnv3 import DCNv3 as ODCN
    from oracle.somi_ref.testing import fill_state
    from somi_amd.dcnv3 import DCNv3
    ref = fill_state(ODCN(C, 3, group=G, offset_scale=1.5, center_feature_scale=cfs), 3).train()
    mod = DCNv3(C, 3, group=G, offset_scale=1.5, center_feature_scale=cfs)
    mod.load_state_dict(ref.state_dict())
    mod = mod.cuda().train()
    g = torch.Generator().manual_seed(17)
    x = torch.randn(2, 12, 10, C, generator=g)
    dout = torch.randn(2, 12, 10, C, generator=g)
    xr = x.clone().requires_grad_(True)
    ref(xr).backward(dout)
    xm = x.cuda().requires_grad_(True)
    out = mod(xm)
    rel_close(out, ref(x).detach(), what='module forward (autograd node)')
    out.backward(dout.cuda())
    rel_close(xm.grad, xr.grad, what='d input')
    for (n, p), (_, q) in zip(mod.named_parameters(), ref.named_parameters()):
        assert p.grad is not None, n
        got, want = p.grad.cpu().double(), q.grad.double()
        err, scale = (got - want).abs().max().item(), want.abs().max().item()
        assert err <= 1e-3 * scale + 1e-6, f'd{n}: max err {err:.3e} vs scale {scale:.3e}'


# ---------------------------------------------------------------------------------------------- DCNv3 wired into the graph (J2)
T = torch.from_numpy


@pytest.mark.parametrize('tag', ['plain', 'cfs', 'g8'])
def test_dcnv3_module_matches_reference_module_vectors(golden, tag):
    """The HIP DCNv3 layer against the reference's own `DCNv3_pytorch` module (models/ops_dcnv3/modules/dcnv3.py:95-219): output,
    input gradient, every parameter gradient."""
    from oracle.somi_ref.dcnv3 import DCNv3 as ODCN
    from oracle.somi_ref.testing import fill_state
    from somi_amd.dcnv3 import DCNv3
    g = golden('dcnv3_module_' + tag)
    C, G, cfs = (int(v) for v in g['cfg'])
    kw = dict(kernel_size=3, stride=1, pad=1, group=G, offset_scale=float(g['offset_scale']), center_feature_scale=bool(cfs))
    mod = DCNv3(C, **kw)
    mod.load_state_dict(fill_state(ODCN(C, **kw), 11).state_dict())
    mod = mod.cuda().train()
    x = T(g['x']).cuda().requires_grad_(True)
    y = mod(x)
    rel_close(y, T(g['y']), what='module output')
    y.backward(T(g['dy']).cuda())
    rel_close(x.grad, T(g['dx']), what='d input')
    for n, p in mod.named_parameters():
        want = T(g['grad.' + n]).double()
        err = (p.grad.cpu().double() - want).abs().max().item()
        assert err <= 1e-3 * want.abs().max().item() + 1e-6, f'd{n}: {err:.3e}'


def test_dcnv3_yolo_block_forward_backward(golden):
    """DCNv3 -> BN -> SiLU: eval (BN folded into the output projection) and train outputs against the block built around the
    reference's module, then the hand-written backward against autograd of the oracle block."""
    import torch.nn as nn
    from oracle.somi_ref import blocks as OB
    from oracle.somi_ref.testing import fill_state
    from somi_amd import blocks as MB
    g = golden('block_dcnv3_yolo')
    ref = fill_state(OB.DCNv3_YOLO(32, 3, 1, 4), 2)
    OB.initialize_weights(ref)
    mine = MB.DCNv3_YOLO(32, 3, 1, 4)
    mine.load_state_dict(ref.state_dict())
    mine.bn.eps, mine.bn.momentum = 1e-3, 0.03
    mine = mine.cuda()
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()          # noqa: E731
    x = MB.Act(nhwc(T(g['in0'])).cuda())
    for mode in ('eval', 'train'):
        mine.train(mode == 'train')
        with torch.no_grad():
            out = mine(x)
        rel_close(out.t, nhwc(T(g[f'out_{mode}'])), what=f'DCNv3_YOLO {mode}')
    ref = fill_state(OB.DCNv3_YOLO(32, 3, 1, 4), 2).train()      # fresh running statistics on both sides
    OB.initialize_weights(ref)
    mine.load_state_dict(ref.state_dict())
    mine.train()
    gen = torch.Generator().manual_seed(6)
    xr = torch.randn(3, 32, 9, 13, generator=gen, requires_grad=True)
    y = ref(xr)
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy)
    out = mine(MB.Act(nhwc(xr.detach()).cuda()))
    rel_close(out.t, nhwc(y), what='train forward')
    dx = mine.backward(MB.Act(nhwc(dy).cuda()))
    rel_close(dx.t, nhwc(xr.grad), what='dx')
    for (n, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        err, scale = (p.grad.cpu().double() - q.grad.double()).abs().max().item(), q.grad.abs().max().item()
        # the output projection's bias sits in front of a batch-statistics BatchNorm: its gradient is zero in exact arithmetic,
        # what both sides hold is rounding noise of the size of the cancelling terms
        atol = 1e-4 if n.endswith('output_proj.bias') else 2e-6
        assert err <= 1e-3 * scale + atol, f'd{n}: max err {err:.3e} vs scale {scale:.3e}'
    for n in ('running_mean', 'running_var'):
        rel_close(getattr(mine.bn, n), getattr(ref.bn, n), what=n)
    assert isinstance(mine.bn, nn.BatchNorm2d)


def test_somi_graph_with_dcnv3_sites_matches_reference_vectors(golden):
    """The SOMI graph with its two DCNv3 sites (BASELINE configs[1] "yolov5l-SOMI (DCNv3 blocks)", here at width 0.25) against the
    numbers of the reference's own Model built around the reference's DCNv3 module: eval predictions, train-mode outputs, the
    reference ComputeLoss and the gradients of every DCNv3-site parameter."""
    from oracle.somi_ref import Model as OModel
    from oracle.somi_ref.testing import HYP_VISDRONE, SOMI_ANCHORS, fill_state, somi_cfg
    from somi_amd.loss import ComputeLoss
    from somi_amd.model import Model
    g = golden('model_w025_dcn')
    cfg = somi_cfg(0.25, 0.33, anchors=SOMI_ANCHORS, dcn=True)
    mine = Model(cfg)
    assert sum(p.numel() for p in mine.parameters()) == int(g['nparams'])
    mine.load_state_dict(fill_state(OModel(cfg), 1).state_dict())
    mine = mine.cuda().eval()
    x = T(g['x']).cuda()
    with torch.no_grad():
        z, raw = mine(x)
    rel_close(z, T(g['z']), what='z')
    for i, r in enumerate(raw):
        rel_close(r, T(g[f'raw{i}']), what=f'raw{i}')
    mine.train()
    mine.hyp = dict(HYP_VISDRONE)
    tr = mine(x)
    for i, r in enumerate(tr):
        rel_close(r, T(g[f'train{i}']), what=f'train{i}')
    loss, items = ComputeLoss(mine)(tr, T(g['targets']).cuda())
    rel_close(loss, T(g['loss']).reshape(1), rel=1e-4, what='loss')
    loss.backward()
    pd = dict(mine.named_parameters())
    for i, n in enumerate(g['grad_names']):
        want = T(g[f'grad{i}']).double()
        err = (pd[str(n)].grad.cpu().double() - want).abs().max().item()
        assert err <= 2e-3 * want.abs().max().item() + 2e-6, f'd{n}: {err:.3e} vs scale {want.abs().max().item():.3e}'


@pytest.mark.parametrize('S', [640, 1280])
def test_full_width_dcn_graph_sites_at_80_and_160(S):
    """Full-width graph with DCNv3 at its real shapes (256 channels, 8 groups of 32): sites at 160x160 and 80x80 for a 640 image,
    320x320 and 160x160 at 1280 (BASELINE configs[3]); one image, eval predictions against the CPU oracle."""
    from oracle.somi_ref import Model as OModel
    from oracle.somi_ref.testing import SOMI_ANCHORS, fill_state, somi_cfg
    from somi_amd.model import Model
    cfg = somi_cfg(1.0, 1.0, nc=3 if S == 1280 else 10, anchors=SOMI_ANCHORS, dcn=True)
    ref = fill_state(OModel(cfg), 4).eval()
    mine = Model(cfg)
    mine.load_state_dict(ref.state_dict())
    mine = mine.cuda().eval()
    x = torch.rand(1, 3, S, S, generator=torch.Generator().manual_seed(S))
    with torch.no_grad():
        zr, _ = ref(x)
        z, _ = mine(x.cuda())
    rel_close(z, zr, what=f'z @{S} with DCNv3 sites')


# ---------------------------------------------------------------------------------------------- boundary: dtypes, module name, custom ops
def _golden_case(g, dtype):
    N, H, W, G, Gc, k, s, p, d = (int(v) for v in g['params'])
    t = lambda key: T(g[key]).to(dtype).cuda()                   # noqa: E731
    return (t('input'), t('offset'), t('mask')), (k, k, s, s, p, p, d, d, G, Gc, float(g['offset_scale'])), t('grad_output')


def test_dcnv3_double_matches_the_reference_test_bar(golden):
    """models/ops_dcnv3/test.py:55: the extension in double against dcnv3_core_pytorch with torch.allclose defaults (rtol 1e-5,
    atol 1e-8) - forward, and the same bar for the three gradients (the reference only asks rtol 1e-2 / atol 1e-3 there, :134-148)."""
    from somi_amd.dcnv3 import dcnv3_backward, dcnv3_forward
    g = golden('dcnv3_testpy_f64')
    (x, off, m), cfg, go = _golden_case(g, torch.float64)
    out = dcnv3_forward(x, off, m, *cfg, 256)
    assert out.dtype == torch.float64 and torch.allclose(out.cpu(), T(g['output'])), float((out.cpu() - T(g['output'])).abs().max())
    gi, goff, gm = dcnv3_backward(x, off, m, *cfg, go, 256)
    for got, key in ((gi, 'grad_input'), (goff, 'grad_offset'), (gm, 'grad_mask')):
        assert got.dtype == torch.float64
        assert torch.allclose(got.cpu(), T(g[key]), rtol=1e-5, atol=1e-8), (key, float((got.cpu() - T(g[key])).abs().max()))
    # ragged group width, stride, dilation, 5x5 in double as well.  dcnv3_core_pytorch builds its reference points in float32
    # (functions/dcnv3_func.py:99-112), so its own double output carries ~1e-7 relative noise that these geometries amplify: 1e-4,
    # a hundred times inside the reference's backward bar (rtol 1e-2 / atol 1e-3, test.py:134-148)
    for tag in ('bwd_D30', 's2_p1', 'd2_p2', 'k5_p2'):
        g = golden('dcnv3_' + tag)
        (x, off, m), cfg, go = _golden_case(g, torch.float64)
        assert torch.allclose(dcnv3_forward(x, off, m, *cfg, 256).cpu(), T(g['output']), rtol=1e-4, atol=1e-7), tag
        for got, key in zip(dcnv3_backward(x, off, m, *cfg, go, 256), ('grad_input', 'grad_offset', 'grad_mask')):
            assert torch.allclose(got.cpu(), T(g[key]), rtol=1e-4, atol=1e-7), (tag, key, float((got.cpu() - T(g[key])).abs().max()))


def test_dcnv3_half_fp32_accumulation():
    """The AMP path (train.py:263; AT_DISPATCH_FLOATING_TYPES_AND_HALF): half storage, fp32 arithmetic, fp32 gradient buffers cast
    back to half (dcnv3_cuda.cu:126-133,168-170).  Yardstick: the CPU oracle in fp32 on the same half-rounded inputs; the only
    difference allowed is the final rounding to half (2^-11 relative) plus fp32 summation order."""
    from oracle.somi_ref import dcnv3 as O
    from somi_amd.dcnv3 import DCNv3Function
    gen = torch.Generator().manual_seed(21)
    N, H, W, G, Gc, k = 2, 14, 11, 4, 16, 3
    x = torch.randn(N, H, W, G * Gc, generator=gen).half()
    off = (torch.randn(N, H, W, G * 9 * 2, generator=gen) * 2).half()
    m = torch.softmax(torch.randn(N, H, W, G, 9, generator=gen), -1).reshape(N, H, W, G * 9).half()
    go = torch.randn(N, H, W, G * Gc, generator=gen).half()
    xr, orr, mr = (t.float().requires_grad_(True) for t in (x, off, m))
    want = O.dcnv3_core(xr, orr, mr, k, k, 1, 1, 1, 1, 1, 1, G, Gc, 1.5)
    want.backward(go.float())
    xd, od, md = (t.cuda().requires_grad_(True) for t in (x, off, m))
    out = DCNv3Function.apply(xd, od, md, k, k, 1, 1, 1, 1, 1, 1, G, Gc, 1.5, 256)
    assert out.dtype == torch.float16
    rel_close(out.float(), want.detach(), rel=1e-3, what='half forward')
    out.backward(go.cuda())
    for got, ref, what in ((xd.grad, xr.grad, 'grad_input'), (od.grad, orr.grad, 'grad_offset'), (md.grad, mr.grad, 'grad_mask')):
        assert got.dtype == torch.float16
        rel_close(got.float(), ref, rel=1e-3, what=f'half {what}')


def test_integration_md_section_1_runs_as_written(golden):
    """INTEGRATION.md section 1 - the reference-side ctypes glue a maintainer would paste over `import DCNv3` - is executed verbatim
    (only the library path is filled in) and then driven exactly like functions/dcnv3_func.py:39-43,54-58 drives the extension."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    md = open(os.path.join(root, 'INTEGRATION.md')).read()
    sec = md[md.index('## 1. DCNv3 extension'):md.index('## 2. Model forward')]
    code = re.search(r'```python\n(.*?)```', sec, re.S).group(1)
    code = code.replace("'.../yolo-somi_amd/lib/libsomi_hip.so'", repr(os.path.join(root, 'yolo-somi_amd', 'lib', 'libsomi_hip.so')))
    ns = {}
    exec(compile(code, 'INTEGRATION.md#1', 'exec'), ns)
    ext = ns['DCNv3']
    for tag, dt in (('testpy_f32', torch.float32), ('testpy_f64', torch.float64), ('s2_p1', torch.float32)):
        g = golden('dcnv3_' + tag)
        (x, off, m), cfg, go = _golden_case(g, dt)
        out = ext.dcnv3_forward(x, off, m, *cfg, 256)
        rel_close(out, T(g['output']), what=f'{tag} forward through the INTEGRATION glue')
        gi, goff, gm = ext.dcnv3_backward(x, off, m, *cfg, go, 256)
        rel_close(gi, T(g['grad_input']), what='grad_input')
        rel_close(goff, T(g['grad_offset']), what='grad_offset')
        rel_close(gm, T(g['grad_mask']), what='grad_mask')
    with pytest.raises(RuntimeError, match='must divide im2col_step'):
        ext.dcnv3_forward(torch.zeros(3, 8, 8, 16, device='cuda'), torch.zeros(3, 8, 8, 72, device='cuda'), torch.zeros(3, 8, 8, 36, device='cuda'),
                          3, 3, 1, 1, 1, 1, 1, 1, 4, 4, 1.0, 2)


def test_import_DCNv3_resolves_to_the_drop_in_module_and_custom_ops(golden):
    """`import DCNv3` (functions/dcnv3_func.py:16) with yolo-somi_amd/ on sys.path, and the torch.library registration: the op pair
    under torch.ops.somi with the autograd formula attached, plus somi.nms against the host wrapper."""
    import DCNv3
    import somi_amd.torch_ops  # noqa: F401
    from somi_amd.nms import non_max_suppression
    g = golden('dcnv3_testpy_f32')
    (x, off, m), cfg, go = _golden_case(g, torch.float32)
    rel_close(DCNv3.dcnv3_forward(x, off, m, *cfg, 256), T(g['output']), what='DCNv3.dcnv3_forward')
    for got, key in zip(DCNv3.dcnv3_backward(x, off, m, *cfg, go, 256), ('grad_input', 'grad_offset', 'grad_mask')):
        rel_close(got, T(g[key]), what=f'DCNv3.dcnv3_backward {key}')
    xs = [t.clone().requires_grad_(True) for t in (x, off, m)]
    out = torch.ops.somi.dcnv3_forward(*xs, *cfg, 256)
    rel_close(out, T(g['output']), what='torch.ops.somi.dcnv3_forward')
    out.backward(go)
    for t, key in zip(xs, ('grad_input', 'grad_offset', 'grad_mask')):
        rel_close(t.grad, T(g[key]), what=f'autograd of the custom op: {key}')
    pred = T(golden('nms')['pred']).cuda()
    det, count = torch.ops.somi.nms(pred, 0.25, 0.45, False, False, 300)
    want = non_max_suppression(pred, 0.25, 0.45)
    for b, w in enumerate(want):
        assert int(count[b]) == w.shape[0] and torch.equal(det[b, :w.shape[0]], w)
    with pytest.raises(NotImplementedError, match="'CPU' backend"):
        torch.ops.somi.dcnv3_forward(x.cpu(), off.cpu(), m.cpu(), *cfg, 256)


@pytest.mark.parametrize('N,H,W,G,Gc,k,s,p,d,osc,spread', [(2, 40, 40, 8, 32, 3, 1, 1, 1, 1.0, 0.6),     # the bench graph's site shape
                                                           (2, 37, 29, 4, 16, 3, 1, 1, 1, 2.0, 0.5),     # ragged tiles, offset_scale 2
                                                           (3, 21, 17, 4, 8, 3, 2, 1, 1, 1.0, 0.8),      # stride 2
                                                           (1, 24, 24, 2, 64, 3, 1, 2, 2, 1.0, 0.7),     # dilation 2, 64-wide groups
                                                           (2, 19, 23, 2, 16, 5, 1, 2, 1, 1.0, 0.6),     # 5x5
                                                           (2, 33, 33, 8, 32, 3, 1, 1, 1, 1.0, 3.0),     # half the taps leave the window, all NEAR (MFMA form)
                                                           (2, 37, 29, 4, 16, 3, 1, 1, 1, 1.0, 3.0),     # ... the list form, ragged tiles
                                                           (2, 26, 41, 2, 64, 3, 1, 1, 1, 2.0, 2.5),     # ... 64-wide groups, offset_scale 2
                                                           (2, 33, 33, 8, 32, 3, 1, 1, 1, 1.0, 14.0)])   # most taps beyond the window, many FAR
def test_dcnv3_backward_windowed_form(N, H, W, G, Gc, k, s, p, d, osc, spread):
    """The windowed backward (grad_input summed per tile in LDS, staged, combined in a fixed order; taps that leave their tile's window
    but land within two tiles of it added by the owner-of-the-destination pass) against the CPU oracle's autograd, against the direct
    fp32-atomic form, and against itself: two launches are bit-identical whenever no tap went FARTHER than that - in particular at
    offsets of several pixels, which a trained offset branch produces (round 2 was only reproducible inside 2 px)."""
    from oracle.somi_ref import dcnv3 as O
    from somi_amd import ops
    from somi_amd.dcnv3 import dcnv3_backward
    g = torch.Generator().manual_seed(H * 131 + W + k)
    Ho, Wo = O.dcnv3_out_size(H, k, s, p, d), O.dcnv3_out_size(W, k, s, p, d)
    K = k * k
    x = torch.randn(N, H, W, G * Gc, generator=g)
    off = torch.randn(N, Ho, Wo, G * K * 2, generator=g) * spread
    if spread < 1.0:
        off.clamp_(-1.9 / osc, 1.9 / osc)                        # stay inside the window's 2 pixels of slack
    elif spread < 8.0:
        off.clamp_(-11.0 / osc, 11.0 / osc)                      # beyond the window, but never more than two 8-pixel tiles away
    m = torch.softmax(torch.randn(N, Ho, Wo, G, K, generator=g), -1).reshape(N, Ho, Wo, G * K)
    go = torch.randn(N, Ho, Wo, G * Gc, generator=g)
    go[0, :3] = 0                                               # an all-zero tile
    want = O.dcnv3_backward(x, off, m, k, k, s, s, p, p, d, d, G, Gc, osc, go, 256)
    dev = torch.device('cuda:0')
    args = (x.to(dev), off.to(dev), m.to(dev), k, k, s, s, p, p, d, d, G, Gc, osc, go.to(dev), 256)
    a = dcnv3_backward(*args)
    over = ops.dcn_overflow_taps()
    assert over is not None or k == 5, 'the windowed form did not run'      # 5x5: the tap lists exceed the LDS budget -> direct form
    for got, ref, what in zip(a, want, ('grad_input', 'grad_offset', 'grad_mask')):
        rel_close(got, ref, rel=1e-4, what=f'windowed {what}')
    b = dcnv3_backward(*args)
    if over is None:
        pass
    elif spread < 8.0:
        assert over == 0, f'{over} taps went through fp32 atomics at offsets ~N(0,{spread})'
        assert all(torch.equal(u, v) for u, v in zip(a, b)), 'two launches of the windowed backward differ'
    else:
        assert over > 0
    ops.DCN_DIRECT = True
    try:
        c = dcnv3_backward(*args)
    finally:
        ops.DCN_DIRECT = False
    for got, ref, what in zip(c, want, ('grad_input', 'grad_offset', 'grad_mask')):
        rel_close(got, ref, rel=1e-4, what=f'direct {what}')


@pytest.mark.parametrize('N,H,W,G,Gc,k,s,p,d,pad,direct', [(2, 40, 40, 8, 32, 3, 1, 1, 1, 0, False),     # the bench graph's site shape: 216-float rows
                                                           (2, 37, 29, 4, 16, 3, 1, 1, 1, 4, False),     # ragged tiles, padded rows
                                                           (3, 21, 17, 4, 8, 3, 2, 1, 1, 0, False),      # stride 2
                                                           (2, 19, 23, 2, 16, 5, 1, 2, 1, 2, True),      # 5x5 -> the tiled (non-windowed) kernels
                                                           (1, 9, 11, 4, 6, 3, 1, 1, 1, 0, True)])       # group width 6: scalar kernels
def test_dcnv3_merged_offset_mask_rows(N, H, W, G, Gc, k, s, p, d, pad, direct):
    """The operator over ONE [pixel][2GK offsets | GK masks | pad] tensor (what a single 1x1 GEMM over the module's stacked offset / mask
    weights produces) is bit-identical to the packed entry points, forward and backward, in every kernel form; the in-place strided group
    softmax and its backward equal the packed ones."""
    from oracle.somi_ref import dcnv3 as O
    from somi_amd import ops
    g = torch.Generator().manual_seed(N * 7 + H)
    dev = torch.device('cuda:0')
    Ho, Wo = O.dcnv3_out_size(H, k, s, p, d), O.dcnv3_out_size(W, k, s, p, d)
    K, GK = k * k, G * k * k
    x = torch.randn(N, H, W, G * Gc, generator=g).to(dev)
    off = (torch.randn(N, Ho, Wo, 2 * GK, generator=g) * 0.7).to(dev)
    mlog = torch.randn(N, Ho, Wo, GK, generator=g).to(dev)
    go = torch.randn(N, Ho, Wo, G * Gc, generator=g).to(dev)
    R = 3 * GK + pad
    om = torch.full((N, Ho, Wo, R), 7.0, device=dev)
    om[..., :2 * GK] = off
    om[..., 2 * GK:3 * GK] = mlog
    mask = ops.group_softmax(mlog, K)
    ops.group_softmax_cols_(om, G, K, 2 * GK)
    assert torch.equal(om[..., 2 * GK:3 * GK], mask) and torch.equal(om[..., :2 * GK], off)
    assert pad == 0 or bool((om[..., 3 * GK:] == 7.0).all()), 'the strided softmax wrote outside its columns'
    cfg = (k, k, s, s, p, p, d, d, G, Gc, 1.0, 256)
    ops.DCN_DIRECT = direct
    try:
        y0 = ops.dcnv3_forward_raw(x, off, mask, *cfg)
        y1 = ops.dcnv3_forward_merged(x, om, *cfg)
        assert torch.equal(y0, y1)
        gi0, goff0, gm0 = ops.dcnv3_backward_raw(x, off, mask, go, *cfg)
        gi1, d_om = ops.dcnv3_backward_merged(x, om, go, *cfg)
        far = ops.dcn_overflow_taps()
    finally:
        ops.DCN_DIRECT = False
    if Gc % 4:                                                   # the scalar kernel sums a pixel's channels with LDS atomics: order-dependent
        rel_close(d_om[..., :2 * GK], goff0, rel=1e-5, what='grad_offset')
        rel_close(d_om[..., 2 * GK:3 * GK], gm0, rel=1e-5, what='grad_mask')
        goff0, gm0 = d_om[..., :2 * GK].clone(), d_om[..., 2 * GK:3 * GK].contiguous()
    assert torch.equal(goff0, d_om[..., :2 * GK]) and torch.equal(gm0, d_om[..., 2 * GK:3 * GK])
    assert pad == 0 or bool((d_om[..., 3 * GK:] == 0).all())
    if direct or far:                                            # fp32 atomics into grad_input: order-dependent rounding
        rel_close(gi1, gi0.cpu(), rel=1e-4, what='grad_input')
    else:
        assert torch.equal(gi0, gi1)
    dlog = ops.group_softmax_backward(mask, gm0, K)
    ops.group_softmax_backward_cols_(om, d_om, G, K, 2 * GK)
    assert torch.equal(d_om[..., 2 * GK:3 * GK], dlog) and torch.equal(d_om[..., :2 * GK], goff0)


def test_dcnv3_full_size_properties():
    """The operator at the bench graph's full site shape (N32, 80x80, C256, G8, K9 - too large for the CPU oracle in a test) through
    properties that do not depend on the size: the forward is LINEAR in its input, the LDS-window form equals the tiled form and the
    windowed backward the direct (fp32-atomic) one, and grad_input is the ADJOINT of the forward: <go, f(x)> == <grad_input(go), x>."""
    import os
    from somi_amd import ops
    from somi_amd.dcnv3 import dcnv3_backward, dcnv3_forward
    d = torch.device('cuda:0')
    g = torch.Generator(device='cuda').manual_seed(5)
    N, H, C, G, k = 32, 80, 256, 8, 3
    K = k * k
    x1, x2 = (torch.randn(N, H, H, C, device=d, generator=g) for _ in range(2))
    off = torch.randn(N, H, H, G * K * 2, device=d, generator=g) * 0.7
    m = torch.softmax(torch.randn(N, H, H, G, K, device=d, generator=g), -1).reshape(N, H, H, G * K).contiguous()
    go = torch.randn(N, H, H, C, device=d, generator=g)
    cfg = (k, k, 1, 1, 1, 1, 1, 1, G, C // G, 1.0)
    f1, f2 = dcnv3_forward(x1, off, m, *cfg, 256), dcnv3_forward(x2, off, m, *cfg, 256)
    f12 = dcnv3_forward(0.5 * x1 - 2.0 * x2, off, m, *cfg, 256)
    rel_close(f12, 0.5 * f1 - 2.0 * f2, rel=1e-5, what='linearity of the forward in its input')
    gi, gof, gm = dcnv3_backward(x1, off, m, *cfg, go, 256)
    assert ops.dcn_overflow_taps() is not None, 'the windowed backward did not run'
    lhs, rhs = (go.double() * f1.double()).sum().item(), (gi.double() * x1.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-5 * (go.double() * f1.double()).abs().sum().item(), (lhs, rhs)
    # mask gradient: f is linear in the mask too, <go, f> == <grad_mask, mask>
    rhs_m = (gm.double() * m.double()).sum().item()
    assert abs(lhs - rhs_m) <= 1e-5 * (go.double() * f1.double()).abs().sum().item(), (lhs, rhs_m)
    os.environ['SOMI_DCN_DIRECT'] = '1'                           # the library reads it per call: tiled gathers from L2 ...
    ops.DCN_DIRECT = True                                         # ... and the one-kernel backward with fp32 atomics
    try:
        f1_t = dcnv3_forward(x1, off, m, *cfg, 256)
        gi_t, gof_t, gm_t = dcnv3_backward(x1, off, m, *cfg, go, 256)
    finally:
        os.environ['SOMI_DCN_DIRECT'] = '0'
        ops.DCN_DIRECT = False
    rel_close(f1, f1_t, rel=1e-5, what='LDS-window forward vs tiled forward')
    rel_close(gi, gi_t, rel=2e-5, what='windowed vs direct grad_input')
    rel_close(gof, gof_t, rel=2e-5, what='windowed vs direct grad_offset')
    rel_close(gm, gm_t, rel=2e-5, what='windowed vs direct grad_mask')
