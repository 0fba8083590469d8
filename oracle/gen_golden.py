#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python (read-only at /root/reference).

Test infrastructure; runs only in the build container (the reference never travels).  It imports
the reference through the stub harness documented in SURVEY.md section 8c / Appendix A: third-party
packages that are not installed (cv2, torchvision, ultralytics, timm, mmcv, ...) and `utils.plots`
(whose import would try a font download) are pre-seeded in sys.modules as inert stubs, the
reference's local `Conv` is installed where `models/common.py:9163` re-imports it from, and nothing
that touches the network is ever called.  Only library functions are executed.

Every fixture holds *inputs and expected outputs* (data); weights are regenerated from names by
`oracle.somi_ref.testing.fill_state`, so no reference state_dict or source is stored.

Usage:  PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python oracle/gen_golden.py
"""
import logging
import os
import sys
from unittest.mock import MagicMock

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden')
sys.dont_write_bytecode = True


class _Stub(MagicMock):
    __all__ = []


for _n in ['cv2', 'torchvision', 'torchvision.transforms', 'torchvision.ops', 'seaborn', 'thop', 'utils.plots', 'DCNv3',
           'ultralytics', 'ultralytics.nn', 'ultralytics.nn.modules', 'ultralytics.nn.modules.utils',
           'ultralytics.nn.modules.conv', 'ultralytics.utils', 'ultralytics.utils.tal', 'timm', 'timm.models',
           'timm.models.efficientnet_blocks', 'timm.models.layers', 'timm.models.layers.norm', 'monai',
           'monai.networks', 'monai.networks.blocks', 'mmcv', 'mmcv.cnn', 'mmcv.ops',
           'mmcv.ops.modulated_deform_conv', 'mmengine', 'mmengine.model']:
    sys.modules[_n] = _Stub()

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

logging.disable(logging.CRITICAL)
torch.set_num_threads(8)

_src = open(f'{REF}/models/common.py').read().split('\n')
_ns = {'nn': nn, 'torch': torch}
exec('\n'.join(_src[42:70]), _ns)                       # the reference's local autopad + Conv (common.py:43-70)
sys.modules['ultralytics.nn.modules.conv'].Conv = _ns['Conv']
sys.path.insert(0, REF)
import models.common as RC  # noqa: E402
import models.yolo as RY  # noqa: E402

RY.Segment = type('Segment', (nn.Module,), {})
from utils.loss import ComputeLoss as RefComputeLoss  # noqa: E402
from utils.metrics import bbox_iou as ref_bbox_iou, box_iou as ref_box_iou  # noqa: E402
from utils.torch_utils import fuse_conv_and_bn as ref_fuse  # noqa: E402
import utils.general as RG  # noqa: E402
import utils.RepulsionLoss as RR  # noqa: E402

sys.path.insert(0, f'{REF}/models/ops_dcnv3')
from functions.dcnv3_func import dcnv3_core_pytorch  # noqa: E402

sys.path.insert(0, ROOT)
from oracle.somi_ref.testing import fill_state, synthetic_batch, somi_cfg, tiny_somi_cfg, yolov5_cfg, SOMI_ANCHORS, HYP_VISDRONE  # noqa: E402
from oracle.somi_ref.nms import greedy_nms  # noqa: E402

# the NMS core is third-party (torchvision): install the restated greedy NMS so the reference's
# surrounding filtering / multi-label logic can run (SURVEY Appendix A note 2)
sys.modules['torchvision'].ops.nms = lambda b, s, t: greedy_nms(b, s, t)
RG.torchvision = sys.modules['torchvision']


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    flat = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        flat[k] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **flat)
    print(f'{name}: {sum(a.nbytes for a in flat.values()) / 1e3:.1f} kB')


# ------------------------------------------------------------------------------------------------ DCNv3
def dcn_case(tag, N, H, W, G, Gc, k, s, p, d, osc, seed, grads=True, dt=torch.float64):
    g = torch.Generator().manual_seed(seed)
    Ho = (H + 2 * p - (d * (k - 1) + 1)) // s + 1
    Wo = (W + 2 * p - (d * (k - 1) + 1)) // s + 1
    K = k * k
    x = (torch.rand(N, H, W, G * Gc, generator=g) * 0.01).to(dt)        # test.py:35-39 value ranges
    off = (torch.rand(N, Ho, Wo, G * K * 2, generator=g) * 10).to(dt)
    m = torch.rand(N, Ho, Wo, G, K, generator=g) + 1e-5
    m = (m / m.sum(-1, keepdim=True)).reshape(N, Ho, Wo, G * K).to(dt)
    x.requires_grad_(grads), off.requires_grad_(grads), m.requires_grad_(grads)
    out = dcnv3_core_pytorch(x, off, m, k, k, s, s, p, p, d, d, G, Gc, osc)
    rec = dict(input=x, offset=off, mask=m, output=out,
               params=np.array([N, H, W, G, Gc, k, s, p, d], dtype=np.int64), offset_scale=np.float64(osc))
    if grads:
        go = torch.rand(out.shape, generator=g).to(dt)
        gi, goff, gm = torch.autograd.grad(out, (x, off, m), go)
        rec.update(grad_output=go, grad_input=gi, grad_offset=goff, grad_mask=gm)
    save(f'dcnv3_{tag}', **rec)


def gen_dcnv3():
    # the reference test's fixture (models/ops_dcnv3/test.py:19-30): fp64 and fp32
    dcn_case('testpy_f64', 2, 8, 8, 4, 16, 3, 1, 1, 1, 2.0, 3)
    dcn_case('testpy_f32', 2, 8, 8, 4, 16, 3, 1, 1, 1, 2.0, 3, dt=torch.float32)
    # backward channel sweep of test.py:257-260 (N=2, M=2); 1025 omitted from the fixture for size
    for D in (1, 16, 30, 32, 64, 71):
        dcn_case(f'bwd_D{D}', 2, 8, 8, 2, D, 3, 1, 1, 1, 2.0, 3 + D)
    # geometry coverage the host launcher allows (dcnv3_cuda.cu:40-45)
    dcn_case('s2_p1', 1, 9, 11, 2, 8, 3, 2, 1, 1, 1.0, 11)
    dcn_case('d2_p2', 1, 10, 7, 3, 4, 3, 1, 2, 2, 1.5, 12)
    dcn_case('k5_p2', 2, 7, 7, 1, 8, 5, 1, 2, 1, 0.5, 13)
    dcn_case('p0', 1, 6, 6, 2, 4, 3, 1, 0, 1, 1.0, 14)


# ------------------------------------------------------------------------------------------------ blocks
def run_block(tag, mod, xs, train_too=True, seed=0):
    fill_state(mod, seed)
    RY.initialize_weights(mod)                                  # BN eps 1e-3 / momentum 0.03 as in Model.__init__
    rec = {}
    for i, x in enumerate(xs if isinstance(xs, (list, tuple)) else [xs]):
        rec[f'in{i}'] = x
    mod.eval()
    with torch.no_grad():
        y = mod(xs.clone() if isinstance(xs, torch.Tensor) else [t.clone() for t in xs])
    rec['out_eval'] = y
    if train_too:
        mod.train()
        with torch.no_grad():
            y = mod(xs.clone() if isinstance(xs, torch.Tensor) else [t.clone() for t in xs])
        rec['out_train'] = y
    save(f'block_{tag}', **rec)


def gen_blocks():
    g = torch.Generator().manual_seed(100)
    r = lambda *s: torch.randn(*s, generator=g)
    run_block('conv3x3_s1', RC.Conv(16, 24, 3, 1), r(2, 16, 12, 10))
    run_block('conv3x3_s2', RC.Conv(3, 16, 3, 2), r(2, 3, 13, 17))
    run_block('conv1x1', RC.Conv(40, 8, 1, 1), r(2, 40, 6, 6))
    run_block('c2fcbam_sc', RC.C2fCBAM(32, 32, 2, True), r(2, 32, 12, 12))
    run_block('c2fcbam_nosc', RC.C2fCBAM(48, 32, 1, False), r(2, 48, 9, 7))
    run_block('cbam_bneck', RC.CBAMBottleneck(32, 32, True, 1, k=(3, 3), e=1.0, ratio=16, kernel_size=7), r(2, 32, 10, 10))
    run_block('odconv_s2', RC.ODConv_3rd(16, 32, 3, 2, 4), r(3, 16, 12, 12))
    run_block('odconv_b1', RC.ODConv_3rd(16, 32, 3, 2, 4), r(1, 16, 12, 12), train_too=False)
    run_block('sppf', RC.SPPF(32, 32, 5), r(2, 32, 9, 9))
    run_block('bifpn2', RC.BiFPN(2), [r(2, 8, 6, 6), r(2, 8, 6, 6)], train_too=False)
    run_block('bifpn3', RC.BiFPN(3), [r(2, 8, 6, 6), r(2, 8, 6, 6), r(2, 8, 6, 6)], train_too=False)
    run_block('seam', RC.SEAM(32, 32, 1, 16), r(2, 32, 10, 10))
    run_block('decouple', RY.Decouple(64, 10, 4), r(2, 64, 8, 8))
    # fuse_conv_and_bn (utils/torch_utils.py:202-222)
    c = RC.Conv(8, 12, 3, 1)
    fill_state(c, 5)
    c.bn.eps = 1e-3
    f = ref_fuse(c.conv, c.bn)
    save('fuse_conv_bn', weight=f.weight, bias=f.bias)


# ------------------------------------------------------------------------------------------------ model
def build_ref_model(width, depth, anchors):
    cfg = somi_cfg(width, depth, anchors=anchors)
    cfg['backbone'] = [[f, n, {'Conv': 'Conv'}.get(m, m), a] for f, n, m, a in cfg['backbone']]
    m = RY.Model(cfg)
    fill_state(m, 1)
    return m


def gen_model():
    g = torch.Generator().manual_seed(200)
    for tag, width, depth, anchors, B, S in (('w025_anch4', 0.25, 0.33, 4, 2, 64),
                                             ('w025_visdrone', 0.25, 0.33, SOMI_ANCHORS, 2, 96),
                                             ('full', 1.0, 1.0, SOMI_ANCHORS, 1, 64)):
        m = build_ref_model(width, depth, anchors)
        x = torch.rand(B, 3, S, S, generator=g)
        m.eval()
        with torch.no_grad():
            z, raw = m(x.clone())
        rec = dict(x=x, z=z, stride=m.stride, anchors=m.model[-1].anchors,
                   nparams=np.int64(sum(p.numel() for p in m.parameters())))
        for i, t in enumerate(raw):
            rec[f'raw{i}'] = t
        # fused (Model.fuse) eval output, on a copy so that the BN statistics are the ones used above
        from copy import deepcopy
        mf = deepcopy(m).eval().fuse()
        with torch.no_grad():
            zf, _ = mf(x.clone())
        rec['z_fused'] = zf
        if B > 1:                                   # train-mode forward last: it updates the BN running statistics
            m.train()
            with torch.no_grad():
                tr = m(x.clone())
            for i, t in enumerate(tr):
                rec[f'train{i}'] = t
        save(f'model_{tag}', **rec)


def gen_tta():
    """Test-time augmentation: the reference's `Model.forward(x, augment=True)` (models/yolo.py:1253-1267,1292-1318) on the small SOMI
    graph (DecoupledDetect, 4 levels) and on yolov5 v6.0 (Detect, 3 levels, 3 classes), 64x96 images."""
    g = torch.Generator().manual_seed(950)
    x = torch.rand(2, 3, 64, 96, generator=g)
    m = build_ref_model(0.25, 0.33, SOMI_ANCHORS).eval()
    m2 = RY.Model(yolov5_cfg(0.25, 0.33, nc=3))
    fill_state(m2, 1)
    m2.eval()
    with torch.no_grad():
        save('model_tta', x=x, z_somi=m(x.clone(), augment=True)[0], z_yolov5=m2(x.clone(), augment=True)[0])


# ------------------------------------------------------------------------------------------------ stock YOLOv5 set
def gen_stock():
    """The stock YOLOv5 modules north_star names (BASELINE configs[0]) through the reference's own classes: Bottleneck, C3, SPP,
    Focus, Concat (models/common.py:1494-1509,1541-1565,1806-1826,1973-1997,2085-2097), Detect (models/yolo.py:46-109), and two
    whole graphs built by the reference's Model / parse_model from layer tables authored here (the reference ships no yolov5s.yaml):
    '6.0' (6x6 stem, SPPF) and '5.0' (Focus stem, SPP) at width 0.25, plus the parameter count of full yolov5s."""
    # Focus passes `act` positionally into Conv's dilation slot (models/common.py:1993 vs :55): dilation=True.  The reference's
    # torch 1.13 runs that as dilation 1; torch 2.10's conv2d rejects a bool, so the harness rewrites the stored tuple to ints
    # (same arithmetic, nothing else touched).
    focus_init = RC.Focus.__init__

    def _focus_init(self, *a, **k):
        focus_init(self, *a, **k)
        self.conv.conv.dilation = tuple(int(v) for v in self.conv.conv.dilation)
    RC.Focus.__init__ = _focus_init
    g = torch.Generator().manual_seed(700)
    r = lambda *s: torch.randn(*s, generator=g)                       # noqa: E731
    run_block('bottleneck_sc', RC.Bottleneck(16, 16, True, 1, k=((1, 1), (3, 3)), e=1.0), r(2, 16, 9, 11))
    run_block('bottleneck_nosc', RC.Bottleneck(16, 24, True), r(2, 16, 8, 8))
    run_block('c3_sc', RC.C3(32, 32, 2, True), r(2, 32, 10, 10))
    run_block('c3_nosc', RC.C3(48, 32, 1, False), r(2, 48, 7, 9))
    run_block('spp', RC.SPP(32, 32, (5, 9, 13)), r(2, 32, 11, 11))
    run_block('focus', RC.Focus(3, 16, 3), r(2, 3, 12, 16))
    run_block('conv6x6_s2', RC.Conv(3, 16, 6, 2, 2), r(2, 3, 20, 16))
    cat = RC.Concat(1)
    xs = [r(2, 8, 5, 5), r(2, 12, 5, 5)]
    save('block_concat', in0=xs[0], in1=xs[1], out_eval=cat(xs))
    det = RY.Detect(7, [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119]], [16, 24])
    fill_state(det, 3)
    det.stride = torch.tensor([8., 16.])
    det.anchors /= det.stride.view(-1, 1, 1)
    xs = [r(2, 16, 6, 6), r(2, 24, 3, 3)]
    det.eval()
    with torch.no_grad():
        z, raw = det([t.clone() for t in xs])
    save('block_detect', in0=xs[0], in1=xs[1], z=z, raw0=raw[0], raw1=raw[1], anchors=det.anchors, stride=det.stride)
    for tag, version, B, S in (('yolov5_v6', '6.0', 2, 64), ('yolov5_v5', '5.0', 2, 64)):
        m = RY.Model(yolov5_cfg(0.25, 0.33, nc=80, version=version))
        fill_state(m, 1)
        x = torch.rand(B, 3, S, S, generator=g)
        m.eval()
        with torch.no_grad():
            z, raw = m(x.clone())
        rec = dict(x=x, z=z, stride=m.stride, anchors=m.model[-1].anchors, nparams=np.int64(sum(p.numel() for p in m.parameters())))
        for i, t in enumerate(raw):
            rec[f'raw{i}'] = t
        from copy import deepcopy
        mf = deepcopy(m).eval().fuse()
        with torch.no_grad():
            rec['z_fused'] = mf(x.clone())[0]
        # one training step's worth of numbers: train-mode outputs, the loss the reference's ComputeLoss gives for them, and the
        # gradients of a few parameters spread over the graph (autograd through the reference's modules)
        m.train()
        m.hyp = dict(HYP_VISDRONE)
        m.zero_grad()
        _, targets = synthetic_batch(B, S, nc=80, seed=7)
        tr = m(x.clone())
        loss, items = RefComputeLoss(m)(tr, targets)
        loss.backward()
        for i, t in enumerate(tr):
            rec[f'train{i}'] = t
        rec.update(targets=targets, loss=loss, loss_items=items)
        names = [n for n, p in m.named_parameters() if p.grad is not None]
        pick = names[:: max(1, len(names) // 24)] + names[-2:]
        rec['grad_names'] = np.array(pick)
        pd = dict(m.named_parameters())
        for i, n in enumerate(pick):
            rec[f'grad{i}'] = pd[n].grad
        save(f'model_{tag}', **rec)
    full = RY.Model(yolov5_cfg())                                     # yolov5s: depth 0.33, width 0.50, nc 80
    save('model_yolov5s_meta', nparams=np.int64(sum(p.numel() for p in full.parameters())),
         nlayers=np.int64(len(list(full.modules()))), stride=full.stride, anchors=full.model[-1].anchors)


# ------------------------------------------------------------------------------------------------ DCNv3 module + graph wiring
def _ref_dcn_yolo_class():
    """DCNv3 -> BN -> SiLU around the reference's own `DCNv3_pytorch` module (models/ops_dcnv3/modules/dcnv3.py:95-219).  The
    reference wires DCNv3 into no graph; this wrapper is the harness's statement of the build's wiring, so that
    the reference's Model / parse_model (generic branch, models/yolo.py:1647-1648: channels pass through, yaml args verbatim) can
    build the whole graph around the reference's module."""
    from models.ops_dcnv3.modules.dcnv3 import DCNv3_pytorch

    class DCNv3_YOLO(nn.Module):
        def __init__(self, c, k=3, s=1, g=4, offset_scale=1.0, center_feature_scale=False):
            super().__init__()
            self.dcnv3 = DCNv3_pytorch(c, kernel_size=k, stride=s, pad=k // 2, group=g, offset_scale=offset_scale,
                                       center_feature_scale=center_feature_scale)
            self.bn = nn.BatchNorm2d(c)
            self.act = nn.SiLU()

        def forward(self, x):
            return self.act(self.bn(self.dcnv3(x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)))
    return DCNv3_pytorch, DCNv3_YOLO


def gen_dcn():
    import warnings
    warnings.simplefilter('ignore')
    RefDCN, RefDCNYolo = _ref_dcn_yolo_class()
    g = torch.Generator().manual_seed(800)
    # the module itself (NHWC in / out), forward + every gradient, with and without the centre feature scale
    for tag, cfs, C, G, osc in (('plain', False, 32, 4, 1.0), ('cfs', True, 32, 4, 2.0), ('g8', False, 64, 8, 1.0)):
        m = RefDCN(C, kernel_size=3, stride=1, pad=1, group=G, offset_scale=osc, center_feature_scale=cfs)
        fill_state(m, 11)
        x = torch.randn(2, 9, 11, C, generator=g, requires_grad=True)
        y = m(x)
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy)
        rec = dict(x=x, y=y, dy=dy, dx=x.grad, cfg=np.array([C, G, int(cfs)]), offset_scale=np.float64(osc))
        for n, p in m.named_parameters():
            rec['grad.' + n] = p.grad
        save(f'dcnv3_module_{tag}', **rec)
    # the wired block: eval and train-mode forward
    run_block('dcnv3_yolo', RefDCNYolo(32, 3, 1, 4), torch.randn(2, 32, 10, 12, generator=g), seed=2)
    # the whole SOMI graph with the two DCNv3 sites, built by the reference's Model (width 0.25: 64 channels, 8 groups of 8)
    RY.DCNv3_YOLO = RefDCNYolo
    cfg = somi_cfg(0.25, 0.33, anchors=SOMI_ANCHORS, dcn=True)
    for l in cfg['head']:
        if l[2] == 'DCNv3_YOLO':
            l[3][0] = 64                                         # the reference's generic branch passes yaml args verbatim
    m = RY.Model(cfg)
    fill_state(m, 1)
    B, S = 2, 64
    x = torch.rand(B, 3, S, S, generator=g)
    m.eval()
    with torch.no_grad():
        z, raw = m(x.clone())
    rec = dict(x=x, z=z, stride=m.stride, nparams=np.int64(sum(p.numel() for p in m.parameters())))
    for i, t in enumerate(raw):
        rec[f'raw{i}'] = t
    m.train()
    m.hyp = dict(HYP_VISDRONE)
    m.zero_grad()
    _, targets = synthetic_batch(B, S, seed=9)
    tr = m(x.clone())
    loss, items = RefComputeLoss(m)(tr, targets)
    loss.backward()
    for i, t in enumerate(tr):
        rec[f'train{i}'] = t
    rec.update(targets=targets, loss=loss, loss_items=items)
    names = [n for n, p in m.named_parameters() if p.grad is not None and ('dcnv3' in n or n.startswith('model.10.') or n.startswith('model.2.cv1'))]
    rec['grad_names'] = np.array(names)
    pd = dict(m.named_parameters())
    for i, n in enumerate(names):
        rec[f'grad{i}'] = pd[n].grad
    save('model_w025_dcn', **rec)


# ------------------------------------------------------------------------------------------------ checkpoint (N4)
def gen_ckpt():
    """A checkpoint exactly as train.py:310-317 writes it - `torch.save` of a dict whose 'model' / 'ema' entries are whole pickled
    module objects of the reference's own classes (models.yolo.Model, models.common.*, and `Conv` under the module path the reference
    really gets it from, ultralytics.nn.modules.conv), in half precision.  The fixture is the file's bytes; the expected
    weights are regenerated from names by fill_state.  ema: a cut-down SOMI graph using every SOMI module class (seed 4); model: yolov5 v6.0 at
    width 0.125 (seed 5)."""
    import io
    from copy import deepcopy
    conv_cls = sys.modules['ultralytics.nn.modules.conv'].Conv
    conv_cls.__module__, conv_cls.__qualname__ = 'ultralytics.nn.modules.conv', 'Conv'
    ema = fill_state(RY.Model(tiny_somi_cfg()), 4)
    mdl = fill_state(RY.Model(yolov5_cfg(0.125, 0.33, nc=80)), 5)
    ema.names = [f'class{i}' for i in range(10)]
    ema.hyp = dict(HYP_VISDRONE)
    ckpt = {'epoch': 12, 'best_fitness': np.array([0.4321]), 'model': deepcopy(mdl).half(), 'ema': deepcopy(ema).half(), 'updates': 345,
            'optimizer': None, 'wandb_id': None, 'date': '2026-01-01T00:00:00'}
    buf = io.BytesIO()
    torch.save(ckpt, buf)
    x = torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(900))
    ema.eval()
    with torch.no_grad():
        z, _ = deepcopy(ema).half().float()(x)                   # what attempt_load(...).float() evaluates (models/experimental.py:96-100)
    save('checkpoint_ref', bytes=np.frombuffer(buf.getvalue(), dtype=np.uint8), x=x, z=z,
         nparams_ema=np.int64(sum(p.numel() for p in ema.parameters())), nparams_model=np.int64(sum(p.numel() for p in mdl.parameters())))


# ------------------------------------------------------------------------------------------------ loss
def gen_loss():
    m = build_ref_model(0.25, 0.33, SOMI_ANCHORS)
    m.hyp = dict(HYP_VISDRONE)
    crit = RefComputeLoss(m)
    det = m.model[-1]
    for tag, B, S, seed, nt_mode in (('a', 2, 64, 0, 'normal'), ('b', 3, 128, 1, 'normal'), ('empty', 2, 64, 2, 'empty'),
                                     ('edge', 2, 64, 3, 'edge')):
        g = torch.Generator().manual_seed(300 + seed)
        p = [torch.randn(B, det.na, S // int(s), S // int(s), det.no, generator=g).requires_grad_(True) for s in m.stride]
        _, targets = synthetic_batch(B, S, seed=seed)
        if nt_mode == 'empty':
            targets = torch.zeros(0, 6)
        elif nt_mode == 'edge':
            targets = targets[:12].clone()
            targets[:6, 2:4] = torch.tensor([[0.0, 0.0], [1.0, 1.0], [0.999, 0.001], [0.5, 0.5], [0.25, 0.75], [0.5, 0.0]])
            targets[6:, 4:6] = torch.tensor([[0.5, 0.5], [0.004, 0.004], [0.3, 0.01], [0.01, 0.3], [0.05, 0.05], [0.1, 0.2]])
            targets[3:5, 2:6] = targets[3:4, 2:6]                  # duplicate box -> same cell twice
        loss, items = crit(p, targets)
        grads = torch.autograd.grad(loss, p, allow_unused=True)
        tcls, tbox, indices, anch = crit.build_targets(p, targets)
        rec = dict(targets=targets, loss=loss, items=items, anchors=det.anchors, stride=m.stride)
        for i in range(len(p)):
            rec[f'p{i}'] = p[i]
            rec[f'g{i}'] = grads[i] if grads[i] is not None else torch.zeros_like(p[i])
            rec[f'tcls{i}'], rec[f'tbox{i}'], rec[f'anch{i}'] = tcls[i], tbox[i], anch[i]
            rec[f'idx{i}'] = torch.stack(indices[i], 0) if indices[i][0].numel() else torch.zeros(4, 0, dtype=torch.long)
        save(f'loss_{tag}', **rec)
    # the branches hyp.VisDrone.yaml leaves off, through the reference's own ComputeLoss under modified hyper-parameters: FocalLoss
    # (utils/loss.py:35-60), SlideLoss (:378-402), both stacked (:125-131), the NWD box term (:162-169, utils/metrics.py:341-354),
    # label smoothing (:123); same predictions / targets as fixture `a`
    g = torch.Generator().manual_seed(300)
    B, S = 2, 64
    p0 = [torch.randn(B, det.na, S // int(s), S // int(s), det.no, generator=g) for s in m.stride]
    _, targets = synthetic_batch(B, S, seed=0)
    for tag, over in (('focal', dict(fl_gamma=1.5)), ('slide', dict(slide_ratio=1.0)), ('focal_slide', dict(fl_gamma=2.0, slide_ratio=1.0)),
                      ('nwd', dict(nwdloss=1.0)), ('all', dict(fl_gamma=1.5, slide_ratio=1.0, nwdloss=1.0, label_smoothing=0.1))):
        m.hyp = dict(HYP_VISDRONE, **over)
        crit_b = RefComputeLoss(m)
        p = [t.clone().requires_grad_(True) for t in p0]
        loss, items = crit_b(p, targets)
        grads = torch.autograd.grad(loss, p, allow_unused=True)
        rec = dict(targets=targets, loss=loss, items=items, anchors=det.anchors, stride=m.stride,
                   hyp_keys=np.array(sorted(over)), hyp_vals=np.array([over[k] for k in sorted(over)], dtype=np.float64))
        for i in range(len(p)):
            rec[f'p{i}'] = p[i]
            rec[f'g{i}'] = grads[i] if grads[i] is not None else torch.zeros_like(p[i])
        save(f'loss_branch_{tag}', **rec)
    # ... the shapeloss NWD variant (utils/loss.py:163-164 -> utils/metrics.py:373) and autobalance (:137,197-201: three consecutive calls,
    # the balance list after each; the first call's weights are scaled so that the objectness means differ between the levels)
    m.hyp = dict(HYP_VISDRONE, nwdloss=1.0, shapeloss=1.0)
    crit_b = RefComputeLoss(m)
    p = [t.clone().requires_grad_(True) for t in p0]
    loss, items = crit_b(p, targets)
    grads = torch.autograd.grad(loss, p, allow_unused=True)
    rec = dict(targets=targets, loss=loss, items=items, anchors=det.anchors, stride=m.stride, hyp_keys=np.array(['nwdloss', 'shapeloss']),
               hyp_vals=np.array([1.0, 1.0]))
    for i in range(len(p)):
        rec[f'p{i}'] = p[i]
        rec[f'g{i}'] = grads[i] if grads[i] is not None else torch.zeros_like(p[i])
    save('loss_branch_shapeloss', **rec)
    m.hyp = dict(HYP_VISDRONE)
    crit_b = RefComputeLoss(m, autobalance=True)
    rec = dict(targets=targets, anchors=det.anchors, stride=m.stride, ssi=np.array(crit_b.ssi))
    for call in range(3):
        p = [(t * (1.0 + 0.5 * call) + 0.3 * i * call).clone().requires_grad_(True) for i, t in enumerate(p0)]
        loss, items = crit_b(p, targets)
        grads = torch.autograd.grad(loss, p, allow_unused=True)
        rec[f'loss{call}'], rec[f'items{call}'], rec[f'balance{call}'] = loss, items, np.array(crit_b.balance, dtype=np.float64)
        for i in range(len(p)):
            rec[f'c{call}_p{i}'] = p[i]
            rec[f'c{call}_g{i}'] = grads[i] if grads[i] is not None else torch.zeros_like(p[i])
    save('loss_autobalance', **rec)
    m.hyp = dict(HYP_VISDRONE)
    # bbox_iou CIoU (utils/metrics.py:476-518)
    g = torch.Generator().manual_seed(350)
    b1 = torch.rand(64, 4, generator=g) * torch.tensor([20, 20, 8, 8.]) + torch.tensor([0, 0, 0.05, 0.05])
    b2 = torch.rand(64, 4, generator=g) * torch.tensor([20, 20, 8, 8.]) + torch.tensor([0, 0, 0.05, 0.05])
    b2[:8] = b1[:8]
    save('ciou', box1=b1, box2=b2, ciou=ref_bbox_iou(b1.T, b2, x1y1x2y2=False, CIoU=True))
    # repulsion loss (utils/RepulsionLoss.py:47-95); its hard-coded .cuda() hops are made identity on CPU
    torch.Tensor.cuda = lambda self, *a, **k: self
    g = torch.Generator().manual_seed(360)
    ctr = torch.rand(2, 40, 2, generator=g) * 50
    wh = torch.rand(2, 40, 2, generator=g) * 20 + 2
    pb = torch.cat((ctr - wh / 2, ctr + wh / 2), -1)
    gt_pool = torch.cat((ctr[:, :6] - 8, ctr[:, :6] + 8), -1)
    gb = gt_pool[:, torch.randint(0, 6, (40,), generator=g)]
    fg = torch.rand(2, 40, generator=g) > 0.4
    rgt, rbox = RR.repulsion_loss(pb, gb, fg, sigma_repgt=0.9, sigma_repbox=0)
    save('repulsion', pbox=pb, gtbox=gb, fg=fg, rep_gt=rgt, rep_box=rbox)


# ------------------------------------------------------------------------------------------------ nms
def gen_nms():
    m = build_ref_model(0.25, 0.33, SOMI_ANCHORS)
    m.eval()
    g = torch.Generator().manual_seed(400)
    x = torch.rand(2, 3, 96, 96, generator=g)
    with torch.no_grad():
        z, _ = m(x)
    # spread objectness so that several thresholds select different subsets
    z = z.clone()
    z[..., 4] = torch.rand(z.shape[:2], generator=g) ** 3
    z[..., 5:] = torch.rand(z[..., 5:].shape, generator=g)
    z[..., :2] = torch.rand(z[..., :2].shape, generator=g) * 96
    z[..., 2:4] = torch.rand(z[..., 2:4].shape, generator=g) * 30 + 2
    z[0, 10:14, :4] = z[0, 10, :4]                                   # exact duplicates -> ties
    z[0, 10:14, 4:] = z[0, 10, 4:]
    cases = dict(default=dict(conf_thres=0.25, iou_thres=0.45),
                 val=dict(conf_thres=0.4, iou_thres=0.2, multi_label=True),
                 bench=dict(conf_thres=0.001, iou_thres=0.6, multi_label=True),
                 agnostic=dict(conf_thres=0.3, iou_thres=0.5, agnostic=True),
                 classes=dict(conf_thres=0.2, iou_thres=0.45, classes=[1, 3, 7]),
                 maxdet=dict(conf_thres=0.05, iou_thres=0.9, multi_label=True, max_det=20),
                 none=dict(conf_thres=0.9999, iou_thres=0.45))
    rec = dict(pred=z)
    for tag, kw in cases.items():
        out = RG.non_max_suppression(z.clone(), **kw)
        for b, o in enumerate(out):
            rec[f'{tag}_{b}'] = o
    # a-priori labels (autolabelling, utils/general.py:651-658; val.py:161-164 passes `lb` when save_hybrid): [cls, x, y, w, h] in pixels
    lb = [torch.tensor([[3., 40., 40., 20., 24.], [0., 12., 70., 10., 10.], [7., z[0, 10, 0], z[0, 10, 1], z[0, 10, 2], z[0, 10, 3]]]),
          torch.zeros(0, 5)]
    out = RG.non_max_suppression(z.clone(), conf_thres=0.3, iou_thres=0.5, labels=lb, multi_label=True)
    rec['labels_in0'] = lb[0]
    for b, o in enumerate(out):
        rec[f'withlabels_{b}'] = o
    save('nms', **rec)
    b1 = z[0, :50, :4].clone(); b1[:, 2:] += b1[:, :2]
    b2 = z[1, :40, :4].clone(); b2[:, 2:] += b2[:, :2]
    save('box_iou', box1=b1, box2=b2, iou=ref_box_iou(b1, b2))


# ------------------------------------------------------------------------------------------------ validation metrics
def gen_val():
    """process_batch (val.py:50-71, executed from the reference's own source text) and ap_per_class
    (utils/metrics.py:21-74) on seeded detections / labels of 24 synthetic images."""
    from utils.metrics import ap_per_class as ref_ap
    src = open(f'{REF}/val.py').read().split('\n')
    lo = next(i for i, l in enumerate(src) if l.startswith('def process_batch'))
    hi = next(i for i in range(lo + 1, len(src)) if src[i].startswith('@torch.no_grad') or src[i].startswith('def '))
    ns = {'np': np, 'torch': torch, 'box_iou': ref_box_iou}
    exec('\n'.join(src[lo:hi]), ns)
    ref_process_batch = ns['process_batch']
    g = torch.Generator().manual_seed(500)
    iouv = torch.linspace(0.5, 0.95, 10)
    rec = dict(iouv=iouv)
    stats = []
    nimg = 24
    for b in range(nimg):
        M = int(torch.randint(0, 40, (1,), generator=g)) if b != 3 else 0          # image 3 has no labels
        N = int(torch.randint(0, 120, (1,), generator=g)) if b != 5 else 0         # image 5 has no detections
        lc = torch.rand(M, 2, generator=g) * 500 + 50
        lwh = torch.rand(M, 2, generator=g) * 80 + 8
        labels = torch.cat((torch.randint(0, 6, (M, 1), generator=g).float(), lc - lwh / 2, lc + lwh / 2), 1)
        # detections: jittered copies of labels (so many IoU levels are exercised) plus clutter
        if M and N:
            pick = torch.randint(0, M, (N,), generator=g)
            jit = (torch.rand(N, 4, generator=g) - 0.5) * lwh[pick].repeat(1, 2) * torch.rand(N, 1, generator=g) * 0.9
            boxes = labels[pick, 1:] + jit
            cls = labels[pick, 0].clone()
            flip = torch.rand(N, generator=g) < 0.25
            cls[flip] = torch.randint(0, 7, (int(flip.sum()),), generator=g).float()
            clutter = torch.rand(N, generator=g) < 0.3
            cc = torch.rand(N, 2, generator=g) * 500 + 50
            cw = torch.rand(N, 2, generator=g) * 80 + 8
            boxes[clutter] = torch.cat((cc - cw / 2, cc + cw / 2), 1)[clutter]
        else:
            cc = torch.rand(N, 2, generator=g) * 500 + 50
            cw = torch.rand(N, 2, generator=g) * 80 + 8
            boxes = torch.cat((cc - cw / 2, cc + cw / 2), 1)
            cls = torch.randint(0, 7, (N,), generator=g).float()
        conf = torch.rand(N, generator=g)
        det = torch.cat((boxes, conf[:, None], cls[:, None]), 1)
        if N and M:
            correct = ref_process_batch(det, labels, iouv)
        else:
            correct = torch.zeros(N, 10, dtype=torch.bool)                          # val.py:174,188
        rec[f'det{b}'], rec[f'lab{b}'], rec[f'correct{b}'] = det, labels, correct
        stats.append((correct, det[:, 4], det[:, 5], labels[:, 0]))
    tp, conf, pcls, tcls = [torch.cat(x, 0).numpy() for x in zip(*stats)]            # val.py:200
    p, r, ap, f1, cls_ = ref_ap(tp, conf, pcls, tcls, plot=False, names={})
    rec.update(nimg=nimg, tp=tp, conf=conf, pred_cls=pcls, target_cls=tcls, p=p, r=r, ap=ap, f1=f1, ap_class=cls_)
    # the confusion matrix of val.py:141,186 (utils/metrics.py:98-142) over the same images; val.py only feeds it images that have
    # both labels and predictions (val.py:171-186)
    from utils.metrics import ConfusionMatrix as RefConfusion
    cm = RefConfusion(nc=7)                                                          # detections carry classes 0..6
    for b in range(nimg):
        if len(rec[f'det{b}']) and len(rec[f'lab{b}']):
            cm.process_batch(rec[f'det{b}'], rec[f'lab{b}'])
    rec.update(confusion=cm.matrix, confusion_nc=7, confusion_conf=cm.conf, confusion_iou=cm.iou_thres)
    save('val_metrics', **rec)


AUGMENT_CASES = (        # name, hyp overrides, augment, seeds
    ('mosaic', {}, True, (0, 1, 2, 3, 4, 5)),
    ('mixup', dict(mixup=1.0), True, (10, 11, 12)),
    ('general', dict(degrees=10.0, translate=0.1, shear=5.0, flipud=0.5, mixup=0.5), True, (20, 21, 22, 23)),
    ('single', dict(mosaic=0.0), True, (30, 31, 32, 33)),
    ('val', {}, False, (40, 41, 42)),
)


def gen_augment():
    """The reference's own `LoadImagesAndLabels.__getitem__` (utils/datasets.py:590-673: load_mosaic, random_perspective,
    mixup, augment_hsv, flips, letterbox) on a cached synthetic image set.  cv2 is not installed: `oracle.somi_ref.cv_port`
    (the restated OpenCV arithmetic, parity unpinned) stands in for it, so this pins everything around the pixel kernels -
    random draw order, mosaic geometry, matrices, label transforms - through the reference's code."""
    import random
    from types import SimpleNamespace
    import utils.augmentations as RA
    import utils.datasets as RD
    from oracle.somi_ref import cv_port
    from oracle.somi_ref.testing import HYP_AUGMENT, synthetic_image_set
    RA.cv2 = RD.cv2 = cv_port
    S = 64
    imgs, labels = synthetic_image_set(S, n=6, seed=600)
    rec = {'img_size': S, 'n': len(imgs)}
    for i, (im, lab) in enumerate(zip(imgs, labels)):
        rec[f'src{i}'], rec[f'lab{i}'] = im, lab
    names, seeds, indices = [], [], []
    for name, over, augment, case_seeds in AUGMENT_CASES:
        hyp = dict(HYP_AUGMENT, **over)
        ds = SimpleNamespace(indices=range(len(imgs)), n=len(imgs), hyp=hyp, augment=augment, rect=False, mosaic=augment,
                             img_size=S, mosaic_border=[-S // 2, -S // 2], imgs=imgs,
                             img_hw0=[im.shape[:2] for im in imgs], img_hw=[im.shape[:2] for im in imgs],
                             labels=labels, segments=[[] for _ in imgs], img_files=[f'{i}.jpg' for i in range(len(imgs))],
                             albumentations=RA.Albumentations() if augment else None)
        for seed in case_seeds:
            random.seed(seed)
            np.random.seed(seed)
            index = seed % len(imgs)
            img, lab, _, shapes = RD.LoadImagesAndLabels.__getitem__(ds, index)
            k = len(names)
            rec[f'out_img{k}'], rec[f'out_lab{k}'] = img, lab
            rec[f'out_pad{k}'] = np.array(shapes[1][1] if shapes else (-1.0, -1.0))
            names.append(name), seeds.append(seed), indices.append(index)
    rec.update(case=np.array(names), seed=np.array(seeds), index=np.array(indices))
    save('augment', **rec)

    # rectangular validation batches (val.py:129-138: rect=True, pad=0.5): the aspect-ratio ordering and batch shapes come from
    # the reference's own source text (datasets.py "if self.rect:" block, executed on a stand-in `self`), the samples from its
    # __getitem__ with augment=False
    import textwrap
    src = open(f'{REF}/utils/datasets.py').read().split('\n')
    lo = next(i for i, l in enumerate(src) if l.strip() == 'if self.rect:' and 'self.shapes' in ''.join(src[i:i + 4]))
    hi = next(i for i in range(lo, len(src)) if 'self.batch_shapes =' in src[i]) + 1
    block = textwrap.dedent('\n'.join(src[lo:hi]))
    S, bs, stride, pad = 96, 3, 32, 0.5
    imgs, labels = synthetic_image_set(S, n=8, seed=601)
    n = len(imgs)
    bi = np.floor(np.arange(n) / bs).astype(int)
    ds = SimpleNamespace(rect=True, shapes=np.array([(im.shape[1], im.shape[0]) for im in imgs], dtype=np.float64),
                         img_files=list(range(n)), label_files=list(range(n)), labels=list(labels))
    exec(block, {'self': ds, 'np': np, 'nb': bi[-1] + 1, 'bi': bi, 'img_size': S, 'stride': stride, 'pad': pad})
    order = np.array(ds.img_files)
    ds.__dict__.update(indices=range(n), n=n, hyp=dict(HYP_AUGMENT), augment=False, mosaic=False, img_size=S, batch=bi,
                       imgs=[imgs[i] for i in order], img_hw0=[imgs[i].shape[:2] for i in order],
                       img_hw=[imgs[i].shape[:2] for i in order], segments=[[] for _ in imgs], albumentations=None)
    rec = dict(img_size=S, n=n, batch_size=bs, stride=stride, pad=pad, order=order, batch_shapes=ds.batch_shapes)
    for i, (im, lab) in enumerate(zip(imgs, labels)):
        rec[f'src{i}'], rec[f'lab{i}'] = im, lab
    for k in range(n):
        img, lab, _, shapes = RD.LoadImagesAndLabels.__getitem__(ds, k)
        rec[f'out_img{k}'], rec[f'out_lab{k}'], rec[f'out_pad{k}'] = img, lab, np.array(shapes[1][1])
    save('augment_rect', **rec)


if __name__ == '__main__':
    which = sys.argv[1:] or ['dcnv3', 'blocks', 'model', 'tta', 'stock', 'dcn', 'ckpt', 'loss', 'nms', 'val', 'augment']
    for w in which:
        globals()[f'gen_{w}']()
