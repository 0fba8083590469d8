"""Whole-model parity on the MI355X: somi_amd.Model (HIP kernels through the C ABI) against the golden outputs
of the reference model (tests/golden/model_*.npz) and against the CPU oracle at other shapes.
Bar: 1e-3 relative in fp32 (BASELINE.json north_star)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def rel_close(got, want, rel=1e-3, what=''):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = (got - want).abs().max().item()
    scale = want.abs().max().item() + 1e-12
    assert err <= rel * scale, f'{what}: max err {err:.3e} vs scale {scale:.3e}'


def build(width, depth, anchors, seed=1):
    from oracle.somi_ref import Model as OModel
    from oracle.somi_ref.testing import fill_state, somi_cfg
    from somi_amd.model import Model
    cfg = somi_cfg(width, depth, anchors=anchors)
    ref = fill_state(OModel(cfg), seed).eval()
    mine = Model(cfg)
    mine.load_state_dict(ref.state_dict())
    return ref, mine.cuda().eval()


@pytest.mark.parametrize('tag,width,depth,anch', [('w025_anch4', 0.25, 0.33, 4), ('w025_visdrone', 0.25, 0.33, 'v'),
                                                  ('full', 1.0, 1.0, 'v')])
def test_model_matches_reference_vectors(golden, tag, width, depth, anch):
    from oracle.somi_ref.testing import SOMI_ANCHORS
    g = golden('model_' + tag)
    _, mine = build(width, depth, SOMI_ANCHORS if anch == 'v' else anch)
    x = T(g['x']).cuda()
    with torch.no_grad():
        z, raw = mine(x)
    torch.cuda.synchronize()
    rel_close(z, T(g['z']), what='z')
    rel_close(z, T(g['z_fused']), what='z vs Model.fuse() output')
    for i, r in enumerate(raw):
        rel_close(r, T(g[f'raw{i}']), what=f'raw{i}')


def test_model_uint8_batch_matches_oracle():
    """The reference's batch contract: uint8 images, /255 in the loop (train.py:249, val.py:150-152)."""
    from oracle.somi_ref.testing import SOMI_ANCHORS, synthetic_batch
    ref, mine = build(0.25, 0.33, SOMI_ANCHORS, seed=2)
    imgs, _ = synthetic_batch(3, 128, seed=5)
    with torch.no_grad():
        zr, rr = ref(imgs.float() / 255)
        z, raw = mine(imgs.cuda())
    rel_close(z, zr, what='z')
    for a, b in zip(raw, rr):
        rel_close(a, b, what='raw')


def test_single_image_skips_odconv_bn():
    """ODConv drops its squeeze BN for a batch of one (models/common.py:4562)."""
    from oracle.somi_ref.testing import SOMI_ANCHORS
    ref, mine = build(0.25, 0.33, SOMI_ANCHORS, seed=3)
    x = torch.rand(1, 3, 64, 96, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        rel_close(mine(x.cuda())[0], ref(x)[0], what='z (B=1, non-square)')


def test_uavdt_config_1280_nc3():
    """BASELINE configs[3] shape: full-width SOMI with nc=3 (UAVDT) at 1280x1280 - grids 320/160/80/40, 544 000 predictions per
    image - one image against the CPU oracle, then NMS on all of them (selection bit-exact)."""
    from oracle.somi_ref import Model as OModel
    from oracle.somi_ref.nms import non_max_suppression as oracle_nms
    from oracle.somi_ref.testing import SOMI_ANCHORS, fill_state, somi_cfg
    from somi_amd.model import Model
    from somi_amd.nms import non_max_suppression
    cfg = somi_cfg(1.0, 1.0, nc=3, anchors=SOMI_ANCHORS)
    ref = fill_state(OModel(cfg), 7).eval()
    mine = Model(cfg)
    mine.load_state_dict(ref.state_dict())
    mine = mine.cuda().eval()
    x = torch.rand(1, 3, 1280, 1280, generator=torch.Generator().manual_seed(12))
    with torch.no_grad():
        zr, _ = ref(x)
        z, _ = mine(x.cuda())
    assert z.shape == (1, 544000, 8)
    rel_close(z, zr, what='z @1280 nc=3')
    det = non_max_suppression(z, 0.25, 0.45, multi_label=True)
    want = oracle_nms(z.cpu(), 0.25, 0.45, multi_label=True)
    assert det[0].shape == want[0].shape and torch.equal(det[0].cpu(), want[0])


def test_val_config_bs128_nms_wbf():
    """BASELINE configs[4] shape: inference at batch 128, 640x640, NMS + WBF over two models.  The CPU oracle checks the first two
    images of each model; the rest of the batch is covered by properties that do not need the oracle: every image's
    predictions equal (to fp32 rounding - the tile schedule depends on the batch) those of the same image run in a batch of 8,
    NMS of the big batch equals NMS of its slices, and WBF of a model with itself keeps every cluster's label and box."""
    import numpy as np
    from oracle.somi_ref import Model as OModel
    from oracle.somi_ref.nms import non_max_suppression as oracle_nms
    from oracle.somi_ref.testing import SOMI_ANCHORS, fill_state, somi_cfg, synthetic_batch
    from oracle.somi_ref.wbf import weighted_boxes_fusion as oracle_wbf
    from somi_amd.model import Model
    from somi_amd.nms import non_max_suppression
    from somi_amd.wbf import weighted_boxes_fusion
    cfg = somi_cfg(1.0, 1.0, anchors=SOMI_ANCHORS)
    imgs, _ = synthetic_batch(128, 640, seed=21)
    dets = []
    for seed in (1, 2):                                          # two "models" for the fusion step (wbf.py:9-78)
        ref = fill_state(OModel(cfg), seed).eval()
        mine = Model(cfg)
        mine.load_state_dict(ref.state_dict())
        mine = mine.cuda().eval()
        with torch.no_grad():
            z, _ = mine(imgs.cuda())
            z8, _ = mine(imgs[40:48].cuda())
            zr, _ = ref(imgs[:2].float() / 255)
        assert z.shape == (128, 136000, 15)
        rel_close(z[:2], zr, what=f'model {seed}: first two images vs oracle')
        rel_close(z[40:48], z8, rel=1e-5, what=f'model {seed}: batch 128 vs batch 8')
        det = non_max_suppression(z, 0.4, 0.2, multi_label=True)                     # val.py:77-78 defaults
        det8 = non_max_suppression(z[40:48].contiguous(), 0.4, 0.2, multi_label=True)
        for a, b in zip(det[40:48], det8):
            assert torch.equal(a, b)
        want = oracle_nms(z[:2].cpu(), 0.4, 0.2, multi_label=True)
        for a, b in zip(det[:2], want):
            assert a.shape == b.shape and torch.equal(a.cpu(), b)
        dets.append([d.cpu() for d in det])
        del mine, z
        torch.cuda.empty_cache()
    for i in (0, 1, 77):                                          # fuse the two models' boxes of one image, like wbf.py:55-68
        bl = [(d[i][:, :4] / 640).clamp(0, 1).numpy() for d in dets]
        sl = [d[i][:, 4].numpy() for d in dets]
        ll = [d[i][:, 5].numpy().astype(np.int64) for d in dets]
        gb, gs, gl = weighted_boxes_fusion(bl, sl, ll, weights=None, iou_thr=0.67, skip_box_thr=0.01)
        wb, ws, wl = oracle_wbf([b.tolist() for b in bl], [s.tolist() for s in sl], [l.tolist() for l in ll], weights=None,
                                iou_thr=0.67, skip_box_thr=0.01)
        assert np.array_equal(gl, wl) and np.array_equal(gb, wb.astype(np.float32)) and np.array_equal(gs, ws.astype(np.float32))


def test_attempt_load_reference_style_checkpoint():
    """somi_amd.checkpoint.attempt_load: a pickled fp16 module tree (train.py:310-317) -> somi_amd.Model with the EMA weights,
    without importing the code base that pickled it; predictions equal the oracle run with the same fp16-rounded weights."""
    import copy
    import io
    from oracle.somi_ref import Model as OModel
    from oracle.somi_ref.testing import SOMI_ANCHORS, fill_state, somi_cfg
    from somi_amd.checkpoint import attempt_load
    cfg = somi_cfg(0.25, 0.33, anchors=SOMI_ANCHORS)
    ema, raw = fill_state(OModel(cfg), 5), fill_state(OModel(cfg), 6)
    ema.names = [f'c{i}' for i in range(10)]
    buf = io.BytesIO()
    torch.save({'epoch': 3, 'best_fitness': 0.5, 'model': copy.deepcopy(raw).half(), 'ema': copy.deepcopy(ema).half(), 'updates': 40}, buf)
    model, info = attempt_load(buf.getvalue(), foreign_prefixes=('oracle',))
    assert info['used'] == 'ema' and info['epoch'] == 3 and model.names == ema.names and not model.training
    want_model = copy.deepcopy(ema).half().float().eval()
    x = torch.rand(2, 3, 96, 96, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        rel_close(model(x.cuda())[0], want_model(x)[0], what='z from the loaded checkpoint')


def test_hipgraph_replay_matches_eager():
    """somi_amd.graph.GraphedModel: one captured inference forward replayed as a hipGraph gives bit-identical predictions, also
    for new inputs copied into its static buffer."""
    from oracle.somi_ref.testing import SOMI_ANCHORS
    from somi_amd.graph import GraphedModel
    _, mine = build(0.25, 0.33, SOMI_ANCHORS, seed=4)
    g = torch.Generator().manual_seed(2)
    x0 = torch.randint(0, 256, (2, 3, 96, 96), generator=g, dtype=torch.uint8).cuda()
    x1 = torch.randint(0, 256, (2, 3, 96, 96), generator=g, dtype=torch.uint8).cuda()
    fast = GraphedModel(mine, x0)
    with torch.no_grad():
        for x in (x0, x1, x0):
            z, raws = fast(x)
            ze, rawe = mine(x)
            assert torch.equal(z, ze) and all(torch.equal(a, b) for a, b in zip(raws, rawe))
    with pytest.raises(RuntimeError, match='captured for'):
        fast(x0[:1])


@pytest.mark.parametrize('nc', [1, 80])
def test_other_class_counts(nc):
    """single-class (no = 6; the reference drops the class loss, utils/loss.py:182) and COCO-sized heads (no = 85): eval
    predictions, training loss and gradients against the oracle."""
    from oracle.somi_ref import Model as OModel
    from oracle.somi_ref.loss import ComputeLoss as OLoss
    from oracle.somi_ref.testing import SOMI_ANCHORS, fill_state, somi_cfg, synthetic_batch, HYP_VISDRONE
    from somi_amd.loss import ComputeLoss
    from somi_amd.model import Model
    cfg = somi_cfg(0.25, 0.33, nc=nc, anchors=SOMI_ANCHORS)
    ref = fill_state(OModel(cfg), 9)
    mine = Model(cfg)
    mine.load_state_dict(ref.state_dict())
    ref.hyp = mine.hyp = dict(HYP_VISDRONE)
    imgs, targets = synthetic_batch(2, 64, nc=nc, seed=2)
    mine = mine.cuda().eval()
    ref.eval()
    with torch.no_grad():
        rel_close(mine(imgs.cuda())[0], ref(imgs.float() / 255)[0], what=f'z nc={nc}')
    ref.train()
    mine.train()
    lr, ir = OLoss(ref)(ref(imgs.float() / 255), targets)
    lr.backward()
    lm, im = ComputeLoss(mine)(mine(imgs.cuda()), targets.cuda())
    rel_close(lm, lr.detach(), rel=1e-4, what='loss')
    rel_close(im, ir, rel=1e-4, what='loss items')
    lm.backward()
    for (n, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        if q.grad is None:
            continue
        err = (p.grad.cpu().double() - q.grad.double()).abs().max().item()
        scale = q.grad.double().abs().max().item()
        assert err <= 2e-3 * scale + 5e-6, (n, err, scale)      # atol: ODConv's squeeze path under a batch-of-2 BN is rounding noise


def test_attempt_load_checkpoint_written_by_the_reference(golden, tmp_path):
    """attempt_load (models/experimental.py:90-122) on the bytes of a checkpoint pickled from the reference's own module classes
    (tests/golden/checkpoint_ref.npz, default foreign_prefixes): EMA weights first, fp16 -> float, eval; its predictions equal what
    the reference's own EMA module computes for the same image."""
    from somi_amd.checkpoint import attempt_load
    g = golden('checkpoint_ref')
    f = tmp_path / 'best.pt'
    f.write_bytes(g['bytes'].tobytes())
    model, info = attempt_load(str(f))
    assert info['used'] == 'ema' and info['epoch'] == 12 and info['updates'] == 345
    assert sum(p.numel() for p in model.parameters()) == int(g['nparams_ema']) and not model.training
    assert model.names[0] == 'class0' and model.hyp['box'] == 0.07
    with torch.no_grad():
        z, _ = model(T(g['x']).cuda())
    rel_close(z, T(g['z']), what='predictions of the loaded EMA weights')


def test_tta_forward_augment_matches_reference_vectors(golden):
    """`model(x, augment=True)` (models/yolo.py:1253-1318; `val.py --augment`): three scales x two flips through one resample kernel each,
    de-scaled, clipped, concatenated - against the reference's own output for the SOMI graph (4 levels) and yolov5 (3 levels)."""
    from oracle.somi_ref import Model as OModel
    from oracle.somi_ref.testing import SOMI_ANCHORS, fill_state, somi_cfg, yolov5_cfg
    from somi_amd.model import Model
    g = golden('model_tta')
    x = T(g['x']).cuda()
    for key, cfg in (('z_somi', somi_cfg(0.25, 0.33, anchors=SOMI_ANCHORS)), ('z_yolov5', yolov5_cfg(0.25, 0.33, nc=3))):
        mine = Model(cfg)
        mine.load_state_dict(fill_state(OModel(cfg), 1).state_dict())
        mine = mine.cuda().eval()
        with torch.no_grad():
            z, none = mine(x, augment=True)
            z8, _ = mine((x * 255).round().to(torch.uint8), augment=True)        # the uint8 batch contract goes through the same path
        assert none is None and z.shape == T(g[key]).shape and z8.shape == z.shape
        rel_close(z, T(g[key]), what=f'TTA {key}')
