"""Input pipeline on the device: `somi_augment_u8` through the C ABI against the fixture produced by the
reference's own `LoadImagesAndLabels.__getitem__` (bit-exact pixels and labels), against the oracle at the full 640 px
size, and through size-independent properties."""
import random

import numpy as np
import pytest
import torch

from oracle.somi_ref.augment import CachedDataset, collate
from oracle.somi_ref.testing import HYP_AUGMENT, synthetic_image_set
from tests.test_augment import fixture_cases

pytestmark = pytest.mark.gpu


def seed_all(seed):
    random.seed(seed), np.random.seed(seed)


def test_device_samples_match_the_reference_fixture(golden):
    from somi_amd.augment import DeviceImageCache
    g = golden('augment')
    for k, name, seed, idx, imgs, labs, S, hyp, augment in fixture_cases(g):
        ds = DeviceImageCache(imgs, labs, S, hyp, augment=augment)
        seed_all(seed)
        img, lab, _, shapes = ds[idx]
        assert img.is_cuda and img.dtype == torch.uint8 and tuple(img.shape) == (3, S, S)
        bad = int((img.cpu().numpy() != g[f'out_img{k}']).sum())
        assert bad == 0, f'{name} seed {seed}: {bad} bytes differ'
        assert np.array_equal(lab.numpy(), g[f'out_lab{k}']), (name, seed)


@pytest.mark.parametrize('over', [dict(), dict(mixup=1.0), dict(degrees=10.0, translate=0.1, shear=5.0, flipud=0.5),
                                  dict(mosaic=0.0)])
def test_full_size_batch_matches_the_oracle(over):
    """640 px, a batch of 8 from one launch, against the oracle sample by sample (same seeds -> same random draws)."""
    from somi_amd.augment import DeviceImageCache
    S = 640
    imgs, labs = synthetic_image_set(S, n=8, seed=11)
    hyp = dict(HYP_AUGMENT, **over)
    picks = [5, 0, 7, 2, 2, 1, 6, 3]
    seed_all(123)
    want_img, want_lab, _ = collate([CachedDataset(imgs, labs, S, hyp)[i] for i in picks])
    ds = DeviceImageCache(imgs, labs, S, hyp)
    seed_all(123)
    got_img, got_lab, _, _ = ds.batch(picks)
    assert tuple(got_img.shape) == (8, 3, S, S)
    assert torch.equal(got_img.cpu(), want_img)
    assert torch.equal(got_lab, want_lab)


def test_validation_path_is_a_letterboxed_copy():
    """augment=False (val.py's loader): no random draw, the image centred on grey 114, BGR -> RGB planes."""
    from somi_amd.augment import DeviceImageCache
    S = 96
    imgs, labs = synthetic_image_set(S, n=4, seed=3)
    ds = DeviceImageCache(imgs, labs, S, augment=False)
    state = random.getstate()
    out, targets, paths, shapes = ds.batch([0, 1, 2, 3])
    assert paths == (0, 1, 2, 3)
    assert random.getstate() == state
    for j, im in enumerate(imgs):
        h, w = im.shape[:2]
        dw, dh = shapes[j][1][1]
        top, left = int(round(dh - 0.1)), int(round(dw - 0.1))
        want = np.full((S, S, 3), 114, np.uint8)
        want[top:top + h, left:left + w] = im
        assert np.array_equal(out[j].cpu().numpy(), want.transpose(2, 0, 1)[::-1])
        rows = targets[targets[:, 0] == j]
        assert rows.shape[0] == labs[j].shape[0]


def test_rect_validation_batches_match_the_reference_fixture(golden):
    """val.py's loader: rect=True, pad=0.5, augment=False -> non-square (B,3,h,w) batches, one launch per batch."""
    from somi_amd.augment import DeviceImageCache
    g = golden('augment_rect')
    n, S, bs = int(g['n']), int(g['img_size']), int(g['batch_size'])
    imgs, labs = [g[f'src{i}'] for i in range(n)], [g[f'lab{i}'] for i in range(n)]
    ds = DeviceImageCache(imgs, labs, S, dict(HYP_AUGMENT), augment=False, rect=True, batch_size=bs, stride=int(g['stride']),
                          pad=float(g['pad']))
    for b0 in range(0, n, bs):
        picks = list(range(b0, min(b0 + bs, n)))
        out, targets, paths, shapes = ds.batch(picks)
        assert [int(g['order'][k]) for k in picks] == list(paths)
        assert tuple(out.shape[2:]) == tuple(g['batch_shapes'][b0 // bs])
        for j, k in enumerate(picks):
            assert np.array_equal(out[j].cpu().numpy(), g[f'out_img{k}']), k
            assert np.array_equal(targets[targets[:, 0] == j][:, 1:].numpy(), g[f'out_lab{k}'][:, 1:])


def test_flips_and_identity_tables_are_pure_moves():
    """Properties at full size: with HSV gains 0 the colour path is skipped, so fliplr/flipud forced on vs off differ by
    exactly a flip; and jitter tables that are the identity change at most the rounding of the HSV round trip."""
    from somi_amd.augment import DeviceImageCache
    S = 640
    imgs, labs = synthetic_image_set(S, n=5, seed=5)
    base = dict(HYP_AUGMENT, hsv_h=0.0, hsv_s=0.0, hsv_v=0.0, mixup=0.0, fliplr=0.0, flipud=0.0)
    outs = {}
    for tag, over in (('none', {}), ('lr', dict(fliplr=1.0)), ('ud', dict(flipud=1.0)), ('both', dict(fliplr=1.0, flipud=1.0))):
        ds = DeviceImageCache(imgs, labs, S, dict(base, **over))
        seed_all(9)
        outs[tag], lab, _, _ = ds.batch([0, 1, 2, 3])
        if tag == 'none':
            lab0 = lab
        elif tag == 'both':
            assert torch.allclose(lab[:, 2:4], 1 - lab0[:, 2:4], atol=1e-6) and torch.equal(lab[:, 4:], lab0[:, 4:])
    assert torch.equal(outs['lr'], outs['none'].flip(3))
    assert torch.equal(outs['ud'], outs['none'].flip(2))
    assert torch.equal(outs['both'], outs['none'].flip(2).flip(3))
    # grey fill survives everything: the border of a mosaic crop is exactly 114 wherever the canvas was empty
    assert int((outs['none'] == 114).sum()) > 0


def test_records_are_validated_before_launch():
    from somi_amd.augment import DeviceImageCache
    imgs, labs = synthetic_image_set(32, n=2, seed=1)
    ds = DeviceImageCache(imgs, labs, 32)
    seed_all(0)
    plan, _, _ = ds.plan(0)
    idx, x1, y1, x2, y2, dx, dy = plan.canvases[0].sources[0]
    plan.canvases[0].sources[0] = (idx, x1, y1, x2 + 64, y2, dx, dy)        # rectangle wider than its source
    with pytest.raises(RuntimeError, match='outside its source'):
        ds.render([plan])


def test_device_batches_drive_the_training_step():
    """The loader's output is what `train.py:247-265` consumes: uint8 (B,3,s,s) on the device + (nt,6) targets.  Two steps fed by
    the device pipeline against the CPU oracle (model, loss, Adam) fed by the oracle pipeline with the same seeds."""
    from oracle.somi_ref.loss import ComputeLoss as OLoss
    from oracle.somi_ref.model import Model as OModel
    from oracle.somi_ref.testing import fill_state, HYP_VISDRONE
    from somi_amd.augment import DeviceImageCache
    from somi_amd.model import Model
    from somi_amd.optim import reference_param_groups
    from somi_amd.train import TrainStep
    from tests.test_train_gpu import _train_cfg
    S, B = 64, 2
    imgs, labs = synthetic_image_set(S, n=6, seed=21)
    cfg = _train_cfg(False)
    ref = fill_state(OModel(cfg), 4)
    mine = Model(cfg)
    mine.load_state_dict(ref.state_dict())
    hyp = dict(HYP_VISDRONE)
    ref.hyp = hyp
    ref.train()
    g0, g1, g2 = reference_param_groups(ref)
    opt = torch.optim.Adam(g0, lr=3e-4, betas=(hyp['momentum'], 0.999))
    opt.add_param_group({'params': g1, 'weight_decay': hyp['weight_decay'] * 2 * 32 / 64})
    opt.add_param_group({'params': g2})
    crit = OLoss(ref)
    tr = TrainStep(mine.cuda(), hyp, B)
    cpu_ds, dev_ds = CachedDataset(imgs, labs, S, dict(HYP_AUGMENT)), DeviceImageCache(imgs, labs, S, dict(HYP_AUGMENT))
    for step, picks in enumerate(([0, 3], [5, 2])):
        seed_all(50 + step)
        ci, ct, _ = collate([cpu_ds[i] for i in picks])
        seed_all(50 + step)
        di, dt, _, _ = dev_ds.batch(picks)
        assert torch.equal(di.cpu(), ci) and torch.equal(dt, ct)
        lr_, _ = crit(ref(ci.float() / 255), ct)
        opt.zero_grad()
        lr_.backward()
        opt.step()
        lm, _ = tr.step(di, dt.cuda())
        tol = (1e-4, 1e-2)[step]
        want = float(lr_.detach())
        assert abs(float(lm) - want) <= tol * abs(want), (step, float(lm), want)


def test_rect_batches_feed_the_eval_forward(golden):
    """val.py:148-157: a rect batch from the device loader through the eval forward, against the oracle model on the oracle
    loader's batch (non-square 96x128 input, batch 3)."""
    from oracle.somi_ref.testing import SOMI_ANCHORS
    from somi_amd.augment import DeviceImageCache
    from tests.test_model_gpu import build, rel_close
    g = golden('augment_rect')
    n, S, bs = int(g['n']), int(g['img_size']), int(g['batch_size'])
    imgs, labs = [g[f'src{i}'] for i in range(n)], [g[f'lab{i}'] for i in range(n)]
    kw = dict(augment=False, rect=True, batch_size=bs, stride=int(g['stride']), pad=float(g['pad']))
    ref, mine = build(0.25, 0.33, SOMI_ANCHORS, seed=6)
    want_img, _, _ = collate([CachedDataset(imgs, labs, S, dict(HYP_AUGMENT), **kw)[k] for k in range(bs)])
    got_img, _, _, _ = DeviceImageCache(imgs, labs, S, dict(HYP_AUGMENT), **kw).batch(range(bs))
    assert torch.equal(got_img.cpu(), want_img) and got_img.shape[2] != got_img.shape[3]
    with torch.no_grad():
        rel_close(mine(got_img)[0], ref(want_img.float() / 255)[0], what='z on a rect batch')


@pytest.mark.parametrize('S,over', [(62, dict(mixup=0.5)),                                        # width % 4 != 0: the 1-pixel kernel
                                    (100, dict(degrees=45.0, scale=0.9, shear=20.0, translate=0.3, flipud=0.5, mixup=0.3)),
                                    (64, dict(mosaic=0.5, degrees=30.0, scale=0.5))])
def test_many_random_samples_match_the_oracle(S, over):
    """Breadth: 24 seeds per setting with extreme zoom / rotation / shear (crops that leave the canvas, footprints that straddle
    mosaic seams and image borders), odd sizes, images smaller than the canvas quadrant - device vs oracle, bit for bit."""
    from somi_amd.augment import DeviceImageCache
    imgs, labs = synthetic_image_set(S, n=8, seed=S)
    if over.get('mosaic', 1.0) == 1.0:                      # (a single-image sample would need cv2.resize to scale it up)
        imgs[2] = np.ascontiguousarray(imgs[2][: max(4, S // 5), : max(4, S // 3)])      # a tiny image inside the mosaics
    hyp = dict(HYP_AUGMENT, **over)
    cpu, dev = CachedDataset(imgs, labs, S, hyp), DeviceImageCache(imgs, labs, S, hyp)
    for seed in range(24):
        picks = [(seed + k) % 8 for k in range(4)]
        seed_all(1000 + seed)
        want_img, want_lab, _ = collate([cpu[i] for i in picks])
        seed_all(1000 + seed)
        got_img, got_lab, _, _ = dev.batch(picks)
        bad = int((got_img.cpu() != want_img).sum())
        assert bad == 0, f'S={S} seed {seed}: {bad} bytes differ'
        assert torch.equal(got_lab, want_lab), (S, seed)


def test_training_through_the_device_loader_learns_the_task():
    """tests/e2e_loader.py: a detector trained ONLY on the loader's augmented samples (mosaic, affine crop, mixup, jitter,
    flips) finds the objects in plain rectangular validation batches - boxes and pixels stay aligned through the pipeline."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('e2e_loader', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'e2e_loader.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.main()
    assert res['loss_last'] < 0.5 * res['loss_first'], res
    assert res['val']['mAP50'] > 0.5, res          # 6 runs measured 0.86 - 0.95 (training uses fp atomics: run-to-run jitter)


def test_e2e_loader_run_is_reproducible():
    """Round 1 saw mAP@0.5 between 0.70 and 0.94 (and once 0.20) over repeated runs of the test above and could not tell training
    chaos from a race in a backward kernel.  With every reduction on the step in a fixed order the run is a pure function of its seeds:
    two runs (device loader -> train steps -> validation) must agree to the last bit - first loss, last loss, P, R, mAP."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('e2e_loader', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'e2e_loader.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    a, b = mod.main(steps=60), mod.main(steps=60)
    for k in ('loss_first', 'loss_last'):
        assert a[k] == b[k], (k, a[k], b[k])
    assert a['val'] == b['val'], (a['val'], b['val'])
