"""Input pipeline, CPU side: the oracle against the fixture produced by the reference's own
`LoadImagesAndLabels.__getitem__`, known answers for the restated OpenCV arithmetic, and the product's host logic
(random draw order, mosaic geometry, label boxes) against the same fixture - no device work here."""
import random

import numpy as np
import pytest
import torch

from oracle.somi_ref import cv_port
from oracle.somi_ref.augment import CachedDataset
from oracle.somi_ref.testing import HYP_AUGMENT

CASE_HYP = {'mosaic': {}, 'mixup': dict(mixup=1.0),
            'general': dict(degrees=10.0, translate=0.1, shear=5.0, flipud=0.5, mixup=0.5), 'single': dict(mosaic=0.0), 'val': {}}


def fixture_cases(g):
    n, S = int(g['n']), int(g['img_size'])
    imgs, labs = [g[f'src{i}'] for i in range(n)], [g[f'lab{i}'] for i in range(n)]
    for k, (name, seed, idx) in enumerate(zip(g['case'], g['seed'], g['index'])):
        name = str(name)
        yield k, name, int(seed), int(idx), imgs, labs, S, dict(HYP_AUGMENT, **CASE_HYP[name]), name != 'val'


def test_oracle_reproduces_the_reference_samples(golden):
    g = golden('augment')
    seen = set()
    for k, name, seed, idx, imgs, labs, S, hyp, augment in fixture_cases(g):
        ds = CachedDataset(imgs, labs, S, hyp, augment=augment)
        random.seed(seed), np.random.seed(seed)
        img, lab, shapes = ds[idx]
        assert np.array_equal(img.numpy(), g[f'out_img{k}']), (name, seed)
        assert np.array_equal(lab.numpy(), g[f'out_lab{k}']), (name, seed)
        if shapes is not None:
            assert tuple(shapes[1][1]) == tuple(g[f'out_pad{k}'])
        seen.add(name)
    assert seen == set(CASE_HYP)


def test_opencv_restatement_known_answers():
    """Published OpenCV behaviour: hue of the primaries on the 0..180 scale, exact HSV round trip of saturated colours, identity
    and integer-shift warps are copies, a half-pixel shift is the rounded mean of neighbours, 90 degree rotation matrix."""
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [0, 0, 0], [128, 128, 128], [0, 255, 255],
                    [255, 255, 0], [255, 0, 255], [10, 20, 30]]], dtype=np.uint8)
    hsv = cv_port.cvtColor(px, cv_port.COLOR_BGR2HSV)
    assert hsv[0].tolist() == [[120, 255, 255], [60, 255, 255], [0, 255, 255], [0, 0, 255], [0, 0, 0], [0, 0, 128],
                               [30, 255, 255], [90, 255, 255], [150, 255, 255], [15, 170, 30]]
    assert np.array_equal(cv_port.cvtColor(hsv, cv_port.COLOR_HSV2BGR), px)
    im = np.random.RandomState(0).randint(0, 256, (20, 30, 3)).astype(np.uint8)
    eye = np.array([[1, 0, 0], [0, 1, 0]], float)
    assert np.array_equal(cv_port.warpAffine(im, eye, (30, 20), borderValue=(114,) * 3), im)
    sh = cv_port.warpAffine(im, np.array([[1, 0, 2], [0, 1, 3]], float), (30, 20), borderValue=(114,) * 3)
    assert np.array_equal(sh[3:, 2:], im[:-3, :-2]) and (sh[:3] == 114).all() and (sh[:, :2] == 114).all()
    half = cv_port.warpAffine(im, np.array([[1, 0, 0.5], [0, 1, 0]], float), (30, 20), borderValue=(114,) * 3)
    assert np.array_equal(half[:, 1:], (im[:, :-1].astype(int) + im[:, 1:] + 1) >> 1)
    assert np.allclose(cv_port.getRotationMatrix2D((0, 0), 90, 1.0), [[0, 1, 0], [-1, 0, 0]], atol=1e-15)


def test_host_plan_matches_the_reference_labels_and_geometry(golden):
    """The product's host side (no kernel involved): same labels as the reference's __getitem__ for the same seeds, and
    every mosaic rectangle it would hand to the kernel lies inside its source image."""
    from somi_amd.augment import DeviceImageCache
    g = golden('augment')
    for k, name, seed, idx, imgs, labs, S, hyp, augment in fixture_cases(g):
        ds = DeviceImageCache(imgs, labs, S, hyp, augment=augment, device='cpu')
        random.seed(seed), np.random.seed(seed)
        plan, labels, shapes = ds.plan(idx)
        canvases, mix_r, luts, flipud, fliplr, out_hw = plan
        assert out_hw == (S, S)
        want = g[f'out_lab{k}']
        assert labels.shape[0] == want.shape[0], (name, seed)
        assert np.array_equal(labels.astype(np.float32), want[:, 1:]), (name, seed)
        assert (mix_r is not None) == (len(canvases) == 2)
        for c in canvases:
            for (i, x1, y1, x2, y2, dx, dy) in c.sources:
                h, w = imgs[i].shape[:2]
                assert 0 <= x1 - dx and x2 - dx <= w and 0 <= y1 - dy and y2 - dy <= h
                assert 0 <= x1 <= x2 <= c.width and 0 <= y1 <= y2 <= c.height
        if shapes is not None:
            assert tuple(shapes[1][1]) == tuple(g[f'out_pad{k}'])
        with pytest.raises(RuntimeError, match='no CPU fallback'):
            ds.render([plan])


def test_rect_batches_follow_the_reference(golden):
    """val.py's loader (rect=True, pad=0.5): aspect-ratio order and batch shapes from the reference's own source text, samples
    from its __getitem__ - against the oracle (pixels + labels) and the product's host plan (order, shapes, labels, placement)."""
    from somi_amd.augment import DeviceImageCache
    g = golden('augment_rect')
    n, S, bs = int(g['n']), int(g['img_size']), int(g['batch_size'])
    imgs, labs = [g[f'src{i}'] for i in range(n)], [g[f'lab{i}'] for i in range(n)]
    kw = dict(augment=False, rect=True, batch_size=bs, stride=int(g['stride']), pad=float(g['pad']))
    ora = CachedDataset(imgs, labs, S, dict(HYP_AUGMENT), **kw)
    dev = DeviceImageCache(imgs, labs, S, dict(HYP_AUGMENT), device='cpu', **kw)
    assert np.array_equal(ora.order, g['order']) and np.array_equal(ora.batch_shapes, g['batch_shapes'])
    assert np.array_equal(dev.order, g['order']) and np.array_equal(dev.batch_shapes, g['batch_shapes'])
    for k in range(n):
        img, lab, shapes = ora[k]
        assert np.array_equal(img.numpy(), g[f'out_img{k}']) and np.array_equal(lab.numpy(), g[f'out_lab{k}'])
        assert tuple(shapes[1][1]) == tuple(g[f'out_pad{k}'])
        plan, labels, shapes = dev.plan(k)
        assert plan.out_hw == tuple(g[f'out_img{k}'].shape[1:]) and plan.mix_r is None and plan.luts is None
        assert np.array_equal(labels, g[f'out_lab{k}'][:, 1:]) and tuple(shapes[1][1]) == tuple(g[f'out_pad{k}'])
    with pytest.raises(RuntimeError, match='different letterbox shapes'):
        DeviceImageCache.render(dev, [dev.plan(0)[0], dev.plan(n - 1)[0]])


def test_batch_collates_like_the_reference(golden):
    from somi_amd.augment import DeviceImageCache
    from oracle.somi_ref.augment import collate
    g = golden('augment')
    n, S = int(g['n']), int(g['img_size'])
    imgs, labs = [g[f'src{i}'] for i in range(n)], [g[f'lab{i}'] for i in range(n)]
    random.seed(7), np.random.seed(7)
    ref = collate([CachedDataset(imgs, labs, S, dict(HYP_AUGMENT))[i] for i in (3, 0, 5, 1)])
    ds = DeviceImageCache(imgs, labs, S, dict(HYP_AUGMENT), device='cpu')
    random.seed(7), np.random.seed(7)
    blocks = []
    for j, i in enumerate((3, 0, 5, 1)):
        _, labels, _ = ds.plan(i)
        b = torch.zeros((len(labels), 6))
        b[:, 1:] = torch.from_numpy(labels)
        b[:, 0] = j
        blocks.append(b)
    assert torch.equal(torch.cat(blocks, 0), ref[1])
    with pytest.raises(RuntimeError, match='no CPU fallback'):          # the public entry point renders on the device only
        ds.batch([3, 0])


def test_unbuilt_options_raise():
    from somi_amd.augment import DeviceImageCache
    im = [np.zeros((8, 8, 3), np.uint8)]
    lab = [np.zeros((0, 5), np.float32)]
    DeviceImageCache(im, lab, 8, dict(HYP_AUGMENT, copy_paste=0.5), device='cpu')      # box labels only: copy_paste has nothing to paste
    with pytest.raises(NotImplementedError):
        DeviceImageCache(im, lab, 8, dict(HYP_AUGMENT, perspective=0.001), device='cpu')
    with pytest.raises(ValueError):
        DeviceImageCache([np.zeros((16, 8, 3), np.uint8)], lab, 8, device='cpu')
