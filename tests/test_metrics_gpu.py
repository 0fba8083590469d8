"""Validation metrics on the MI355X (somi_amd.metrics over somi_val_match_f32 / somi_ap_per_class_f64) against the outputs of the
reference's own process_batch (val.py:50-71) and ap_per_class (utils/metrics.py:21-74) in tests/golden/val_metrics.npz, and
against the CPU oracle on larger seeded cases.  Bars: the correct-matrix is bit-exact; AP / P / R / F1 are fp64, 1e-9."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
T = torch.from_numpy
DEV = 'cuda:0'


def test_process_batch_matches_reference_vectors(golden):
    from somi_amd.metrics import process_batch, process_batches
    g = golden('val_metrics')
    iouv = T(g['iouv']).to(DEV)
    n = int(g['nimg'])
    dets = [T(g[f'det{b}']).to(DEV) for b in range(n)]
    labs = [T(g[f'lab{b}']).to(DEV) for b in range(n)]
    for b, got in enumerate(process_batches(dets, labs, iouv)):          # one launch for the whole list
        assert np.array_equal(got.cpu().numpy(), g[f'correct{b}']), f'image {b}'
    for b in (0, 3, 5, 7):                                               # the single-image form val.py:184 calls (3: no labels, 5: no detections)
        assert np.array_equal(process_batch(dets[b], labs[b], iouv).cpu().numpy(), g[f'correct{b}'])


def test_ap_per_class_matches_reference_vectors(golden):
    from somi_amd.metrics import ap_per_class
    g = golden('val_metrics')
    p, r, ap, f1, cls_ = ap_per_class(T(g['tp']).to(DEV), T(g['conf']).to(DEV), T(g['pred_cls']).to(DEV), T(g['target_cls']).to(DEV))
    assert np.array_equal(cls_.cpu().numpy(), g['ap_class'])
    for a, b, what in ((p, g['p'], 'p'), (r, g['r'], 'r'), (ap, g['ap'], 'ap'), (f1, g['f1'], 'f1')):
        np.testing.assert_allclose(a.cpu().numpy(), b, rtol=1e-9, atol=1e-12, err_msg=what)


def _scene(g, nimg, nc, max_l, max_d):
    dets, labs = [], []
    for _ in range(nimg):
        M = int(torch.randint(0, max_l, (1,), generator=g))
        N = int(torch.randint(0, max_d, (1,), generator=g))
        lc, lwh = torch.rand(M, 2, generator=g) * 600 + 20, torch.rand(M, 2, generator=g) * 90 + 6
        lab = torch.cat((torch.randint(0, nc, (M, 1), generator=g).float(), lc - lwh / 2, lc + lwh / 2), 1)
        if M and N:
            pick = torch.randint(0, M, (N,), generator=g)
            box = lab[pick, 1:] + (torch.rand(N, 4, generator=g) - 0.5) * lwh[pick].repeat(1, 2) * torch.rand(N, 1, generator=g)
            cls = torch.where(torch.rand(N, generator=g) < 0.2, torch.randint(0, nc + 2, (N,), generator=g).float(), lab[pick, 0])
        else:
            c, w = torch.rand(N, 2, generator=g) * 600 + 20, torch.rand(N, 2, generator=g) * 90 + 6
            box, cls = torch.cat((c - w / 2, c + w / 2), 1), torch.randint(0, nc, (N,), generator=g).float()
        dets.append(torch.cat((box, torch.rand(N, 1, generator=g), cls[:, None]), 1))
        labs.append(lab)
    return dets, labs


@pytest.mark.parametrize('nimg,nc,max_l,max_d', [(64, 10, 150, 301), (9, 3, 700, 300), (5, 80, 40, 60)])
def test_metrics_match_oracle(nimg, nc, max_l, max_d):
    """VisDrone-like density (up to 150 labels, 300 detections per image), a crowded 3-class set (700 labels: the large-LDS
    path of the matcher) and an 80-class set with classes that never appear among the targets or the predictions."""
    from oracle.somi_ref.metrics import ap_per_class as o_ap, process_batch as o_pb
    from somi_amd.metrics import ap_per_class, process_batches
    g = torch.Generator().manual_seed(nimg * 7 + nc)
    dets, labs = _scene(g, nimg, nc, max_l, max_d)
    iouv = torch.linspace(0.5, 0.95, 10)
    want = [o_pb(d, l, iouv) if d.shape[0] and l.shape[0] else torch.zeros(d.shape[0], 10, dtype=torch.bool) for d, l in zip(dets, labs)]
    got = process_batches([d.to(DEV) for d in dets], [l.to(DEV) for l in labs], iouv.to(DEV))
    for b, (a, w) in enumerate(zip(got, want)):
        assert torch.equal(a.cpu(), w), f'image {b}'
    tp = torch.cat(want).numpy()
    conf, pcls = torch.cat([d[:, 4] for d in dets]).numpy(), torch.cat([d[:, 5] for d in dets]).numpy()
    tcls = torch.cat([l[:, 0] for l in labs]).numpy()
    wp, wr, wap, wf1, wcls = o_ap(tp, conf, pcls, tcls)
    p, r, ap, f1, cls_ = ap_per_class(torch.cat(got), T(conf).to(DEV), T(pcls).to(DEV), T(tcls).to(DEV))
    assert np.array_equal(cls_.cpu().numpy(), wcls)
    for a, b, what in ((p, wp, 'p'), (r, wr, 'r'), (ap, wap, 'ap'), (f1, wf1, 'f1')):
        np.testing.assert_allclose(a.cpu().numpy(), b, rtol=1e-9, atol=1e-12, err_msg=what)


def test_metrics_empty_inputs():
    from somi_amd.metrics import ap_per_class, process_batches
    iouv = torch.linspace(0.5, 0.95, 10, device=DEV)
    out = process_batches([torch.zeros(0, 6, device=DEV)], [torch.zeros(0, 5, device=DEV)], iouv)
    assert out[0].shape == (0, 10)
    p, r, ap, f1, cls_ = ap_per_class(torch.zeros(0, 10, dtype=torch.bool, device=DEV), torch.zeros(0, device=DEV), torch.zeros(0, device=DEV),
                                      torch.tensor([2., 2., 5.], device=DEV))
    assert cls_.tolist() == [2, 5] and float(ap.abs().max()) == 0.0 and float(p.abs().max()) == 0.0


def test_scale_coords_matches_reference_formula():
    """val.scale_coords (utils/general.py:602-628): letterbox undo + clipping, with and without an explicit ratio / pad."""
    from somi_amd.val import scale_coords
    g = torch.Generator().manual_seed(3)
    boxes = torch.rand(50, 4, generator=g) * 700 - 30
    for img1, img0, rp in (((640, 640), (480, 720), None), ((384, 640), (1080, 1920), None), ((640, 640), (300, 500), ((1.28, 1.28), (0.0, 128.0)))):
        want = boxes.clone().double()
        if rp is None:
            gain = min(img1[0] / img0[0], img1[1] / img0[1])
            pad = ((img1[1] - img0[1] * gain) / 2, (img1[0] - img0[0] * gain) / 2)
        else:
            gain, pad = rp[0][0], rp[1]
        want[:, [0, 2]] = ((want[:, [0, 2]] - pad[0]) / gain).clamp(0, img0[1])
        want[:, [1, 3]] = ((want[:, [1, 3]] - pad[1]) / gain).clamp(0, img0[0])
        got = scale_coords(img1, boxes.clone().to(DEV), img0, rp)
        np.testing.assert_allclose(got.cpu().numpy(), want.float().numpy(), rtol=1e-6, atol=1e-4)


def test_confusion_matrix_matches_the_reference_and_the_oracle(golden):
    """val.py:141,186: the device confusion matrix over the golden images equals the reference class's matrix exactly, image by
    image and as one batch; plus random images (many labels / detections, empty cases) against the oracle."""
    from oracle.somi_ref.metrics import ConfusionMatrix as OCM
    from somi_amd.metrics import ConfusionMatrix
    g = golden('val_metrics')
    nc, conf, thr = int(g['confusion_nc']), float(g['confusion_conf']), float(g['confusion_iou'])
    dets = [torch.from_numpy(g[f'det{b}']).cuda() for b in range(int(g['nimg']))]
    labs = [torch.from_numpy(g[f'lab{b}']).cuda() for b in range(int(g['nimg']))]
    keep = [b for b in range(len(dets)) if len(dets[b]) and len(labs[b])]
    one = ConfusionMatrix(nc, conf, thr)
    for b in keep:
        one.process_batch(dets[b], labs[b])
    assert np.array_equal(one.matrix, g['confusion'])
    batch = ConfusionMatrix(nc, conf, thr)
    batch.process_batches([dets[b] for b in keep], [labs[b] for b in keep])
    assert np.array_equal(batch.matrix, g['confusion'])
    gen = torch.Generator().manual_seed(77)
    ocm, dcm = OCM(12), ConfusionMatrix(12)
    dl, ll = [], []
    for i in range(40):
        M = int(torch.randint(0, 120, (1,), generator=gen)) if i % 7 else 0
        N = int(torch.randint(0, 300, (1,), generator=gen)) if i % 5 else 0
        c, wh = torch.rand(M, 2, generator=gen) * 600 + 20, torch.rand(M, 2, generator=gen) * 90 + 5
        lab = torch.cat((torch.randint(0, 12, (M, 1), generator=gen).float(), c - wh / 2, c + wh / 2), 1)
        if M and N:
            pick = torch.randint(0, M, (N,), generator=gen)
            box = lab[pick, 1:] + (torch.rand(N, 4, generator=gen) - 0.5) * wh[pick].repeat(1, 2) * torch.rand(N, 1, generator=gen) * 1.5
            cls = torch.where(torch.rand(N, generator=gen) < 0.7, lab[pick, 0], torch.randint(0, 12, (N,), generator=gen).float())
        else:
            cc, cw = torch.rand(N, 2, generator=gen) * 600 + 20, torch.rand(N, 2, generator=gen) * 90 + 5
            box, cls = torch.cat((cc - cw / 2, cc + cw / 2), 1), torch.randint(0, 12, (N,), generator=gen).float()
        det = torch.cat((box, torch.rand(N, 1, generator=gen), cls[:, None]), 1)
        ocm.process_batch(det, lab)
        dl.append(det.cuda()), ll.append(lab.cuda())
    dcm.process_batches(dl, ll)
    assert np.array_equal(dcm.matrix, ocm.matrix) and ocm.matrix.sum() > 2000
