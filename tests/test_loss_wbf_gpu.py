"""Loss and WBF kernels on the MI355X against reference-generated vectors (loss) and the CPU oracle (WBF)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def rel_close(got, want, rel=1e-3, what=''):
    got, want = torch.as_tensor(got).detach().cpu().double(), torch.as_tensor(want).detach().cpu().double()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = (got - want).abs().max().item()
    scale = want.abs().max().item() + 1e-12
    assert err <= rel * scale, f'{what}: max err {err:.3e} vs scale {scale:.3e}'


class _M:
    def __init__(self, anchors, hyp, nc=10):
        class Det:
            pass
        d = Det()
        d.anchors, d.nl, d.na, d.nc = anchors, anchors.shape[0], anchors.shape[1], nc
        self.model, self.hyp = [d], hyp


@pytest.mark.parametrize('tag', ['a', 'b', 'empty', 'edge'])
def test_compute_loss_matches_reference_vectors(golden, tag):
    from somi_amd.configs import HYP_VISDRONE
    from somi_amd.loss import ComputeLoss
    g = golden('loss_' + tag)
    crit = ComputeLoss(_M(T(g['anchors']), dict(HYP_VISDRONE)))
    p = [T(g[f'p{i}']).cuda().requires_grad_(True) for i in range(4)]
    loss, items = crit(p, T(g['targets']).cuda())
    rel_close(loss, g['loss'], rel=1e-4, what='loss')
    rel_close(items, g['items'], rel=1e-4, what='loss_items')
    loss.backward()
    for i in range(4):
        rel_close(p[i].grad, g[f'g{i}'], rel=1e-3, what=f'd loss / d p[{i}]')
    with torch.no_grad():                                        # value-only path (val.py:159-160)
        l2, it2 = crit([t.detach() for t in p], T(g['targets']).cuda())
    rel_close(l2, g['loss'], rel=1e-4, what='loss (no grad)')


@pytest.mark.parametrize('tag', ['focal', 'slide', 'focal_slide', 'nwd', 'all', 'shapeloss'])
def test_compute_loss_branches_match_reference_vectors(golden, tag):
    """FocalLoss / SlideLoss / both stacked / the NWD box term / everything with label smoothing (utils/loss.py:35-60,125-131,162-169,
    378-402; hyp.VisDrone.yaml leaves them off): loss, items and gradients against the reference's own ComputeLoss under those
    hyper-parameters, plus the value-only path."""
    from somi_amd.configs import HYP_VISDRONE
    from somi_amd.loss import ComputeLoss
    g = golden('loss_branch_' + tag)
    hyp = dict(HYP_VISDRONE, **{str(k): float(v) for k, v in zip(g['hyp_keys'], g['hyp_vals'])})
    crit = ComputeLoss(_M(T(g['anchors']), hyp))
    p = [T(g[f'p{i}']).cuda().requires_grad_(True) for i in range(4)]
    loss, items = crit(p, T(g['targets']).cuda())
    rel_close(loss, g['loss'], rel=1e-4, what=f'{tag}: loss')
    rel_close(items, g['items'], rel=1e-4, what=f'{tag}: loss_items')
    loss.backward()
    for i in range(4):
        rel_close(p[i].grad, g[f'g{i}'], rel=1e-3, what=f'{tag}: d loss / d p[{i}]')
    with torch.no_grad():
        l2, _ = crit([t.detach() for t in p], T(g['targets']).cuda())
    rel_close(l2, g['loss'], rel=1e-4, what=f'{tag}: loss (no grad)')


def test_compute_loss_autobalance_matches_reference_vectors(golden):
    """ComputeLoss(autobalance=True) (utils/loss.py:137,197-201): three consecutive calls against the reference's own - loss, items,
    gradients and the balance list each call leaves behind (the kernel reports the per-level objectness means, the list is host state)."""
    from somi_amd.configs import HYP_VISDRONE
    from somi_amd.loss import ComputeLoss
    g = golden('loss_autobalance')
    mdl = _M(T(g['anchors']), dict(HYP_VISDRONE))
    mdl.model[0].stride = T(g['stride'])
    crit = ComputeLoss(mdl, autobalance=True)
    assert crit.ssi == int(g['ssi'])
    for call in range(3):
        p = [T(g[f'c{call}_p{i}']).cuda().requires_grad_(True) for i in range(4)]
        loss, items = crit(p, T(g['targets']).cuda())
        rel_close(loss, g[f'loss{call}'], rel=1e-4, what=f'call {call}: loss')
        rel_close(items, g[f'items{call}'], rel=1e-4, what=f'call {call}: loss_items')
        np.testing.assert_allclose(np.array(crit.balance), g[f'balance{call}'], rtol=1e-5, err_msg=f'balance after call {call}')
        loss.backward()
        for i in range(4):
            rel_close(p[i].grad, g[f'c{call}_g{i}'], rel=1e-3, what=f'call {call}: d loss / d p[{i}]')


def _wbf_inputs(seed, nm=2, n=120):
    rng = np.random.RandomState(seed)
    boxes, scores, labels = [], [], []
    base = rng.uniform(0.05, 0.8, (40, 2)).astype(np.float32)
    for t in range(nm):
        k = rng.randint(n // 2, n)
        pick = rng.randint(0, 40, k)
        xy = base[pick] + rng.normal(0, 0.004, (k, 2)).astype(np.float32)
        wh = rng.uniform(0.02, 0.15, (k, 2)).astype(np.float32)
        b = np.concatenate([xy, xy + wh], 1).astype(np.float32)
        b[:3] = b[:3, [2, 3, 0, 1]]                             # reversed corners
        b[3] = [0.2, 0.2, 0.2, 0.5]                             # zero area
        b[4] = [-0.1, 0.9, 0.3, 1.2]                            # needs clipping
        s = rng.uniform(0.0, 1.0, k).astype(np.float32)
        s[5:8] = s[5]                                           # score ties
        boxes.append(b), scores.append(s), labels.append((pick % 10).astype(np.int64))
    return boxes, scores, labels


@pytest.mark.parametrize('seed,nm,weights', [(0, 2, None), (1, 3, [1.0, 2.0, 1.0]), (2, 1, None)])
def test_wbf_matches_oracle(seed, nm, weights):
    from oracle.somi_ref.wbf import weighted_boxes_fusion as oracle
    from somi_amd.wbf import weighted_boxes_fusion
    b, s, l = _wbf_inputs(seed, nm)
    wb, wsc, wl = oracle([x.tolist() for x in b], [x.tolist() for x in s], [x.tolist() for x in l], weights=weights,
                         iou_thr=0.67, skip_box_thr=0.01)       # wbf.py:34-35
    gb, gs, gl = weighted_boxes_fusion(b, s, l, weights=weights, iou_thr=0.67, skip_box_thr=0.01)
    assert gb.shape == wb.shape and len(gs) == len(wsc)
    assert np.array_equal(gl, wl)                               # same clusters, same order
    assert np.array_equal(gb, wb.astype(np.float32))            # coordinates bit-exact after the float32 store
    assert np.array_equal(gs, wsc.astype(np.float32))


def test_wbf_empty():
    from somi_amd.wbf import weighted_boxes_fusion
    gb, gs, gl = weighted_boxes_fusion([np.zeros((0, 4))], [np.zeros(0)], [np.zeros(0)])
    assert gb.shape == (0, 4) and gs.shape == (0,) and gl.shape == (0,)
    gb, gs, gl = weighted_boxes_fusion([np.array([[0.1, 0.1, 0.2, 0.2]])], [np.array([0.001])], [np.array([1])], skip_box_thr=0.01)
    assert gb.shape == (0, 4)


def test_wbf_batch_matches_oracle_image_by_image():
    """somi_wbf_batch_f32 (one workgroup per image, fed by NMS-shaped (B, max_det, 6) + count tensors of two models, wbf.py:44-68)
    against the oracle run per image on the same normalised, clipped boxes - bit-exact, including an image where one model and an
    image where both models have no detections."""
    import torch
    from oracle.somi_ref.wbf import weighted_boxes_fusion as oracle
    from somi_amd.wbf import weighted_boxes_fusion_batch
    B, max_det, nm, S = 6, 40, 2, 640.0
    rng = np.random.default_rng(11)
    dets, counts = [], []
    for t in range(nm):
        d = np.zeros((B, max_det, 6), np.float32)
        c = rng.integers(5, max_det + 1, B).astype(np.int32)
        c[3] = 0 if t == 0 else c[3]                            # image 3: model 0 found nothing
        c[4] = 0                                                # image 4: nobody did
        for b in range(B):
            base = np.random.default_rng(100 + b).uniform(20, 560, (12, 2)).astype(np.float32)   # the models see the same objects
            pick = rng.integers(0, 12, max_det)
            xy = base[pick] + rng.normal(0, 3.0, (max_det, 2)).astype(np.float32)
            wh = rng.uniform(12, 90, (max_det, 2)).astype(np.float32)
            d[b, :, :2], d[b, :, 2:4] = xy, xy + wh
            d[b, :, 4] = np.sort(rng.uniform(0.0, 1.0, max_det).astype(np.float32))[::-1]       # NMS rows come sorted by confidence
            d[b, :, 5] = pick % 10
        d[0, 1, :4] = [-30, 600, 200, 700]                      # leaves the image: clipped
        dets.append(d), counts.append(c)
    gb, gs, gl, gc = weighted_boxes_fusion_batch([torch.from_numpy(d).cuda() for d in dets], [torch.from_numpy(c).cuda() for c in counts],
                                                 (S, S), weights=None, iou_thr=0.67, skip_box_thr=0.01)
    gb, gs, gl, gc = gb.cpu().numpy(), gs.cpu().numpy(), gl.cpu().numpy(), gc.cpu().numpy()
    for b in range(B):
        bl = [np.clip(d[b, :c[b], :4] / np.float32(S), 0, 1) for d, c in zip(dets, counts)]
        sl = [d[b, :c[b], 4] for d, c in zip(dets, counts)]
        ll = [d[b, :c[b], 5].astype(np.int64) for d, c in zip(dets, counts)]
        wb, wsc, wl = oracle([x.tolist() for x in bl], [x.tolist() for x in sl], [x.tolist() for x in ll], weights=None,
                             iou_thr=0.67, skip_box_thr=0.01)
        k = int(gc[b])
        assert k == len(wsc), f'image {b}: {k} fused boxes, oracle {len(wsc)}'
        assert np.array_equal(gl[b, :k], np.asarray(wl).astype(np.int32))
        assert np.array_equal(gb[b, :k], np.asarray(wb, dtype=np.float64).reshape(-1, 4).astype(np.float32))
        assert np.array_equal(gs[b, :k], np.asarray(wsc).astype(np.float32))


def test_repulsion_matches_reference_vector(golden):
    """RepGT / RepBox (utils/RepulsionLoss.py:47-95) against the value the reference's own function produced."""
    from somi_amd.loss import repulsion_loss
    g = golden('repulsion')
    dev = torch.device('cuda:0')
    rgt, rbox = repulsion_loss(torch.from_numpy(g['pbox']).to(dev), torch.from_numpy(g['gtbox']).to(dev), torch.from_numpy(g['fg']).to(dev))
    np.testing.assert_allclose(rgt.cpu().numpy(), g['rep_gt'], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(rbox.cpu().numpy(), g['rep_box'], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize('B,A,pfg,kw', [(4, 900, 0.3, {}), (3, 2500, 0.1, dict(sigma_repgt=0.5, sigma_repbox=0.3, pnms=0.2, gtnms=0.1)),
                                        (2, 257, 1.0, {}), (3, 300, 0.0, {})])
def test_repulsion_matches_oracle(B, A, pfg, kw):
    """Larger ragged cases against the CPU oracle: shared ground-truth boxes (the same-gt mask), an image without foreground
    anchors, all anchors foreground, and no foreground at all (0/0 like the reference)."""
    from oracle.somi_ref.loss import repulsion_loss as oracle_rep
    from somi_amd.loss import repulsion_loss
    g = torch.Generator().manual_seed(B * 1000 + A)
    ctr = torch.rand(B, A, 2, generator=g) * 60
    wh = torch.rand(B, A, 2, generator=g) * 24 + 2
    pb = torch.cat((ctr - wh / 2, ctr + wh / 2), -1)
    pool = torch.cat((ctr[:, :9] - 9, ctr[:, :9] + 9), -1)
    gb = torch.stack([pool[b, torch.randint(0, 9, (A,), generator=g)] for b in range(B)])
    fg = torch.rand(B, A, generator=g) < pfg
    if B > 2 and pfg > 0:
        fg[1] = False                                           # one image contributes nothing
    want = oracle_rep(pb, gb, fg, **kw)
    dev = torch.device('cuda:0')
    got = repulsion_loss(pb.to(dev), gb.to(dev), fg.to(dev), **kw)
    for a, b, what in zip(got, want, ('rep_gt', 'rep_box')):
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=2e-5, atol=1e-7, err_msg=what)


@pytest.mark.parametrize('B,nc,nl,na,nt,S,ls', [(3, 10, 4, 4, 200, 96, 0.0), (1, 3, 3, 3, 7, 64, 0.1), (5, 1, 4, 4, 60, 64, 0.0),
                                                (2, 80, 3, 3, 900, 128, 0.0), (4, 10, 4, 4, 0, 64, 0.0)])
def test_compute_loss_random_against_oracle(B, nc, nl, na, nt, S, ls):
    """Differential test of the fused loss (value + gradient) on random predictions: 3- and 4-level heads, 1 / 3 / 10 / 80 classes,
    label smoothing, no targets at all, and 900 targets of which many land in the same cell (duplicate (b, a, gj, gi) rows: the
    reference's argsort-then-scatter keeps the highest IoU, utils/loss.py:174-178)."""
    from oracle.somi_ref.loss import ComputeLoss as OLoss
    from somi_amd.configs import HYP_VISDRONE
    from somi_amd.loss import ComputeLoss
    g = torch.Generator().manual_seed(B * 100 + nc + nt)
    no = nc + 5
    strides = [4, 8, 16, 32][:nl] if nl == 4 else [8, 16, 32]
    anchors = torch.rand(nl, na, 2, generator=g) * 6 + 0.5                # in grid units, like Detect.anchors after /stride
    hyp = dict(HYP_VISDRONE, label_smoothing=ls)
    p = [torch.randn(B, na, S // s_, S // s_, no, generator=g) for s_ in strides]
    tg = torch.zeros(nt, 6)
    if nt:
        tg[:, 0] = torch.randint(0, B, (nt,), generator=g).float()
        tg[:, 1] = torch.randint(0, nc, (nt,), generator=g).float()
        tg[:, 2:4] = torch.rand(nt, 2, generator=g) * 0.98 + 0.01
        tg[:, 4:6] = torch.exp(torch.randn(nt, 2, generator=g) * 0.7 - 2.5).clamp(0.01, 0.6)
        tg[: nt // 3, 2:6] = tg[nt // 3: 2 * (nt // 3), 2:6][: nt // 3] + torch.randn(nt // 3, 4, generator=g) * 1e-3   # near-duplicates
        tg[:, 2:6] = tg[:, 2:6].clamp(0.005, 0.995)
    pr = [t.clone().requires_grad_(True) for t in p]
    want, want_items = OLoss(_M(anchors, hyp, nc))(pr, tg)
    want.backward()
    pm = [t.clone().cuda().requires_grad_(True) for t in p]
    got, items = ComputeLoss(_M(anchors, hyp, nc))(pm, tg.cuda())
    rel_close(got, want.detach(), rel=1e-4, what='loss')
    rel_close(items, want_items, rel=1e-4, what='loss items')
    got.backward()
    for i in range(nl):
        gr, gm = pr[i].grad, pm[i].grad.cpu()
        assert (gm - gr).abs().max().item() <= 1e-3 * gr.abs().max().item() + 1e-9, f'd loss / d p[{i}]'


def test_compute_loss_full_size_by_replication():
    """ComputeLoss at the bench step's size (batch 32, 640x640: grids 160 / 80 / 40 / 20, 4 anchors, 10 classes, ~2000 targets) tied to
    the CPU oracle through a size-independent property: a batch made of 16 copies of a 2-image batch has the same per-term means, so
    loss items(32) == items(2), total(32) == 16 x total(2), and every copy's gradient equals the 2-image gradient - and the 2-image
    case is checked against the oracle directly."""
    from oracle.somi_ref.loss import ComputeLoss as OLoss
    from somi_amd.configs import HYP_VISDRONE
    from somi_amd.loss import ComputeLoss
    g = torch.Generator().manual_seed(77)
    nc, na, S, rep = 10, 4, 640, 16
    anchors = torch.rand(4, na, 2, generator=g) * 6 + 0.5
    p2 = [torch.randn(2, na, S // s_, S // s_, nc + 5, generator=g) for s_ in (4, 8, 16, 32)]
    nt = 130
    tg = torch.zeros(nt, 6)
    tg[:, 0] = torch.randint(0, 2, (nt,), generator=g).float()
    tg[:, 1] = torch.randint(0, nc, (nt,), generator=g).float()
    tg[:, 2:4] = torch.rand(nt, 2, generator=g) * 0.98 + 0.01
    tg[:, 4:6] = torch.exp(torch.randn(nt, 2, generator=g) * 0.7 - 3.0).clamp(0.005, 0.4)
    hyp = dict(HYP_VISDRONE)
    pr = [t.clone().requires_grad_(True) for t in p2]
    want, want_items = OLoss(_M(anchors, hyp, nc))(pr, tg)
    want.backward()
    p32 = [t.repeat(rep, 1, 1, 1, 1).cuda().requires_grad_(True) for t in p2]
    tg32 = torch.cat([torch.cat([tg[:, :1] + 2 * r, tg[:, 1:]], 1) for r in range(rep)])
    got, items = ComputeLoss(_M(anchors, hyp, nc))(p32, tg32.cuda())
    rel_close(items, want_items, rel=1e-4, what='loss items of the replicated batch')
    rel_close(got, want.detach() * rep, rel=1e-4, what='total loss of the replicated batch')
    got.backward()
    for i in range(4):
        gm, gr = p32[i].grad.cpu(), pr[i].grad
        scale = gr.abs().max().item()
        for r in (0, 7, 15):
            assert (gm[2 * r:2 * r + 2] - gr).abs().max().item() <= 1e-3 * scale + 1e-9, f'd loss / d p[{i}], copy {r}'
