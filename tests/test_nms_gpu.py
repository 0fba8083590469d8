"""NMS on the MI355X against the golden outputs of the reference's non_max_suppression flow and the CPU oracle.
Bar: bit-exact selection AND values (BASELINE.json: 'bit-exact for NMS index selection')."""
import pytest
import torch

pytestmark = pytest.mark.gpu
T = torch.from_numpy

CASES = dict(default=dict(conf_thres=0.25, iou_thres=0.45),
             val=dict(conf_thres=0.4, iou_thres=0.2, multi_label=True),
             bench=dict(conf_thres=0.001, iou_thres=0.6, multi_label=True),
             agnostic=dict(conf_thres=0.3, iou_thres=0.5, agnostic=True),
             classes=dict(conf_thres=0.2, iou_thres=0.45, classes=[1, 3, 7]),
             maxdet=dict(conf_thres=0.05, iou_thres=0.9, multi_label=True, max_det=20),
             none=dict(conf_thres=0.9999, iou_thres=0.45))


@pytest.mark.parametrize('tag', list(CASES))
def test_nms_matches_reference_vectors(golden, tag):
    from somi_amd.nms import non_max_suppression
    g = golden('nms')
    out = non_max_suppression(T(g['pred']).cuda(), **CASES[tag])
    for b, o in enumerate(out):
        want = T(g[f'{tag}_{b}'])
        assert o.shape == want.shape, (tag, b, o.shape, want.shape)
        assert torch.equal(o.cpu(), want), f'{tag} image {b}: selection/values differ'


@pytest.mark.parametrize('n,nc,conf,iou,ml', [(20000, 10, 0.001, 0.6, True),      # > 30000 candidates -> max_nms cap
                                              (5000, 3, 0.25, 0.45, False),
                                              (1500, 10, 0.1, 0.3, True),
                                              (700, 1, 0.1, 0.5, True),            # nc == 1 switches multi_label off
                                              (900, 150, 0.3, 0.5, True)])         # head too wide for the LDS row staging
def test_nms_matches_oracle_random(n, nc, conf, iou, ml):
    from oracle.somi_ref.nms import non_max_suppression as oracle
    from somi_amd.nms import non_max_suppression
    g = torch.Generator().manual_seed(n + nc)
    B = 3
    pred = torch.rand(B, n, 5 + nc, generator=g)
    pred[..., :2] *= 640
    pred[..., 2:4] = pred[..., 2:4] * 120 + 4
    pred[..., 4] = pred[..., 4] ** 2
    # clusters of near-duplicates and exact ties
    pred[0, 100:160, :4] = pred[0, 100, :4] + torch.rand(60, 4, generator=g)
    pred[1, 10:20] = pred[1, 10]
    pred[2, :, 4] = 0                                        # an image with no candidates
    want = oracle(pred.clone(), conf, iou, multi_label=ml)
    got = non_max_suppression(pred.cuda(), conf, iou, multi_label=ml)
    for b in range(B):
        assert got[b].shape == want[b].shape, (b, got[b].shape, want[b].shape)
        assert torch.equal(got[b].cpu(), want[b])


@pytest.mark.parametrize('kw', [dict(conf_thres=0.001, iou_thres=0.6, multi_label=True),       # val.sh benchmark setting on a COCO head
                                dict(conf_thres=0.25, iou_thres=0.45),
                                dict(conf_thres=0.05, iou_thres=0.5, multi_label=True, classes=[0, 17, 63, 64, 79]),
                                dict(conf_thres=0.1, iou_thres=0.45, classes=[70]),
                                dict(conf_thres=0.2, iou_thres=0.5, agnostic=True)])
def test_nms_80_class_head(kw):
    """nc = 80 (coco128, BASELINE target): the class filter is a bit array, not one 64-bit word."""
    from oracle.somi_ref.nms import non_max_suppression as oracle
    from somi_amd.nms import non_max_suppression
    g = torch.Generator().manual_seed(80)
    B, n, nc = 2, 3000, 80
    pred = torch.rand(B, n, 5 + nc, generator=g)
    pred[..., :2] *= 640
    pred[..., 2:4] = pred[..., 2:4] * 150 + 4
    pred[..., 4] = pred[..., 4] ** 2
    pred[..., 5:] = pred[..., 5:] ** 6                       # a few confident classes per box, most below the threshold
    pred[0, 50:90, :4] = pred[0, 50, :4] + torch.rand(40, 4, generator=g)
    want = oracle(pred.clone(), **kw)
    got = non_max_suppression(pred.cuda(), **kw)
    for b in range(B):
        assert got[b].shape == want[b].shape, (b, got[b].shape, want[b].shape)
        assert torch.equal(got[b].cpu(), want[b])
    assert sum(int(w.shape[0]) for w in want) > 0


def test_nms_with_a_priori_labels(golden):
    """utils/general.py:651-658 (`labels=`, the save_hybrid path of val.py:161-164): label rows join the candidates behind the image's
    own predictions; against the reference's own output."""
    from somi_amd.nms import non_max_suppression
    g = golden('nms')
    lb = [T(g['labels_in0']).cuda(), torch.zeros(0, 5, device='cuda')]
    out = non_max_suppression(T(g['pred']).cuda(), conf_thres=0.3, iou_thres=0.5, labels=lb, multi_label=True)
    for b, o in enumerate(out):
        want = T(g[f'withlabels_{b}'])
        assert o.shape == want.shape and torch.equal(o.cpu(), want), f'image {b}'
    assert any(float(o[:, 4].max()) == 1.0 for o in out if len(o))          # a label row (confidence 1) survived


def test_nms_argument_errors():
    from somi_amd.nms import non_max_suppression
    p = torch.zeros(1, 10, 15, device='cuda')
    with pytest.raises(AssertionError, match='Invalid Confidence threshold'):
        non_max_suppression(p, conf_thres=1.5)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        non_max_suppression(p.cpu())
