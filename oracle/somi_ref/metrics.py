"""Oracle (test infrastructure): CPU restatement of the validation metrics that follow NMS in val.py.

Follows val.py:50-71 (process_batch: which detections count as correct at the 10 IoU levels),
utils/metrics.py:21-95 (ap_per_class, compute_ap: per-class precision / recall curves and the 101-point
interpolated AP) and utils/metrics.py:98-142 (ConfusionMatrix.process_batch).  Pinned by tests/golden/val_metrics.npz, produced by the reference's own functions.

Ties: the reference orders candidate matches with `argsort()[::-1]` and detections with `np.argsort(-conf)`, both
unstable sorts, so the outcome for *exactly equal* IoUs / confidences depends on numpy's sort internals.  This
restatement (and the device path) breaks such ties by lowest index; the golden inputs have no exact ties.
"""
import numpy as np
import torch

from .nms import box_iou


def process_batch(detections, labels, iouv):
    """detections (N,6) x1,y1,x2,y2,conf,cls; labels (M,5) cls,x1,y1,x2,y2; iouv (T,) -> correct (N,T) bool.

    val.py:59-70 per IoU level: candidate pairs (label, detection) with IoU >= level and equal class are sorted by IoU
    (descending), reduced to the best label per detection (`np.unique` on the detection column keeps the first = highest
    IoU occurrence and re-orders by detection index) and then to one detection per label - the first in that order,
    i.e. the LOWEST detection index that chose the label, not the highest IoU.
    """
    detections, labels, iouv = torch.as_tensor(detections), torch.as_tensor(labels), torch.as_tensor(iouv)
    N, T = detections.shape[0], iouv.shape[0]
    correct = np.zeros((N, T), dtype=bool)
    if N == 0 or labels.shape[0] == 0:
        return torch.from_numpy(correct)
    iou = box_iou(labels[:, 1:], detections[:, :4])               # (M, N)
    same = labels[:, 0:1] == detections[:, 5]
    iou_np = iou.numpy()
    for i in range(T):
        cand = ((iou >= iouv[i]) & same).numpy()
        if not cand.any():
            continue
        masked = np.where(cand, iou_np, -1.0)
        best_l = masked.argmax(0)                                 # first maximum: lowest label index on ties
        has = cand.any(0)
        taken = {}
        for d in np.nonzero(has)[0]:                              # ascending detection index
            taken.setdefault(int(best_l[d]), int(d))
        for d in taken.values():
            correct[d, i] = True
    return torch.from_numpy(correct)


def compute_ap(recall, precision):
    """utils/metrics.py:76-95 ('interp' method): 101-point interpolated area under the precision envelope."""
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([1.0], precision, [0.0]))
    mpre = np.flip(np.maximum.accumulate(np.flip(mpre)))
    x = np.linspace(0, 1, 101)
    trapz = getattr(np, 'trapezoid', None) or np.trapz
    return trapz(np.interp(x, mrec, mpre), x), mpre, mrec


def ap_per_class(tp, conf, pred_cls, target_cls):
    """utils/metrics.py:21-74 without the plotting: -> p, r, ap (nc_present, T), f1, classes (int32).

    p, r, f1 are taken at the confidence (one of 1000 grid points) that maximises the mean F1 over the present classes.
    """
    tp, conf, pred_cls, target_cls = (np.asarray(a) for a in (tp, conf, pred_cls, target_cls))
    order = np.argsort(-conf, kind='stable')
    tp, conf, pred_cls = tp[order], conf[order], pred_cls[order]
    classes = np.unique(target_cls)
    nc = classes.shape[0]
    px = np.linspace(0, 1, 1000)
    ap, p, r = np.zeros((nc, tp.shape[1])), np.zeros((nc, 1000)), np.zeros((nc, 1000))
    for ci, c in enumerate(classes):
        sel = pred_cls == c
        n_l, n_p = (target_cls == c).sum(), sel.sum()
        if n_p == 0 or n_l == 0:
            continue
        fpc, tpc = (1 - tp[sel]).cumsum(0), tp[sel].cumsum(0)
        recall = tpc / (n_l + 1e-16)
        r[ci] = np.interp(-px, -conf[sel], recall[:, 0], left=0)
        precision = tpc / (tpc + fpc)
        p[ci] = np.interp(-px, -conf[sel], precision[:, 0], left=1)
        for j in range(tp.shape[1]):
            ap[ci, j] = compute_ap(recall[:, j], precision[:, j])[0]
    f1 = 2 * p * r / (p + r + 1e-16)
    i = f1.mean(0).argmax()
    return p[:, i], r[:, i], ap, f1[:, i], classes.astype('int32')


class ConfusionMatrix:
    """utils/metrics.py:98-142: matrix[predicted class, true class], row / column `nc` = background.

    Detections above `conf` are matched to labels regardless of class: every detection picks the label of highest IoU (> iou_thres),
    every label keeps the detection of highest IoU among those that picked it.  A label without a detection counts as
    (background, class); a detection that matched nothing counts as (class, background) - but only in images where at least one
    match exists (`if n:` :138), a quirk kept here.  Exact IoU ties (the reference's argsort is unstable) go to the lowest index.
    """

    def __init__(self, nc, conf=0.25, iou_thres=0.2):
        self.matrix = np.zeros((nc + 1, nc + 1))
        self.nc, self.conf, self.iou_thres = nc, conf, iou_thres

    def process_batch(self, detections, labels):
        detections, labels = torch.as_tensor(detections), torch.as_tensor(labels)
        detections = detections[detections[:, 4] > self.conf]
        gt = labels[:, 0].int().numpy()
        dc = detections[:, 5].int().numpy()
        iou = box_iou(labels[:, 1:], detections[:, :4]).numpy()               # (M, N)
        M, N = iou.shape
        ok = iou > self.iou_thres
        det_of = np.full(M, -1)
        if ok.any():
            masked = np.where(ok, iou, -1.0)
            pick = np.where(ok.any(0), masked.argmax(0), -1)                    # best label per detection (first maximum)
            for l in range(M):
                mine = np.nonzero(pick == l)[0]
                if mine.size:
                    det_of[l] = mine[np.argmax(iou[l, mine])]                   # best detection per label (first maximum)
        any_match = (det_of >= 0).any()
        for l in range(M):
            if det_of[l] >= 0:
                self.matrix[dc[det_of[l]], gt[l]] += 1
            else:
                self.matrix[self.nc, gt[l]] += 1
        if any_match:
            taken = set(det_of[det_of >= 0].tolist())
            for d in range(N):
                if d not in taken:
                    self.matrix[dc[d], self.nc] += 1
