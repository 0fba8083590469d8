"""Oracle (test infrastructure): CPU restatement of post-processing NMS.

``non_max_suppression`` follows utils/general.py:629-711 (filter by objectness, conf = obj*cls,
xywh->xyxy, multi-label expansion or best class, cap at 30000 by score, class-offset trick,
NMS, cap at max_det).  The NMS core itself is third-party in the reference
(``torchvision.ops.nms``, torchvision==0.14.1 per requirements.txt:188, call site
utils/general.py:694) and is not under /root/reference: **parity unpinned** for that step.
It is restated here as the published greedy algorithm: visit boxes by descending score
(ties: lower original index first - a stable sort), keep a box, drop every later box whose
IoU with it is strictly greater than the threshold.  The reference's own plain-torch loop
``NMS()`` (utils/general.py:925-951) documents the same control flow.
"""
import numpy as np
import torch


def xywh2xyxy(x):
    """centre/size -> corners (utils/general.py:541-547)."""
    y = x.clone() if isinstance(x, torch.Tensor) else np.copy(x)
    y[:, 0] = x[:, 0] - x[:, 2] / 2
    y[:, 1] = x[:, 1] - x[:, 3] / 2
    y[:, 2] = x[:, 0] + x[:, 2] / 2
    y[:, 3] = x[:, 1] + x[:, 3] / 2
    return y


def box_iou(box1, box2):
    """Pairwise IoU of xyxy boxes, (N,4) x (M,4) -> (N,M) (utils/metrics.py:208-235)."""
    a1 = (box1[:, 2] - box1[:, 0]) * (box1[:, 3] - box1[:, 1])
    a2 = (box2[:, 2] - box2[:, 0]) * (box2[:, 3] - box2[:, 1])
    inter = (torch.min(box1[:, None, 2:], box2[:, 2:]) - torch.max(box1[:, None, :2], box2[:, :2])).clamp(0).prod(2)
    return inter / (a1[:, None] + a2 - inter)


def greedy_nms(boxes, scores, iou_thres):
    """Indices kept by greedy NMS, in descending-score order (see module docstring).

    fp32 arithmetic in exactly this order, so a device kernel can match bit for bit:
    area = (x2-x1)*(y2-y1); inter = max(0, min(x2)-max(x1)) * max(0, min(y2)-max(y1));
    iou = inter / (area_i + area_j - inter); suppress iff iou > thres.
    """
    if boxes.numel() == 0:
        return torch.zeros(0, dtype=torch.long)
    b = boxes.detach().cpu().float().numpy()
    s = scores.detach().cpu().float().numpy()
    order = np.argsort(-s, kind='stable')
    x1, y1, x2, y2 = (b[order, k] for k in range(4))
    area = (x2 - x1) * (y2 - y1)
    n = len(order)
    dead = np.zeros(n, dtype=bool)
    keep = []
    thr = np.float32(iou_thres)
    for i in range(n):
        if dead[i]:
            continue
        keep.append(order[i])
        if i + 1 < n:
            xx1 = np.maximum(x1[i], x1[i + 1:])
            yy1 = np.maximum(y1[i], y1[i + 1:])
            xx2 = np.minimum(x2[i], x2[i + 1:])
            yy2 = np.minimum(y2[i], y2[i + 1:])
            inter = np.maximum(np.float32(0), xx2 - xx1) * np.maximum(np.float32(0), yy2 - yy1)
            iou = inter / (area[i] + area[i + 1:] - inter)
            dead[i + 1:] |= iou > thr
    return torch.from_numpy(np.asarray(keep, dtype=np.int64))


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False,
                        multi_label=False, labels=(), max_det=300):
    """(B, n, 5+nc) decoded predictions -> list of (n_i, 6) [x1,y1,x2,y2,conf,cls] (utils/general.py:629-711).

    Like the reference this multiplies the class scores by objectness **in place** on rows it
    selects (they are copies after boolean indexing, so the caller's tensor is not changed).
    The 10 s wall-clock bail-out (:707-709) is not restated: it is not arithmetic.
    """
    assert 0 <= conf_thres <= 1 and 0 <= iou_thres <= 1
    nc = prediction.shape[2] - 5
    cand = prediction[..., 4] > conf_thres
    max_wh, max_nms = 4096, 30000
    multi_label &= nc > 1
    out = [torch.zeros((0, 6), device=prediction.device)] * prediction.shape[0]
    for xi, x in enumerate(prediction):
        x = x[cand[xi]]
        if labels and len(labels[xi]):                         # a-priori labels (:651-658)
            l = labels[xi]
            v = torch.zeros((len(l), nc + 5), device=x.device)
            v[:, :4] = l[:, 1:5]
            v[:, 4] = 1.0
            v[range(len(l)), l[:, 0].long() + 5] = 1.0
            x = torch.cat((x, v), 0)
        if not x.shape[0]:
            continue
        x[:, 5:] *= x[:, 4:5]
        box = xywh2xyxy(x[:, :4])
        if multi_label:
            i, j = (x[:, 5:] > conf_thres).nonzero(as_tuple=False).T
            x = torch.cat((box[i], x[i, j + 5, None], j[:, None].float()), 1)
        else:
            conf, j = x[:, 5:].max(1, keepdim=True)
            x = torch.cat((box, conf, j.float()), 1)[conf.view(-1) > conf_thres]
        if classes is not None:
            x = x[(x[:, 5:6] == torch.tensor(classes, device=x.device)).any(1)]
        n = x.shape[0]
        if not n:
            continue
        if n > max_nms:
            x = x[torch.argsort(x[:, 4], descending=True, stable=True)[:max_nms]]
        c = x[:, 5:6] * (0 if agnostic else max_wh)
        keep = greedy_nms(x[:, :4] + c, x[:, 4], iou_thres)
        out[xi] = x[keep[:max_det]]
    return out
