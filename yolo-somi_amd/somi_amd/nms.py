"""`non_max_suppression` with the reference's signature (utils/general.py:629-630), executed by somi_nms_f32.

Returns a list of (n_i, 6) [x1, y1, x2, y2, conf, cls] tensors on the prediction's device.  One device->host copy of
the per-image counts happens at the end (the reference syncs per image inside its Python loop, and val.py:189 copies
results to the CPU anyway).  `labels` (a-priori boxes for autolabelling, :651-658) are appended as rows of the prediction tensor.
"""
import torch

from . import _lib
from ._lib import check
from .ops import _ptr, _stream


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False,
                        labels=(), max_det=300):
    det, count = non_max_suppression_raw(prediction, conf_thres, iou_thres, classes, agnostic, multi_label, labels, max_det)
    counts = count.cpu().tolist()
    return [det[b, :counts[b]] for b in range(det.shape[0])]


def non_max_suppression_raw(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False,
                            labels=(), max_det=300):
    """Same selection, left on the device: det (B, max_det, 6) and count (B) int32 - no host synchronisation, so a following device
    step (wbf.weighted_boxes_fusion_batch) can consume them stream-ordered."""
    assert 0 <= conf_thres <= 1, f'Invalid Confidence threshold {conf_thres}, valid values are between 0.0 and 1.0'
    assert 0 <= iou_thres <= 1, f'Invalid IoU {iou_thres}, valid values are between 0.0 and 1.0'
    if not prediction.is_cuda:
        raise RuntimeError('somi_amd NMS runs on the MI355X only (no CPU fallback)')
    if prediction.dtype != torch.float32 or not prediction.is_contiguous():
        raise RuntimeError('prediction tensor has to be contiguous float32')
    B, n, no = prediction.shape
    nc = no - 5
    if labels and any(len(l) for l in labels):
        # a-priori labels (utils/general.py:651-658): each image's label rows [cls, x, y, w, h] become predictions with objectness 1 and a
        # one-hot class, appended BEHIND the image's own rows (the candidate order the reference builds); padding rows have
        # objectness 0 and drop out at the first threshold
        kmax = max(len(l) for l in labels)
        aug = torch.zeros(B, n + kmax, no, dtype=torch.float32, device=prediction.device)
        aug[:, :n] = prediction
        for xi, l in enumerate(labels):
            if len(l):
                l = torch.as_tensor(l, dtype=torch.float32, device=prediction.device)
                aug[xi, n:n + len(l), :4] = l[:, 1:5]
                aug[xi, n:n + len(l), 4] = 1.0
                aug[xi, torch.arange(n, n + len(l), device=prediction.device), l[:, 0].long() + 5] = 1.0
        prediction, n = aug, n + kmax
    ml = bool(multi_label) and nc > 1
    mask = None                                                  # NULL: keep every class
    if classes is not None:
        import ctypes
        words = [0] * ((nc + 63) // 64)
        for c in classes:
            if 0 <= int(c) < nc:
                words[int(c) >> 6] |= 1 << (int(c) & 63)
        mask = (ctypes.c_uint64 * len(words))(*words)
    L = _lib.lib()
    nbytes = L.somi_nms_workspace_bytes(B, n, nc, int(ml))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=prediction.device)
    det = torch.empty(B, max_det, 6, dtype=torch.float32, device=prediction.device)
    count = torch.empty(B, dtype=torch.int32, device=prediction.device)
    check(L.somi_nms_f32(_ptr(prediction), B, n, nc, float(conf_thres), float(iou_thres), int(ml), int(bool(agnostic)), mask,
                         int(max_det), _ptr(det), _ptr(count), _ptr(ws), nbytes, _stream()), 'non_max_suppression')
    return det, count
