"""`weighted_boxes_fusion` with the signature wbf.py:68 uses (ensemble_boxes 1.0.9: boxes_list, scores_list, labels_list,
weights, iou_thr, skip_box_thr; conf_type 'avg'), executed by somi_wbf_f32 on the MI355X.

Inputs per model: boxes (n_t,4) xyxy normalised to [0,1], scores (n_t), labels (n_t).  Returns numpy arrays
(boxes (n,4), scores (n), labels (n)) sorted by descending score, as the package does.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check
from .ops import _ptr, _stream


def weighted_boxes_fusion(boxes_list, scores_list, labels_list, weights=None, iou_thr=0.55, skip_box_thr=0.0,
                          conf_type='avg', allows_overflow=False, device='cuda'):
    if conf_type != 'avg' or allows_overflow:
        raise NotImplementedError("only conf_type='avg', allows_overflow=False (what wbf.py uses) is on the HIP path")
    if not torch.cuda.is_available():
        raise RuntimeError('somi_amd WBF runs on the MI355X only (no CPU fallback)')
    nm = len(boxes_list)
    if weights is None or len(weights) != nm:
        weights = [1.0] * nm
    bs, ss, ls, ms = [], [], [], []
    for t in range(nm):
        b = torch.as_tensor(np.asarray(boxes_list[t], dtype=np.float32)).reshape(-1, 4)
        s = torch.as_tensor(np.asarray(scores_list[t], dtype=np.float32)).reshape(-1)
        l = torch.as_tensor(np.asarray(labels_list[t])).reshape(-1).to(torch.int32)
        if not (len(b) == len(s) == len(l)):
            raise ValueError('boxes / scores / labels length mismatch')
        bs.append(b), ss.append(s), ls.append(l), ms.append(torch.full((len(b),), t, dtype=torch.int32))
    boxes, scores = torch.cat(bs).to(device).contiguous(), torch.cat(ss).to(device).contiguous()
    labels, model = torch.cat(ls).to(device).contiguous(), torch.cat(ms).to(device).contiguous()
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0, 4)), np.zeros((0,)), np.zeros((0,))
    L = _lib.lib()
    nbytes = L.somi_wbf_workspace_bytes(n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
    ob = torch.empty(n, 4, dtype=torch.float32, device=device)
    osc = torch.empty(n, dtype=torch.float32, device=device)
    ol = torch.empty(n, dtype=torch.int32, device=device)
    cnt = torch.zeros(1, dtype=torch.int32, device=device)
    w = (C.c_float * nm)(*[float(v) for v in weights])
    check(L.somi_wbf_f32(_ptr(boxes), _ptr(scores), _ptr(labels), _ptr(model), n, nm, w, float(iou_thr), float(skip_box_thr),
                         _ptr(ob), _ptr(osc), _ptr(ol), _ptr(cnt), _ptr(ws), nbytes, _stream()), 'weighted_boxes_fusion')
    k = int(cnt.item())
    return ob[:k].cpu().numpy(), osc[:k].cpu().numpy(), ol[:k].cpu().numpy().astype(np.float64)


def weighted_boxes_fusion_batch(dets, counts, img_size, weights=None, iou_thr=0.55, skip_box_thr=0.0):
    """wbf.py:44-68 for a whole batch on the device.  dets[t] (B, max_det, 6) / counts[t] (B) = nms.non_max_suppression_raw of model t
    (pixels); img_size = (width, height) the boxes are normalised by.  Returns device tensors (boxes (B,N,4) in [0,1], scores (B,N),
    labels (B,N) int32, count (B) int32), N = len(dets) * max_det; row order per image = descending fused score.  No host sync."""
    nm = len(dets)
    if nm < 1 or nm != len(counts):
        raise ValueError('one (det, count) pair per model')
    B, max_det, six = dets[0].shape
    for d, c in zip(dets, counts):
        if not d.is_cuda or d.dtype != torch.float32 or not d.is_contiguous() or d.shape != (B, max_det, 6):
            raise RuntimeError('det tensors have to be contiguous float32 (B, max_det, 6) on the MI355X')
        if not c.is_cuda or c.dtype != torch.int32 or c.shape != (B,):
            raise RuntimeError('count tensors have to be int32 (B) on the MI355X')
    dev = dets[0].device
    N = nm * max_det
    L = _lib.lib()
    nbytes = L.somi_wbf_batch_workspace_bytes(B, max_det, nm)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    ob = torch.empty(B, N, 4, dtype=torch.float32, device=dev)
    osc = torch.empty(B, N, dtype=torch.float32, device=dev)
    ol = torch.empty(B, N, dtype=torch.int32, device=dev)
    cnt = torch.empty(B, dtype=torch.int32, device=dev)
    w = (C.c_float * nm)(*[float(v) for v in (weights if weights is not None and len(weights) == nm else [1.0] * nm)])
    dp = (C.c_void_p * nm)(*[_ptr(d) for d in dets])
    cp = (C.c_void_p * nm)(*[_ptr(c) for c in counts])
    check(L.somi_wbf_batch_f32(dp, cp, B, max_det, nm, w, float(img_size[0]), float(img_size[1]), float(iou_thr), float(skip_box_thr),
                               _ptr(ob), _ptr(osc), _ptr(ol), _ptr(cnt), _ptr(ws), nbytes, _stream()), 'weighted_boxes_fusion_batch')
    return ob, osc, ol, cnt
