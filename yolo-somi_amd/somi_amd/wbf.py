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
