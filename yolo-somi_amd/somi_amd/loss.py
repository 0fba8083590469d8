"""`ComputeLoss` with the reference's interface (utils/loss.py:112-208) on top of somi_yolo_loss_f32.

`ComputeLoss(model)(p, targets) -> (loss[1], loss_items[3] detached)`; p is the list of (B,na,ny,nx,no) training
outputs, targets (nt,6) = [image, class, x, y, w, h] normalised.  The loss value and d loss / d p come out of the same
fused launches; autograd sees one node.  hyp keys read: cls_pw, obj_pw, fl_gamma, slide_ratio, nwdloss, shapeloss, box, obj, cls,
anchor_t (+label_smoothing) - the ones the reference's __init__ / __call__ read.  FocalLoss (fl_gamma > 0), SlideLoss (slide_ratio > 0)
and the NWD box term (nwdloss > 0; constant 12.8, or 2.5 under shapeloss > 0) run inside the same kernels; autobalance is host state
(the balance list, updated from the per-level objectness means the launch reports) as in the reference.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import LossDesc, check
from .ops import _ptr, _stream


def smooth_BCE(eps=0.1):
    """utils/loss.py:14-15."""
    return 1.0 - 0.5 * eps, 0.5 * eps


class _AttachGrad(torch.autograd.Function):
    """Makes the pre-computed d loss / d p an autograd edge from the loss to the prediction tensors."""

    @staticmethod
    def forward(ctx, loss, grads, *p):
        ctx.grads = grads
        return loss.clone()

    @staticmethod
    def backward(ctx, go):
        return (None, None) + tuple(g * go for g in ctx.grads)


class ComputeLoss:
    def __init__(self, model, autobalance=False):
        self.sort_obj_iou = False
        h = model.hyp
        det = model.model[-1]
        self.cp, self.cn = smooth_BCE(eps=h.get('label_smoothing', 0.0))
        self.balance = {3: [4.0, 1.0, 0.4]}.get(det.nl, [4.0, 1.0, 0.25, 0.06, 0.02])       # utils/loss.py:135
        self.gr, self.hyp, self.autobalance = 1.0, h, autobalance
        self.ssi = list(det.stride).index(16) if autobalance else 0                           # stride 16 index (:137)
        self.na, self.nc, self.nl, self.anchors = det.na, det.nc, det.nl, det.anchors
        if self.nl > 4:
            raise NotImplementedError('at most 4 detection levels')
        self._anc = None

    def _anchors(self, dev):
        if self._anc is None or self._anc.device != dev:
            self._anc = self.anchors.detach().float().contiguous().to(dev)
        return self._anc

    def _launch(self, p, targets, need_grad):
        d = LossDesc()
        grads = []
        for i, t in enumerate(p):
            if t.dtype != torch.float32 or not t.is_cuda:
                raise RuntimeError('prediction tensors have to be float32 on the GPU (no CPU fallback)')
            d.p[i] = _ptr(t)
            g = torch.empty_like(t) if need_grad else None
            grads.append(g)
            d.grad[i] = _ptr(g)
            d.ny[i], d.nx[i] = t.shape[2], t.shape[3]
        dev = p[0].device
        tg = targets.detach().to(dev).float().contiguous()
        d.nl, d.na, d.nc, d.B, d.nt = len(p), self.na, self.nc, p[0].shape[0], tg.shape[0]
        d.targets, d.anchors = (_ptr(tg) if tg.numel() else None), _ptr(self._anchors(dev))
        for i in range(len(p)):
            d.balance[i] = self.balance[i]
        h = self.hyp
        d.box_gain, d.obj_gain, d.cls_gain = float(h['box']), float(h['obj']), float(h['cls'])
        d.cls_pw, d.obj_pw, d.anchor_t = float(h['cls_pw']), float(h['obj_pw']), float(h['anchor_t'])
        d.cp, d.cn, d.gr = float(self.cp), float(self.cn), float(self.gr)
        d.fl_gamma, d.slide = float(h['fl_gamma']), int(h['slide_ratio'] > 0)
        d.nwd_ratio = 0.5 if h['nwdloss'] > 0 else 0.0           # iou_ratio, utils/loss.py:148
        d.nwd_constant = 2.5 if h.get('shapeloss', 0) > 0 else 12.8   # wasserstein (utils/metrics.py:373) / wasserstein_loss (:341), :163-166
        L = _lib.lib()
        nbytes = L.somi_loss_workspace_bytes(C.byref(d))
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
        out = torch.empty(8, dtype=torch.float32, device=dev)
        check(L.somi_yolo_loss_f32(C.byref(d), _ptr(out), _ptr(ws), nbytes, _stream()), 'ComputeLoss')
        if self.autobalance:                                      # utils/loss.py:197-201 (a host sync per call, like the reference's .item())
            obji = out[4:4 + len(p)].tolist()
            bal = [self.balance[i] * 0.9999 + 0.0001 / obji[i] for i in range(len(p))] + list(self.balance[len(p):])
            self.balance = [x / bal[self.ssi] for x in bal]
        return out[:4], grads

    def __call__(self, p, targets):
        if len(p) != self.nl:
            raise RuntimeError(f'expected {self.nl} prediction levels, got {len(p)}')
        need_grad = torch.is_grad_enabled() and any(t.requires_grad for t in p)
        pc = [t.detach().contiguous() for t in p]
        out, grads = self._launch(pc, targets, need_grad)
        loss = out[0:1]
        if need_grad:
            loss = _AttachGrad.apply(loss, grads, *p)
        return loss, out[1:4].detach()


def repulsion_loss(pbox, gtbox, fg_mask, sigma_repgt=0.9, sigma_repbox=0, pnms=0, gtnms=0):
    """RepGT + RepBox with the signature of utils/RepulsionLoss.py:47 -> (rep_gt, rep_box) 0-d GPU tensors.  The reference
    imports this term but never adds it to the loss (utils/loss.py:8); it is provided the same way: optional, value only."""
    if not (pbox.is_cuda and gtbox.is_cuda and fg_mask.is_cuda):
        raise RuntimeError('repulsion_loss runs on GPU tensors only (no CPU fallback)')
    if pbox.dim() != 3 or pbox.shape[-1] != 4 or gtbox.shape != pbox.shape or fg_mask.shape != pbox.shape[:2]:
        raise RuntimeError('repulsion_loss expects pbox, gtbox (B,A,4) and fg_mask (B,A)')
    Bn, A = fg_mask.shape
    pb, gb = pbox.detach().float().contiguous(), gtbox.detach().float().contiguous()
    fg = fg_mask.to(torch.uint8).contiguous()
    out = torch.empty(2, device=pbox.device, dtype=torch.float32)
    L = _lib.lib()
    nbytes = L.somi_repulsion_workspace_bytes(Bn, A)
    ws = torch.empty(nbytes, device=pbox.device, dtype=torch.uint8)
    check(L.somi_repulsion_loss_f32(pb.data_ptr(), gb.data_ptr(), fg.data_ptr(), Bn, A, float(sigma_repgt), float(sigma_repbox),
                                    float(pnms), float(gtnms), out.data_ptr(), ws.data_ptr(), nbytes,
                                    torch.cuda.current_stream().cuda_stream), 'repulsion_loss')
    return out[0], out[1]
