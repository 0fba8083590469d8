"""Oracle (test infrastructure): CPU restatement of the YOLO-SOMI loss.

Follows utils/loss.py:112-262 (ComputeLoss), utils/metrics.py:476-518 (bbox_iou, CIoU branch),
utils/loss.py:14-15 (smooth_BCE) and utils/RepulsionLoss.py:5-95 (repulsion_loss, optional term that
the reference imports but never calls - kept separate and off by default).
The branches hyp.VisDrone.yaml leaves off are restated too: FocalLoss (utils/loss.py:35-60, fl_gamma > 0), SlideLoss
(:378-402, slide_ratio > 0, stacked on top of the focal wrapper exactly as :125-131 stacks them) and the NWD box term
(:162-169 with utils/metrics.py:341-354, nwdloss > 0, shapeloss == 0).  autobalance and the shapeloss variant raise.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def smooth_BCE(eps=0.1):
    """Positive / negative label-smoothing targets (utils/loss.py:14-15)."""
    return 1.0 - 0.5 * eps, 0.5 * eps


def bbox_ciou_xywh(box1, box2, eps=1e-7):
    """CIoU of xywh boxes; box1 is (4,n), box2 is (n,4) (utils/metrics.py:476-518, alpha=1).

    Keeps the reference's quirks: eps added to both heights, union+eps then another +eps in the
    IoU denominator, and alpha_ciou computed without gradient.
    """
    box2 = box2.T
    b1_x1, b1_x2 = box1[0] - box1[2] / 2, box1[0] + box1[2] / 2
    b1_y1, b1_y2 = box1[1] - box1[3] / 2, box1[1] + box1[3] / 2
    b2_x1, b2_x2 = box2[0] - box2[2] / 2, box2[0] + box2[2] / 2
    b2_y1, b2_y2 = box2[1] - box2[3] / 2, box2[1] + box2[3] / 2
    inter = (torch.minimum(b1_x2, b2_x2) - torch.maximum(b1_x1, b2_x1)).clamp(0) * \
            (torch.minimum(b1_y2, b2_y2) - torch.maximum(b1_y1, b2_y1)).clamp(0)
    w1, h1 = b1_x2 - b1_x1, b1_y2 - b1_y1 + eps
    w2, h2 = b2_x2 - b2_x1, b2_y2 - b2_y1 + eps
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / (union + eps)
    cw = torch.maximum(b1_x2, b2_x2) - torch.minimum(b1_x1, b2_x1)
    ch = torch.maximum(b1_y2, b2_y2) - torch.minimum(b1_y1, b2_y1)
    c2 = cw ** 2 + ch ** 2 + eps
    rho2 = ((b2_x1 + b2_x2 - b1_x1 - b1_x2) ** 2 + (b2_y1 + b2_y2 - b1_y1 - b1_y2) ** 2) / 4
    v = (4 / math.pi ** 2) * (torch.atan(w2 / h2) - torch.atan(w1 / h1)).pow(2)
    with torch.no_grad():
        a = v / (v - iou + (1 + eps))
    return iou - (rho2 / c2 + (v * a + eps))          # torch.pow(x, alpha=1) == x


class ComputeLoss:
    """utils/loss.py:112-208.  ``model`` needs .hyp and a last layer with na/nc/nl/anchors."""

    def __init__(self, model, autobalance=False):
        h = model.hyp
        det = model.model[-1]
        self.hyp = h
        self.autobalance = autobalance
        self.ssi = list(det.stride).index(16) if autobalance else 0                        # :137
        # hyp['shapeloss'] > 0 swaps wasserstein_loss for wasserstein (utils/metrics.py:373-397, :163-164): called with its default
        # scale1 = 0 the shape weights ww, hh are 2 * x^0 / (x^0 + y^0) = 1, so the only difference left is the constant, 2.5 for 12.8
        self.nwd_constant = 2.5 if h.get('shapeloss', 0) > 0 else 12.8
        self.fl_gamma, self.slide, self.nwd = float(h['fl_gamma']), h['slide_ratio'] > 0, h['nwdloss'] > 0
        self.cls_pw, self.obj_pw = float(h['cls_pw']), float(h['obj_pw'])
        self.cp, self.cn = smooth_BCE(eps=h.get('label_smoothing', 0.0))
        self.balance = {3: [4.0, 1.0, 0.4]}.get(det.nl, [4.0, 1.0, 0.25, 0.06, 0.02])     # :135
        self.gr = 1.0
        self.na, self.nc, self.nl, self.anchors = det.na, det.nc, det.nl, det.anchors

    def _bce(self, logits, target, pw, auto_iou=None):
        """BCEWithLogits(pos_weight) -> [FocalLoss(gamma, alpha=0.25), utils/loss.py:35-60] -> [SlideLoss, :378-402] -> mean.
        auto_iou: the level's mean IoU when the reference passes it (:186-187,191-192), else SlideLoss's default 0.5."""
        if self.fl_gamma <= 0 and not self.slide:
            return F.binary_cross_entropy_with_logits(logits, target, pos_weight=torch.tensor([pw], device=logits.device))
        loss = F.binary_cross_entropy_with_logits(logits, target, pos_weight=torch.tensor([pw], device=logits.device), reduction='none')
        if self.fl_gamma > 0:
            prob = torch.sigmoid(logits)
            p_t = target * prob + (1 - target) * (1 - prob)
            loss = loss * ((target * 0.25 + (1 - target) * 0.75) * (1.0 - p_t) ** self.fl_gamma)
        if self.slide:
            auto_iou = 0.5 if auto_iou is None else float(auto_iou)
            if auto_iou < 0.2:
                auto_iou = 0.2
            b1 = target <= auto_iou - 0.1
            b2 = (target > (auto_iou - 0.1)) & (target < auto_iou)
            b3 = target >= auto_iou
            loss = loss * (1.0 * b1 + math.exp(1.0 - auto_iou) * b2 + torch.exp(-(target - 1.0)) * b3)
        return loss.mean()

    @staticmethod
    def _wasserstein(pred, target, eps=1e-7, constant=12.8):
        """utils/metrics.py:341-354, fed xywh boxes although it reads its columns as x1y1x2y2 - kept as the reference has it."""
        b1_x1, b1_y1, b1_x2, b1_y2 = pred.split(1, dim=-1)
        b2_x1, b2_y1, b2_x2, b2_y2 = target.split(1, dim=-1)
        w1, h1 = b1_x2 - b1_x1, b1_y2 - b1_y1 + eps
        w2, h2 = b2_x2 - b2_x1, b2_y2 - b2_y1 + eps
        cd = ((b1_x1 + b1_x2) / 2 - (b2_x1 + b2_x2) / 2).pow(2) + ((b1_y1 + b1_y2) / 2 - (b2_y1 + b2_y2) / 2).pow(2) + eps
        return torch.exp(-torch.sqrt(cd + ((w1 - w2).pow(2) + (h1 - h2).pow(2)) / 4) / constant)

    def __call__(self, p, targets):
        dev = targets.device
        lcls, lbox, lobj = (torch.zeros(1, device=dev) for _ in range(3))
        tcls, tbox, indices, anchors = self.build_targets(p, targets)
        for i, pi in enumerate(p):
            b, a, gj, gi = indices[i]
            tobj = torch.zeros_like(pi[..., 0])
            n = b.shape[0]
            if n:
                ps = pi[b, a, gj, gi]
                pxy = ps[:, :2].sigmoid() * 2 - 0.5
                pwh = (ps[:, 2:4].sigmoid() * 2) ** 2 * anchors[i]
                pbox = torch.cat((pxy, pwh), 1)
                iou = bbox_ciou_xywh(pbox.T, tbox[i])
                if self.nwd:                                                               # :162-169, iou_ratio = 0.5 (:148)
                    nwd = self._wasserstein(pbox, tbox[i], constant=self.nwd_constant).squeeze()
                    lbox = lbox + 0.5 * (1.0 - iou).mean() + 0.5 * (1.0 - nwd).mean()
                    iou = (iou.detach() * 0.5 + nwd.detach() * 0.5).clamp(0, 1).type(tobj.dtype)
                else:
                    lbox = lbox + (1.0 - iou).mean()
                    iou = iou.detach().clamp(0, 1).type(tobj.dtype)
                order = torch.argsort(iou)                                                 # :174-176
                b, a, gj, gi, iou = b[order], a[order], gj[order], gi[order], iou[order]
                tobj[b, a, gj, gi] = (1.0 - self.gr) + self.gr * iou                       # last write wins
                auto_iou = iou.mean()                                                      # :180
                if self.nc > 1:
                    t = torch.full_like(ps[:, 5:], self.cn)
                    t[range(n), tcls[i]] = self.cp
                    lcls = lcls + self._bce(ps[:, 5:], t, self.cls_pw, auto_iou)
            obji = self._bce(pi[..., 4], tobj, self.obj_pw, auto_iou if n else None)
            lobj = lobj + obji * self.balance[i]
            if self.autobalance:                                                           # :197-198
                self.balance[i] = self.balance[i] * 0.9999 + 0.0001 / obji.detach().item()
        if self.autobalance:                                                               # :200-201
            self.balance = [x / self.balance[self.ssi] for x in self.balance]
        lbox = lbox * self.hyp['box']
        lobj = lobj * self.hyp['obj']
        lcls = lcls * self.hyp['cls']
        bs = p[0].shape[0]
        return (lbox + lobj + lcls) * bs, torch.cat((lbox, lobj, lcls)).detach()

    def build_targets(self, p, targets):
        """Anchor matching (utils/loss.py:210-262). targets: (nt,6) = image, class, x, y, w, h (normalised)."""
        na, nt = self.na, targets.shape[0]
        dev = targets.device
        tcls, tbox, indices, anch = [], [], [], []
        gain = torch.ones(7, device=dev).long()
        ai = torch.arange(na, device=dev).float().view(na, 1).repeat(1, nt)
        targets = torch.cat((targets.repeat(na, 1, 1), ai[:, :, None]), 2)                 # na,nt,7
        g = 0.5
        off = torch.tensor([[0, 0], [1, 0], [0, 1], [-1, 0], [0, -1]], device=dev).float() * g
        for i in range(self.nl):
            anchors = self.anchors[i]
            gain[2:6] = torch.tensor(p[i].shape)[[3, 2, 3, 2]]
            t = targets * gain
            if nt:
                r = t[:, :, 4:6] / anchors[:, None]
                keep = torch.max(r, 1 / r).max(2)[0] < self.hyp['anchor_t']
                t = t[keep]
                gxy = t[:, 2:4]
                gxi = gain[[2, 3]] - gxy
                j, k = ((gxy % 1 < g) & (gxy > 1)).T
                l, m = ((gxi % 1 < g) & (gxi > 1)).T
                sel = torch.stack((torch.ones_like(j), j, k, l, m))
                t = t.repeat((5, 1, 1))[sel]
                offsets = (torch.zeros_like(gxy)[None] + off[:, None])[sel]
            else:
                t = targets[0]
                offsets = 0
            b, c = t[:, :2].long().T
            gxy, gwh = t[:, 2:4], t[:, 4:6]
            gij = (gxy - offsets).long()
            gi, gj = gij.T
            a = t[:, 6].long()
            indices.append((b, a, gj.clamp_(0, gain[3] - 1), gi.clamp_(0, gain[2] - 1)))
            tbox.append(torch.cat((gxy - gij, gwh), 1))
            anch.append(anchors[a])
            tcls.append(c)
        return tcls, tbox, indices, anch


# ----------------------------------------------------------------------------------------------
# Repulsion loss (utils/RepulsionLoss.py) - optional term, never called by ComputeLoss.__call__.

def pairwise_iou_xyxy(box1, box2):
    """utils/RepulsionLoss.py:5-24, 'xyxy' branch: zero where the boxes do not strictly overlap."""
    lt = torch.max(box1[:, None, :2], box2[:, :2])
    rb = torch.min(box1[:, None, 2:], box2[:, 2:])
    area1 = (box1[:, 2:] - box1[:, :2]).prod(1)
    area2 = (box2[:, 2:] - box2[:, :2]).prod(1)
    valid = (lt < rb).to(lt.dtype).prod(dim=2)
    inter = (rb - lt).prod(2) * valid
    return inter / (area1[:, None] + area2 - inter)


def iog(gt, pred):
    """Intersection over ground-truth area (utils/RepulsionLoss.py:27-36)."""
    iw = (torch.min(gt[:, 2], pred[:, 2]) - torch.max(gt[:, 0], pred[:, 0])).clamp(min=0)
    ih = (torch.min(gt[:, 3], pred[:, 3]) - torch.max(gt[:, 1], pred[:, 1])).clamp(min=0)
    garea = ((gt[:, 2] - gt[:, 0]) * (gt[:, 3] - gt[:, 1])).clamp(1e-6)
    return iw * ih / garea


def smooth_ln(x, sigma=0.5):
    """utils/RepulsionLoss.py:39-44."""
    return torch.where(x <= sigma, -torch.log(1 - x), (x - sigma) / (1 - sigma) - np.log(1 - sigma))


def repulsion_loss(pbox, gtbox, fg_mask, sigma_repgt=0.9, sigma_repbox=0, pnms=0, gtnms=0):
    """RepGT + RepBox (utils/RepulsionLoss.py:47-95) without the hard-coded .cuda() hops.

    pbox, gtbox: (B, A, 4) xyxy; fg_mask: (B, A) bool.  For positives j<=z the pred-pred IoU is zeroed
    (upper triangle incl. diagonal); pairs matched to the *same* gt box are zeroed in both matrices.
    """
    dev = pbox.device
    rep_gt, rep_box = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    pbox, gtbox = pbox.detach(), gtbox.detach()
    used = 0
    for idx in range(pbox.shape[0]):
        sel = fg_mask[idx].bool()
        if sel.sum() <= 0:
            continue
        pp, gp = pbox[idx][sel], gtbox[idx][sel]
        used += 1
        pg = pairwise_iou_xyxy(pp, gp)
        ppi = pairwise_iou_xyxy(pp, pp)
        n = pp.shape[0]
        same_gt = (gp[:, None, :] == gp[None, :, :]).all(-1)
        upper = torch.ones(n, n, dtype=torch.bool, device=dev).triu(0)
        ppi = ppi.masked_fill(upper | same_gt, 0)
        pg = pg.masked_fill(same_gt, 0)
        best, arg = pg.max(1)
        hit = best > gtnms
        if hit.sum() > 0:
            rep_gt = rep_gt + smooth_ln(iog(gp[arg[hit]], pp[hit]), sigma_repgt).mean()
        if (ppi > pnms).sum() > 0:
            rep_box = rep_box + smooth_ln(ppi, sigma_repbox).mean()
    return (rep_gt / used).squeeze(0), (rep_box / used).squeeze(0)
