"""Validation loop body of the reference (val.py:148-212) on the device path: forward -> NMS -> per-image matching -> AP.

    mp, mr, map50, map_, per_class = run(model, batches, conf_thres=0.001, iou_thres=0.6)

`batches` yields (imgs uint8 (B,3,H,W), targets (nt,6) [image, class, x, y, w, h normalised]) or, like the reference's
dataloader (`collate_fn`, utils/datasets.py:676-680), (imgs, targets, paths, shapes) with shapes[i] = (native (h, w), (ratio,
pad)) of the letterbox - predictions and labels are then mapped back to the native image with `scale_coords` (val.py:180-183).
Predictions and statistics stay on the GPU until the five result numbers are read.  Plots, json / txt dumps and the confusion
matrix are the reference's CPU tooling and stay there.
"""
import torch

from .metrics import ap_per_class, process_batches
from .nms import non_max_suppression


def xywh2xyxy(x):
    """utils/general.py:541-547."""
    y = x.clone()
    y[:, 0] = x[:, 0] - x[:, 2] / 2
    y[:, 1] = x[:, 1] - x[:, 3] / 2
    y[:, 2] = x[:, 0] + x[:, 2] / 2
    y[:, 3] = x[:, 1] + x[:, 3] / 2
    return y


def scale_coords(img1_shape, coords, img0_shape, ratio_pad=None):
    """Map xyxy boxes from the (letterboxed) network input `img1_shape` (h, w) back to the native image `img0_shape`, in place,
    and clip them to it (utils/general.py:602-628)."""
    if ratio_pad is None:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad = (img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2
    else:
        gain, pad = ratio_pad[0][0], ratio_pad[1]
    coords[:, [0, 2]] -= pad[0]
    coords[:, [1, 3]] -= pad[1]
    coords[:, :4] /= gain
    coords[:, 0].clamp_(0, img0_shape[1])
    coords[:, 1].clamp_(0, img0_shape[0])
    coords[:, 2].clamp_(0, img0_shape[1])
    coords[:, 3].clamp_(0, img0_shape[0])
    return coords


@torch.no_grad()
def run(model, batches, conf_thres=0.001, iou_thres=0.6, single_cls=False, device=None, confusion_matrix=None):
    """-> (mp, mr, map50, map, dict(p, r, ap50, ap, ap_class, seen, nt)); val.py:148-212 with plots / saving off.
    confusion_matrix: a `somi_amd.metrics.ConfusionMatrix` to fill as val.py:141,186 does when `plots` is on."""
    device = device or next(model.parameters()).device
    if device.type != 'cuda':
        raise RuntimeError('somi_amd.val runs on the MI355X only (no CPU fallback)')
    was_training = model.training
    model.eval()
    iouv = torch.linspace(0.5, 0.95, 10, device=device)
    niou = iouv.numel()
    seen, tps, confs, pclss, tclss = 0, [], [], [], []
    for batch in batches:
        imgs, targets = batch[0], batch[1]
        shapes = batch[3] if len(batch) > 3 else None
        imgs, targets = imgs.to(device), targets.to(device).float().clone()
        nb, _, height, width = imgs.shape
        out, _ = model(imgs)
        targets[:, 2:] *= torch.tensor([width, height, width, height], device=device, dtype=torch.float32)     # val.py:166
        out = non_max_suppression(out, conf_thres, iou_thres, multi_label=True, agnostic=single_cls)            # val.py:169
        labs, predn = [], []
        for si in range(nb):
            labels = targets[targets[:, 0] == si, 1:]
            tbox = xywh2xyxy(labels[:, 1:5]) if len(labels) else labels.new_zeros(0, 4)
            if single_cls and len(out[si]):
                out[si][:, 5] = 0
            pn = out[si].clone()
            if shapes is not None:                                # native-space boxes (val.py:180-183)
                scale_coords((height, width), pn[:, :4], shapes[si][0], shapes[si][1])
                scale_coords((height, width), tbox, shapes[si][0], shapes[si][1])
            labs.append(torch.cat((labels[:, 0:1], tbox), 1) if len(labels) else labels.new_zeros(0, 5))
            predn.append(pn)
            tclss.append(labels[:, 0])
        seen += nb
        corrects = process_batches(predn, labs, iouv)             # the whole batch in one launch (val.py:184 per image)
        if confusion_matrix is not None:                          # val.py:185-186 (`if plots:`), only images with labels AND predictions
            both = [i for i in range(nb) if len(predn[i]) and len(labs[i])]
            confusion_matrix.process_batches([predn[i] for i in both], [labs[i] for i in both])
        for pred, correct in zip(out, corrects):
            tps.append(correct)
            confs.append(pred[:, 4])
            pclss.append(pred[:, 5])
    tp = torch.cat(tps) if tps else torch.zeros(0, niou, dtype=torch.bool, device=device)
    conf, pcls, tcls = (torch.cat(x) if x else torch.zeros(0, device=device) for x in (confs, pclss, tclss))
    mp = mr = map50 = map_ = 0.0
    detail = dict(p=None, r=None, ap50=None, ap=None, ap_class=None, seen=seen, nt=int(tcls.numel()))
    if tp.numel() and bool(tp.any()):                             # val.py:201
        p, r, ap, f1, ap_class = ap_per_class(tp, conf, pcls, tcls)
        ap50, apm = ap[:, 0], ap.mean(1)
        mp, mr, map50, map_ = float(p.mean()), float(r.mean()), float(ap50.mean()), float(apm.mean())
        detail.update(p=p, r=r, ap50=ap50, ap=apm, ap_class=ap_class, f1=f1)
    if was_training:
        model.train()
    return mp, mr, map50, map_, detail
