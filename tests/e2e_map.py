#!/usr/bin/env python3
"""Test infrastructure (uses the CPU oracle).  End-to-end check on the MI355X: TRAIN the small SOMI graph with the product path on a synthetic rectangles task until it
detects them, then evaluate the trained weights twice - product path (HIP forward, NMS, matching, AP) and CPU oracle (reference
restatement of the same steps) - and compare mAP@0.5 / mAP@0.5:0.95 (BASELINE.json: "mAP@0.5 parity")."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo-somi_amd')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def rect_batch(B, S, nc, seed):
    """uint8 images with 1-4 filled, class-coloured rectangles on noise; targets (nt,6) normalised xywh."""
    g = torch.Generator().manual_seed(seed)
    imgs = torch.randint(0, 40, (B, 3, S, S), generator=g, dtype=torch.uint8)
    tg = []
    for b in range(B):
        for _ in range(int(torch.randint(1, 5, (1,), generator=g))):
            c = int(torch.randint(0, nc, (1,), generator=g))
            w, h = (int(v) for v in torch.randint(S // 8, S // 3, (2,), generator=g))
            x0, y0 = int(torch.randint(0, S - w, (1,), generator=g)), int(torch.randint(0, S - h, (1,), generator=g))
            col = torch.tensor([60 + 60 * (c % 3), 60 + 90 * ((c // 3) % 3), 200 - 50 * (c % 4)], dtype=torch.uint8)
            imgs[b, :, y0:y0 + h, x0:x0 + w] = col[:, None, None]
            tg.append([b, c, (x0 + w / 2) / S, (y0 + h / 2) / S, w / S, h / S])
    return imgs, torch.tensor(tg, dtype=torch.float32)


def oracle_eval(state, cfg, batches, conf_thres, iou_thres):
    from oracle.somi_ref import Model as OModel
    from oracle.somi_ref.metrics import ap_per_class, process_batch
    from oracle.somi_ref.nms import non_max_suppression
    ref = OModel(cfg)
    ref.load_state_dict(state)
    ref.eval()
    iouv = torch.linspace(0.5, 0.95, 10)
    tps, confs, pcls, tcls = [], [], [], []
    for imgs, targets in batches:
        S = imgs.shape[-1]
        with torch.no_grad():
            z, _ = ref(imgs.float() / 255)
        out = non_max_suppression(z, conf_thres, iou_thres, multi_label=True)
        t = targets.clone()
        t[:, 2:] *= S
        for si, pred in enumerate(out):
            lab = t[t[:, 0] == si, 1:]
            box = lab[:, 1:5].clone()
            xy, wh = box[:, :2].clone(), box[:, 2:].clone()
            labn = torch.cat((lab[:, :1], xy - wh / 2, xy + wh / 2), 1)
            correct = process_batch(pred, labn, iouv) if len(pred) and len(lab) else torch.zeros(len(pred), 10, dtype=torch.bool)
            tps.append(correct); confs.append(pred[:, 4]); pcls.append(pred[:, 5]); tcls.append(lab[:, 0])
    tp = torch.cat(tps).numpy()
    if not tp.any():
        return 0.0, 0.0, 0.0, 0.0
    p, r, ap, f1, _ = ap_per_class(tp, torch.cat(confs).numpy(), torch.cat(pcls).numpy(), torch.cat(tcls).numpy())
    return float(p.mean()), float(r.mean()), float(ap[:, 0].mean()), float(ap.mean(1).mean())


def main(steps=300, S=128, B=16, nc=4, width=0.25, depth=0.33):
    from somi_amd import val as V
    from somi_amd.configs import HYP_VISDRONE, SOMI_ANCHORS, fill_state, somi_cfg
    from somi_amd.model import Model
    from somi_amd.train import TrainStep
    cfg = somi_cfg(width, depth, nc=nc, anchors=SOMI_ANCHORS)
    model = fill_state(Model(cfg), 1).cuda()
    with torch.no_grad():                                         # fill_state randomises BN statistics for parity tests; start clean
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.zero_(); m.running_var.fill_(1.0); m.weight.fill_(1.0); m.bias.zero_()
    hyp = dict(HYP_VISDRONE)
    tr = TrainStep(model, hyp, B)
    for g_ in tr.optimizer.param_groups:
        g_['lr'] = 2e-3
    train = [rect_batch(B, S, nc, 100 + i) for i in range(8)]
    t0 = time.time()
    first = last = None
    for it in range(steps):
        imgs, tg = train[it % len(train)]
        loss, items = tr.step(imgs.cuda(), tg.cuda())
        if it == 0:
            first = float(loss)
        last = float(loss)
    torch.cuda.synchronize()
    t_train = time.time() - t0
    # three validation batches (two seen in training, one not): the CPU oracle's pass over them is what this test's time goes to
    val_batches = [rect_batch(B, S, nc, 100 + i) for i in range(2)] + [rect_batch(B, S, nc, 900)]
    from somi_amd.metrics import ConfusionMatrix
    cm = ConfusionMatrix(nc)
    mp, mr, m50, m, det = V.run(model, val_batches, conf_thres=0.001, iou_thres=0.6, confusion_matrix=cm)
    mat = cm.matrix                                               # trained detector: the mass sits on the diagonal
    diag = float(np.trace(mat[:nc, :nc]) / max(mat[:, :nc].sum(), 1.0))
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    omp, omr, om50, om = oracle_eval(state, cfg, val_batches, 0.001, 0.6)
    res = {'width': width, 'depth': depth, 'imgsz': S, 'batch': B, 'train_steps': steps, 'train_seconds': round(t_train, 1), 'loss_first': round(first, 4), 'loss_last': round(last, 4),
           'product': {'P': mp, 'R': mr, 'mAP50': m50, 'mAP50_95': m}, 'oracle': {'P': omp, 'R': omr, 'mAP50': om50, 'mAP50_95': om},
           'abs_diff_mAP50': abs(m50 - om50), 'abs_diff_mAP50_95': abs(m - om),
           'confusion_diagonal_share_of_labels': round(diag, 4)}
    print(json.dumps(res))
    return res


if __name__ == '__main__':
    a = sys.argv[1:]
    main(*(int(v) for v in a[:4]), *(float(v) for v in a[4:6]))
