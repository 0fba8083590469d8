"""Validation metrics on the device: mirrors val.py:50-71 `process_batch` and utils/metrics.py:21-74 `ap_per_class` over
libsomi_hip.so, so the per-image device->host copies of val.py:189 and the numpy post-processing disappear.

    correct = process_batch(predn, labelsn, iouv)                    # one image, like val.py:184
    corrects = process_batches(dets, labels, iouv)                   # a list of images in one launch
    p, r, ap, f1, ap_class = ap_per_class(tp, conf, pred_cls, target_cls)
"""
import torch

from . import _lib
from ._lib import check


def _stream():
    return torch.cuda.current_stream().cuda_stream


FITNESS_WEIGHTS = (0.1, 0.1, 0.1, 0.7)
"""utils/metrics.py:16 - this fork weights P, R, mAP@0.5, mAP@0.5:0.95 as 0.1 / 0.1 / 0.1 / 0.7 (upstream YOLOv5: 0, 0, 0.1, 0.9)."""


def fitness(x):
    """utils/metrics.py:15-18: the scalar `train.py:304-305` ranks checkpoints by; x: (n, >=4) rows of [P, R, mAP@0.5, mAP@0.5:0.95, ...]
    (numpy array or tensor) -> (n,)."""
    return (x[:, :4] * x.new_tensor(FITNESS_WEIGHTS)).sum(1) if isinstance(x, torch.Tensor) else (x[:, :4] * FITNESS_WEIGHTS).sum(1)


def process_batches(detections, labels, iouv):
    """detections: list of (N_b,6) x1,y1,x2,y2,conf,cls; labels: list of (M_b,5) cls,x1,y1,x2,y2 (GPU tensors, pixel units);
    iouv (T,) -> list of (N_b,T) bool tensors."""
    if len(detections) != len(labels):
        raise RuntimeError('process_batches: one label tensor per detection tensor')
    dev = iouv.device
    if dev.type != 'cuda':
        raise RuntimeError('somi_amd.metrics runs on GPU tensors only (no CPU fallback)')
    nd, nl = [int(d.shape[0]) for d in detections], [int(l.shape[0]) for l in labels]
    det = torch.cat([d.reshape(-1, 6) for d in detections]).float().contiguous() if sum(nd) else torch.zeros(0, 6, device=dev)
    lab = torch.cat([l.reshape(-1, 5) for l in labels]).float().contiguous() if sum(nl) else torch.zeros(0, 5, device=dev)
    doff = torch.tensor([0] + nd, dtype=torch.int32).cumsum(0).to(torch.int32).to(dev)
    loff = torch.tensor([0] + nl, dtype=torch.int32).cumsum(0).to(torch.int32).to(dev)
    T = int(iouv.numel())
    correct = torch.zeros(sum(nd), T, dtype=torch.uint8, device=dev)
    if sum(nd):
        check(_lib.lib().somi_val_match_f32(det.data_ptr(), doff.data_ptr(), lab.data_ptr(), loff.data_ptr(),
                                            iouv.float().contiguous().data_ptr(), T, len(nd), max(nd), max(nl + [0]),
                                            correct.data_ptr(), _stream()), 'val_match')
    return list(correct.bool().split(nd))


def process_batch(detections, labels, iouv):
    """val.py:50-71 for one image -> correct (N,T) bool on the device."""
    return process_batches([detections], [labels], iouv)[0]


def ap_per_class(tp, conf, pred_cls, target_cls, ncap=None):
    """utils/metrics.py:21-74 (plot=False) -> (p, r, ap, f1, classes): fp64 GPU tensors / int32 classes, in the reference's
    order.  Class ids must be non-negative integers (as floats or ints)."""
    dev = conf.device
    if dev.type != 'cuda':
        raise RuntimeError('somi_amd.metrics runs on GPU tensors only (no CPU fallback)')
    tp8 = tp.to(torch.uint8).contiguous()
    N, T = tp8.shape
    conf, pred_cls, target_cls = (t.float().contiguous() for t in (conf, pred_cls, target_cls))
    M = int(target_cls.numel())
    if ncap is None:
        ncap = 1 + int(max(float(pred_cls.max()) if N else 0.0, float(target_cls.max()) if M else 0.0))
    out_cls = torch.zeros(ncap, dtype=torch.int32, device=dev)
    out_n = torch.zeros(1, dtype=torch.int32, device=dev)
    ap = torch.zeros(ncap, T, dtype=torch.float64, device=dev)
    p, r, f1 = (torch.zeros(ncap, dtype=torch.float64, device=dev) for _ in range(3))
    L = _lib.lib()
    nbytes = L.somi_ap_per_class_workspace_bytes(N, T, ncap)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    check(L.somi_ap_per_class_f64(tp8.data_ptr(), conf.data_ptr(), pred_cls.data_ptr(), target_cls.data_ptr(), N, M, T, ncap,
                                  out_cls.data_ptr(), out_n.data_ptr(), ap.data_ptr(), p.data_ptr(), r.data_ptr(), f1.data_ptr(),
                                  ws.data_ptr(), nbytes, _stream()), 'ap_per_class')
    n = int(out_n.item())
    return p[:n], r[:n], ap[:n], f1[:n], out_cls[:n]


class ConfusionMatrix:
    """utils/metrics.py:98-142 on the device: `ConfusionMatrix(nc, conf=0.25, iou_thres=0.2)`, `.process_batch(detections, labels)`
    per image (val.py:186), `.matrix` -> (nc+1, nc+1) float64 numpy array [predicted, true] with index nc = background.  The counts
    stay in an int32 device tensor; reading `.matrix` is the only synchronisation.  `process_batches` takes whole batches."""

    def __init__(self, nc, conf=0.25, iou_thres=0.2, device='cuda:0'):
        self.nc, self.conf, self.iou_thres = int(nc), float(conf), float(iou_thres)
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('somi_amd.metrics runs on GPU tensors only (no CPU fallback)')
        self.counts = torch.zeros((self.nc + 1, self.nc + 1), dtype=torch.int32, device=self.device)

    def process_batches(self, detections, labels):
        if len(detections) != len(labels):
            raise RuntimeError('ConfusionMatrix: one label tensor per detection tensor')
        if not detections:
            return
        dev = self.device
        nd, nl = [int(d.shape[0]) for d in detections], [int(l.shape[0]) for l in labels]
        det = torch.cat([d.reshape(-1, 6) for d in detections]).float().contiguous().to(dev) if sum(nd) else torch.zeros(0, 6, device=dev)
        lab = torch.cat([l.reshape(-1, 5) for l in labels]).float().contiguous().to(dev) if sum(nl) else torch.zeros(0, 5, device=dev)
        doff = torch.tensor([0] + nd, dtype=torch.int32).cumsum(0).to(torch.int32).to(dev)
        loff = torch.tensor([0] + nl, dtype=torch.int32).cumsum(0).to(torch.int32).to(dev)
        with torch.cuda.device(dev):
            check(_lib.lib().somi_confusion_matrix_f32(det.data_ptr(), doff.data_ptr(), lab.data_ptr(), loff.data_ptr(), len(nd),
                                                       max(nd), max(nl), self.nc, self.conf, self.iou_thres, self.counts.data_ptr(),
                                                       _stream()), 'somi_confusion_matrix_f32')

    def process_batch(self, detections, labels):
        self.process_batches([detections], [labels])

    @property
    def matrix(self):
        return self.counts.cpu().numpy().astype('float64')
