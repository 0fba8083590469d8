"""One training step with the reference's semantics (train.py:247-277) on the MI355X path.

    pred = model(imgs); loss, items = compute_loss(pred, targets); loss *= WORLD_SIZE; loss.backward()
    optimizer.step(); optimizer.zero_grad(); ema.update(model)

Gradient scale with N ranks.  The reference multiplies the loss by WORLD_SIZE (train.py:266-267) only to undo the MEAN that DDP's
all-reduce applies, so every rank steps on  sum_r dL_r/dw  (L_r = the unscaled loss of rank r's shard).  Here the exchange is a SUM
all-reduce, so the backward pass runs on the UNSCALED loss and the sum over ranks is exactly that gradient; only the returned
(logged) loss carries the WORLD_SIZE factor, as the reference's does.

Everything arithmetic runs in libsomi_hip.so: forward in training mode (batch-norm statistics), the fused loss (value + gradient),
the hand-written reverse walk (conv dgrad / wgrad on MFMA, BN / attention / ODConv backward), the fused Adam + EMA update.
With more than one rank the flat gradient buffers are all-reduced (SUM) in buckets on a side stream while the backward walk is
still running (ddp.GradBuckets) - DDP's semantics without the wrapper.  With gradient accumulation the exchange happens once, on
the micro-batch that steps the optimizer (the sum is linear, so this equals DDP re-averaging its already-averaged buffers on every
micro-batch, and moves 1/accumulate of the bytes).  fp32 throughout (no GradScaler: nothing to scale).
"""
import torch

from .ddp import GradBuckets, SINGLE_RANK_REHEARSAL, layer_offsets
from .loss import ComputeLoss
from .optim import build_optimizer


def one_cycle(y1=0.0, y2=1.0, steps=100):
    """Sinusoidal ramp from y1 to y2 over `steps` epochs (utils/general.py:462-468): the `lf` of train.py:146."""
    import math
    return lambda x: ((1 - math.cos(x * math.pi / steps)) / 2) * (y2 - y1) + y1


def warmup_lr(optimizer, ni, nw, epoch, lf, hyp, batch_size, nbs=64):
    """The warm-up of train.py:250-256 for global batch index `ni` (<= nw): every group's lr ramps linearly from 0 (biases - group
    2 - from hyp['warmup_bias_lr']) to initial_lr * lf(epoch); returns the accumulation count the reference would use at `ni`
    (ramping 1 -> nbs / batch_size).  Groups that carry a 'momentum' key (SGD) ramp it too; Adam's groups have none."""
    import numpy as np
    xi = [0, nw]
    accumulate = max(1, np.interp(ni, xi, [1, nbs / batch_size]).round())
    for j, x in enumerate(optimizer.param_groups):
        x['lr'] = float(np.interp(ni, xi, [hyp['warmup_bias_lr'] if j == 2 else 0.0, x['initial_lr'] * lf(epoch)]))
        if 'momentum' in x:
            x['momentum'] = float(np.interp(ni, xi, [hyp['warmup_momentum'], hyp['momentum']]))
    return int(accumulate)


def multi_scale_size(shape_hw, imgsz, gs, rng=None):
    """The batch size `--multi-scale` draws (train.py:257-261): sz uniform in [0.5, 1.5] imgsz on the stride grid, the batch resized so that its
    long side is sz, both sides rounded up to stride multiples.  -> (H, W) or None when the scale factor is 1.  rng: a random.Random."""
    import math
    import random
    rng = rng or random
    sz = rng.randrange(int(imgsz * 0.5), int(imgsz * 1.5 + gs)) // gs * gs
    sf = sz / max(shape_hw)
    if sf == 1:
        return None
    return tuple(math.ceil(x * sf / gs) * gs for x in shape_hw)


def scheduler_step(optimizer, epoch, lf):
    """lr_scheduler.LambdaLR(optimizer, lr_lambda=lf).step() (train.py:148,284-285): lr = initial_lr * lf(epoch) for every group."""
    for x in optimizer.param_groups:
        x['lr'] = x['initial_lr'] * lf(epoch)


class TrainStep:
    def __init__(self, model, hyp, batch_size, dist=None, nbs=64, bucket_mb=48, accumulate=1, adam=True, sync_bn=False, amp=None, multi_scale=False, imgsz=640,
                 broadcast_buffers=True):
        """accumulate: optimizer step every `accumulate` batches (train.py:121,252,272: max(round(nbs / total_batch), 1) in the
        reference loop; gradients simply keep accumulating in the flat buffers in between).  Default 1: every batch.
        sync_bn: --sync-bn (train.py:165-167, SyncBatchNorm.convert_sync_batchnorm): every BatchNorm layer takes its training
        statistics - forward mean / variance, backward sums - over the batches of all ranks (ops.SYNC_BN).
        amp: None / 'f32' (default: exact fp32 products, the path every parity claim is made on), 'bf16' or 'bf16x3' - the reference's GPU
        loop runs forward under amp.autocast (train.py:263): the conv family's products (forward, data and weight gradients) then go
        through the bf16 matrix instructions with fp32 accumulation (ops.CONV_PREC).  Tensors, BatchNorm statistics, the loss, the
        optimizer and the EMA stay fp32, so no GradScaler is needed (bf16 has fp32's exponent range).
        multi_scale: `--multi-scale` (train.py:257-262): every batch is resized (bilinear, align_corners=False) to a random size in
        [0.5, 1.5] imgsz on the stride grid before the forward pass (Python's `random`, as the reference draws it).
        broadcast_buffers: N > 1 only - DDP's default (train.py:208-209): rank 0's BatchNorm running statistics overwrite every rank's before
        each forward (one flat-buffer broadcast on the side stream, ddp.GradBuckets.broadcast_buffers).  False: the statistics are made equal
        once at construction and then follow each rank's own shard (rank 0's are the ones EMA / checkpoints use either way)."""
        if not next(model.parameters()).is_cuda:
            raise RuntimeError('TrainStep runs on the MI355X only (no CPU fallback)')
        self.model, self.dist = model, dist
        self.world = dist.get_world_size() if dist is not None else 1
        self.sync_bn = dist if (sync_bn and dist is not None) else None   # like the reference: only under DDP (RANK != -1)
        # a process group of its own for the per-layer statistics all-gathers: on the default group they would queue behind the gradient
        # buckets already in flight (one communicator = one issue order) and stall the backward pass until those finish
        self.sync_bn_group = dist.new_group() if self.sync_bn is not None else None
        from . import ops as _ops
        if amp not in _ops.PREC:
            raise ValueError(f'amp must be one of {sorted(k for k in _ops.PREC if k)} or None, got {amp!r}')
        self.amp = _ops.PREC[amp]
        self.multi_scale, self.imgsz = bool(multi_scale), int(imgsz)
        model.hyp = hyp
        model.train()
        self.optimizer = build_optimizer(model, hyp, batch_size * self.world, nbs=nbs, ema=True, adam=adam)
        self.compute_loss = ComputeLoss(model)
        self.accumulate, self._since_step = max(int(accumulate), 1), 0
        self.broadcast_buffers = bool(broadcast_buffers)
        self.buckets = None
        if self.world > 1 or (dist is not None and SINGLE_RANK_REHEARSAL):
            self.buckets = GradBuckets(self.optimizer.flat_grads, layer_offsets(model, self.optimizer), dist=dist,
                                       bucket_bytes=bucket_mb << 20)
            model.__dict__['_grad_hook'] = self.buckets.layer_done
            for buf in self.optimizer.flat_params:                # one set of initial weights (DDP broadcasts from rank 0)
                dist.broadcast(buf, src=0)
            # ... and of initial buffers (BatchNorm running statistics); step() repeats this before every forward when
            # broadcast_buffers is on (DDP's default, train.py:208) - rank 0's are the ones the EMA / checkpoints / validation use
            if self.optimizer.flat_buffers.numel():
                dist.broadcast(self.optimizer.flat_buffers, src=0)
            self.optimizer.reset_ema()
            model.invalidate()

    def step(self, imgs, targets, size=None):
        """imgs: (B,3,H,W) uint8 on the GPU; targets (nt,6).  Returns (loss, loss_items) like train.py:265.
        size: (H, W) to resize this batch to (what multi_scale draws by itself; tests pass it explicitly)."""
        stepping = self._since_step + 1 >= self.accumulate        # this micro-batch ends with optimizer.step()
        if self.buckets:
            self.buckets.reset()
            self.buckets.enabled = stepping                       # local accumulation only on the others (DDP's no_sync)
            if self.broadcast_buffers:
                self.buckets.broadcast_buffers(self.optimizer.flat_buffers)
        from . import ops
        ops.SYNC_BN, ops.SYNC_BN_GROUP = self.sync_bn, self.sync_bn_group
        ops.CONV_PREC = self.amp
        try:
            if size is None and self.multi_scale:
                size = multi_scale_size(imgs.shape[2:], self.imgsz, int(self.model.stride.max()))
            if size is not None and tuple(size) != tuple(imgs.shape[2:]):
                x4 = ops.image_to_nhwc4(imgs.contiguous(), scale=1.0 / 255.0 if imgs.dtype == torch.uint8 else 1.0)
                pred = self.model._forward_once(None, ingested=ops.resize_bilinear_nhwc4(x4, int(size[0]), int(size[1])))
            else:
                pred = self.model(imgs)
            loss, items = self.compute_loss(pred, targets)
            loss.backward()                                       # unscaled: the SUM all-reduce supplies the WORLD_SIZE factor
        finally:
            ops.SYNC_BN = ops.SYNC_BN_GROUP = None
            ops.CONV_PREC = 0
        if self.buckets and stepping:
            self.buckets.finish()
        self._since_step += 1
        if stepping:
            self.optimizer.step()                                 # Adam + EMA, one pass
            self.optimizer.zero_grad()
            self._since_step = 0
        return loss.detach() * self.world, items                  # the reported loss is scaled like train.py:266-267
