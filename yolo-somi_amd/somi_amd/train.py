"""One training step with the reference's semantics (train.py:247-277) on the MI355X path.

    pred = model(imgs); loss, items = compute_loss(pred, targets); loss *= WORLD_SIZE; loss.backward()
    optimizer.step(); optimizer.zero_grad(); ema.update(model)

Everything arithmetic runs in libsomi_hip.so: forward in training mode (batch-norm statistics), the fused loss (value + gradient),
the hand-written reverse walk (conv dgrad / wgrad on MFMA, BN / attention / ODConv backward), the fused Adam + EMA update.
With more than one rank the flat gradient buffers are all-reduced (SUM) in buckets on a side stream while the backward walk is
still running (ddp.GradBuckets) - DDP's semantics without the wrapper.  fp32 throughout (no GradScaler: nothing to scale).
"""
import torch

from .ddp import GradBuckets, layer_offsets
from .loss import ComputeLoss
from .optim import build_optimizer


class TrainStep:
    def __init__(self, model, hyp, batch_size, dist=None, nbs=64, bucket_mb=48, accumulate=1):
        """accumulate: optimizer step every `accumulate` batches (train.py:121,252,272: max(round(nbs / total_batch), 1) in the
        reference loop; gradients simply keep accumulating in the flat buffers in between).  Default 1: every batch."""
        if not next(model.parameters()).is_cuda:
            raise RuntimeError('TrainStep runs on the MI355X only (no CPU fallback)')
        self.model, self.dist = model, dist
        self.world = dist.get_world_size() if dist is not None else 1
        model.hyp = hyp
        model.train()
        self.optimizer = build_optimizer(model, hyp, batch_size * self.world, nbs=nbs, ema=True)
        self.compute_loss = ComputeLoss(model)
        self.accumulate, self._since_step = max(int(accumulate), 1), 0
        self.buckets = None
        if self.world > 1:
            self.buckets = GradBuckets(self.optimizer.flat_grads, layer_offsets(model, self.optimizer), dist=dist,
                                       bucket_bytes=bucket_mb << 20)
            model.__dict__['_grad_hook'] = self.buckets.layer_done
            for buf in self.optimizer.flat_params:                # one set of initial weights (DDP broadcasts from rank 0)
                dist.broadcast(buf, src=0)
            self.optimizer.reset_ema()
            model.invalidate()

    def step(self, imgs, targets):
        """imgs: (B,3,H,W) uint8 on the GPU; targets (nt,6).  Returns (loss, loss_items) like train.py:265."""
        if self.buckets:
            self.buckets.reset()
        pred = self.model(imgs)
        loss, items = self.compute_loss(pred, targets)
        if self.world > 1:
            loss = loss * self.world                              # train.py:266-267
        loss.backward()
        if self.buckets:
            self.buckets.finish()
        self._since_step += 1
        if self._since_step >= self.accumulate:
            self.optimizer.step()                                 # Adam + EMA, one pass
            self.optimizer.zero_grad()
            self._since_step = 0
        return loss.detach(), items
