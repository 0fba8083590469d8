"""Data-parallel gradient exchange for the training path: one process per GPU, RCCL over xGMI through
torch.distributed (backend "nccl" on ROCm), the reference's DDP semantics without the DDP wrapper.

Reference (train.py:208-209,266-267): DDP averages gradients over ranks and the loss is multiplied by WORLD_SIZE, i.e. every
rank ends up with the SUM over ranks of the gradients of the UNSCALED per-rank losses.  The caller (train.TrainStep) therefore
back-propagates the unscaled loss and this module SUMs.  Here the gradients already live in the optimizer's flat per-group buffers
(optim.FusedAdamEMA.flat_grads); they are all-reduced (SUM) in buckets cut from the END of each buffer, because the reverse
layer walk finalises gradients from the last layer to the first.  A bucket is launched on a side HIP stream as soon as every
layer that owns a slice of it has finished its backward, so the exchange overlaps the rest of the backward pass; the optimizer
step waits for the side stream.  xGMI is point-to-point (7 links x ~153 GB/s per GPU): ~48 MB buckets keep each RCCL call large
enough to use all links while leaving several buckets to overlap (310 MB of fp32 gradients -> 7 buckets).
"""
import os

import torch

# SOMI_DDP_SINGLE_RANK=1: run the collectives even in a one-rank process group - a rehearsal of the RCCL call pattern (init,
# broadcast, bucketed all-reduce on the side stream, barrier) on a box with a single GPU; results are unchanged by it.
SINGLE_RANK_REHEARSAL = os.environ.get('SOMI_DDP_SINGLE_RANK') == '1'


class GradBuckets:
    def __init__(self, flat_grads, layer_start_offsets, dist=None, bucket_bytes=48 << 20, use_streams=True):
        """flat_grads: list of 1-D gradient buffers (one per parameter group).
        layer_start_offsets: list (per buffer) of dicts layer_index -> first element offset of that layer's parameters in the
        buffer (layers own contiguous, increasing ranges).  dist: torch.distributed module or None (single process)."""
        self.flat, self.dist = flat_grads, dist
        self.use_streams = use_streams and flat_grads[0].is_cuda
        self.comm_stream = torch.cuda.Stream() if self.use_streams else None
        self.handles = []
        self.buckets = []                                        # per buffer: list of (start, end) from the end backwards
        per = max(bucket_bytes // 4, 1)
        for buf in flat_grads:
            n, cuts = buf.numel(), []
            end = n
            while end > 0:
                start = max(end - per, 0)
                cuts.append((start, end))
                end = start
            self.buckets.append(cuts)
        self.layer_off = layer_start_offsets
        self.next = [0] * len(flat_grads)                        # next bucket to launch per buffer
        self.launched = []                                       # (buffer index, start, end) in launch order - for tests
        self.enabled = True                                      # False: layer_done() is a no-op (a non-stepping micro-batch)
        self.timing = False                                      # True: HIP events around every bucket on the side stream (overlap_stats)
        self._spans, self._bwd_end = [], None

    def reset(self):
        self.next = [0] * len(self.flat)
        self.handles, self.launched = [], []
        self._spans, self._bwd_end = [], None

    def _launch(self, bi, start, end):
        self.launched.append((bi, start, end))
        if self.dist is None or (self.dist.get_world_size() == 1 and not SINGLE_RANK_REHEARSAL):
            return
        view = self.flat[bi][start:end]
        if self.use_streams:
            ev = torch.cuda.Event()
            ev.record()                                           # gradients of this bucket are final on the compute stream
            self.comm_stream.wait_event(ev)
            with torch.cuda.stream(self.comm_stream):
                if self.timing:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                h = self.dist.all_reduce(view, op=self.dist.ReduceOp.SUM, async_op=True)
                self.handles.append(h)
                if self.timing:
                    # the collective runs on the backend's own stream: make the SIDE stream wait for its end before the closing event
                    # (stream-side wait, the host does not block), otherwise e1 would mark the enqueue, not the end of the exchange
                    h.wait()
                    e1.record()
                    self._spans.append((e0, e1))
        else:
            self.handles.append(self.dist.all_reduce(view, op=self.dist.ReduceOp.SUM, async_op=True))

    def layer_done(self, layer_index):
        """Call after layer `layer_index` (and every later layer) has accumulated its parameter gradients."""
        if not self.enabled:
            return
        for bi, cuts in enumerate(self.buckets):
            final_from = self.layer_off[bi].get(layer_index)
            if final_from is None:
                continue
            while self.next[bi] < len(cuts) and cuts[self.next[bi]][0] >= final_from:
                self._launch(bi, *cuts[self.next[bi]])
                self.next[bi] += 1

    def finish(self):
        """Launch whatever is left (layer 0 side) and make the compute stream wait for the exchange."""
        if self.timing and self.use_streams:
            self._bwd_end = torch.cuda.Event(enable_timing=True)  # device time at which the reverse walk's last kernel has run
            self._bwd_end.record()
        for bi, cuts in enumerate(self.buckets):
            while self.next[bi] < len(cuts):
                self._launch(bi, *cuts[self.next[bi]])
                self.next[bi] += 1
        for h in self.handles:
            h.wait()
        if self.use_streams:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        self.handles = []

    def broadcast_buffers(self, buf, src=0):
        """DDP's `broadcast_buffers` (train.py:208-209 wraps the model with the default broadcast_buffers=True): before every forward the
        module buffers - here ONE flat tensor holding every BatchNorm running mean / variance - are overwritten with rank `src`'s, so the
        running statistics every rank updates in that forward start from the same values and rank != 0 evaluates with rank 0's statistics.
        One collective on the side stream; the compute stream waits for it (the forward reads and updates the buffers)."""
        if self.dist is None or buf.numel() == 0 or (self.dist.get_world_size() == 1 and not SINGLE_RANK_REHEARSAL):
            return
        if self.use_streams:
            ev = torch.cuda.Event()
            ev.record()                                           # the previous step's updates of the buffers are complete on the compute stream
            self.comm_stream.wait_event(ev)
            with torch.cuda.stream(self.comm_stream):
                self.dist.broadcast(buf, src=src, async_op=True).wait()
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        else:
            self.dist.broadcast(buf, src=src)

    def overlap_stats(self):
        """After a step run with `timing = True` (and a device synchronise): (overlap_frac, exchange_ms, buckets) - the share of the
        side stream's all-reduce time that had already elapsed when the backward pass's last kernel finished on the compute stream, i.e.
        how much of the exchange the reverse walk actually hid (1.0 = all of it; 0.0 = the exchange ran behind the backward pass)."""
        if not self._spans or self._bwd_end is None:
            return None
        total = hidden = 0.0
        for e0, e1 in self._spans:
            dur = e0.elapsed_time(e1)
            before = e0.elapsed_time(self._bwd_end)              # ms from this bucket's start to the end of the backward pass
            total += dur
            hidden += min(max(before, 0.0), dur)
        return (hidden / total if total > 0 else 0.0), total, len(self._spans)

    def measure_exchange(self, iters=5, sync=None):
        """The whole bucketed exchange on its own (nothing to overlap with), `iters` times: seconds per exchange and the bytes
        one exchange reduces.  Collective: every rank has to call it.  The buffers are zeroed first (repeated SUMs of live
        gradients would overflow) - call it outside the training loop only."""
        import time
        sync = sync or (torch.cuda.synchronize if self.flat[0].is_cuda else (lambda: None))
        for buf in self.flat:
            buf.zero_()
        nbytes = sum(buf.numel() * buf.element_size() for buf in self.flat)
        secs = []
        for it in range(iters + 1):                               # first pass = warm-up (communicator / ring setup)
            self.reset()
            sync()
            if self.dist is not None and (self.dist.get_world_size() > 1 or SINGLE_RANK_REHEARSAL):
                self.dist.barrier()
            t0 = time.perf_counter()
            self.finish()
            sync()
            if it:
                secs.append(time.perf_counter() - t0)
        self.reset()
        return sum(secs) / len(secs), nbytes


def layer_offsets(model, optimizer):
    """For every flat group buffer: layer index -> smallest offset of that layer's (and all later layers') parameters.  Parameters are
    laid out in module order, so 'layer i and everything after it' is the tail of the buffer starting at that offset."""
    owner = {}
    for m in model.model:
        for p in m.parameters():
            owner[id(p)] = m.i
    res = []
    for st in optimizer._flat:
        first = {}
        for p, o in zip(st['p'].tensors, st['p'].offsets):
            li = owner.get(id(p))
            if li is not None:
                first[li] = min(first.get(li, o), o)
        # tail property: offset for layer i = min over layers >= i
        run, out = None, {}
        for li in sorted(first, reverse=True):
            run = first[li] if run is None else min(run, first[li])
            out[li] = run
        res.append(out)
    return res
