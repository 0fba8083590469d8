"""Capture-and-isolate helpers shared by tests/test_train_gpu.py and tools/block_isolate.py (test infrastructure: the CPU oracle is the checker).

One whole-graph fp64 oracle pass records, for chosen top-level layers, their inputs and the gradients of their outputs (`capture`); the fp64
oracle block, the fp32 CPU oracle block (`oracle_alone`) and the HIP block (`hip_alone`) are then run ALONE on those tensors rounded to fp32 -
bit-identical inputs, output gradients and weights for all three - so whatever a block shows there is error it generates itself."""
import torch


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def capture(model64, run, layers=None):
    """Forward hooks on the top-level layers (all, or the indices in `layers`): inputs and output gradients of one pass, kept as fp32 CPU tensors."""
    cap, handles = {}, []

    def hook(m, inp, out):
        x = inp[0]
        xs = list(x) if isinstance(x, (list, tuple)) else [x]
        rec = cap[m.i] = dict(x=[t.detach().float() for t in xs], multi_in=isinstance(x, (list, tuple)), dy=None)
        outs = list(out) if isinstance(out, (list, tuple)) else [out]
        rec['multi_out'] = isinstance(out, (list, tuple))
        rec['dy'] = [None] * len(outs)
        for k, o in enumerate(outs):
            if o.requires_grad:
                o.register_hook(lambda g, rec=rec, k=k: rec['dy'].__setitem__(k, g.detach().float()))
    for m in model64.model:
        if layers is None or m.i in layers:
            handles.append(m.register_forward_hook(hook))
    run()
    for h in handles:
        h.remove()
    return cap


def hip_decisions(module):
    """The arg-max decisions the HIP path took in its LAST train-mode forward of `module` (a Model or any block), read from the contexts its
    blocks keep for their backward - call it between forward and backward.  -> the table oracle.somi_ref.testing.ForcedDecisions takes:
    names as in named_modules() (the oracle's modules carry the same names)."""
    import torch.nn.functional as F
    from somi_amd import blocks as B
    table = {}
    for name, mod in module.named_modules():
        ctx = mod.__dict__.get('_ctx')
        if ctx is None:
            continue
        if isinstance(mod, B.SpatialAttentionModule):              # (x, ca, stats, sa, w): channel of each pixel's maximum of ca * x
            x, ca = ctx[0], ctx[1]
            t = x.t[..., x.coff:x.coff + x.c]
            table[name] = (t * ca[:, None, None, :]).argmax(-1).cpu()
        elif isinstance(mod, B.ChannelAttentionModule):            # (x, avg, mx, ca, weights): pixel of each channel's spatial maximum
            x = ctx[0]
            t = x.t[..., x.coff:x.coff + x.c]
            table[name] = t.flatten(1, 2).argmax(1).cpu()
        elif isinstance(mod, B.SPPF):                              # the concat buffer: slice 0 is what the three windows pool
            c_ = mod.cv1.conv.out_channels
            ctx = ctx[0]                                           # (concat buffer, the kernel's per-level window codes)
            x1 = ctx.t[..., ctx.coff:ctx.coff + c_].permute(0, 3, 1, 2)
            Bn, C, H, W = x1.shape
            res = []
            for k in (5, 9, 13):                                   # first maximum in row-major order of the window, like the kernel and torch's pool
                p_ = k // 2
                a = F.pad(x1, (p_,) * 4, value=float('-inf')).unfold(2, k, 1).unfold(3, k, 1).flatten(4).argmax(-1)
                hh = torch.arange(H, device=a.device).view(1, 1, H, 1) + a // k - p_
                ww = torch.arange(W, device=a.device).view(1, 1, 1, W) + a % k - p_
                res.append((hh * W + ww).cpu())
            table[name] = res
    return table


def oracle_alone(blk, rec, dtype, with_sums=False, forced=None):
    """The oracle block on the captured tensors in `dtype`: -> (outputs, input gradients, {param: grad}, sums or None).
    forced: a hip_decisions table - the block is then evaluated at those arg-max decisions (ForcedDecisions)."""
    from oracle.somi_ref.testing import AbsTermSums, ForcedDecisions
    for p in blk.parameters():
        p.grad = None
    xs = [t.to(dtype).requires_grad_(True) for t in rec['x']]
    ctx = AbsTermSums(blk, squares=False) if with_sums else None
    if ctx:
        ctx.__enter__()
    if forced:
        with ForcedDecisions(blk, forced):
            out = blk(xs if rec['multi_in'] else xs[0])
    else:
        out = blk(xs if rec['multi_in'] else xs[0])
    outs = list(out) if isinstance(out, (list, tuple)) else [out]
    keep = [(o, d.to(dtype)) for o, d in zip(outs, rec['dy']) if d is not None]
    torch.autograd.backward([o for o, _ in keep], [d for _, d in keep])
    if ctx:
        ctx.__exit__(None, None, None)
    grads = {n: p.grad.detach().clone() for n, p in blk.named_parameters() if p.grad is not None}
    return [o.detach() for o in outs], [x.grad for x in xs], grads, (ctx.sums if ctx else None)


def hip_alone(m, rec, decisions=None):
    """The HIP block on the same tensors: -> (outputs NCHW / raw, input gradients NCHW or None, {param: grad}).
    decisions: a dict that receives the arg-max decisions of the forward (hip_decisions)."""
    from somi_amd import blocks as B
    from somi_amd import ops
    for p in m.parameters():
        p.grad = None
    first = m.i == 0
    if first:
        acts = [B.Act(ops.image_to_nhwc4(rec['x'][0].cuda().contiguous(), scale=1.0), 0, 3)]
    else:
        acts = [B.Act(nhwc(t).cuda()) for t in rec['x']]
    out = m(acts if rec['multi_in'] else acts[0])
    if decisions is not None:
        decisions.update(hip_decisions(m))
    det = isinstance(m, (B.DecoupledDetect, B.Detect))
    if det:
        outs = [r.detach().cpu() for r in out]
        dxs = m.backward([d.cuda() for d in rec['dy']])
    else:
        outs = [out.t[..., out.coff:out.coff + out.c].permute(0, 3, 1, 2).cpu()]
        d = B.Act(nhwc(rec['dy'][0]).cuda())
        if first:
            dxs = m.backward(d, need_dx=False)
        else:
            dxs = m.backward(d)
    torch.cuda.synchronize()
    if dxs is None:
        dxs = []
    dxs = dxs if isinstance(dxs, (list, tuple)) else [dxs]
    dxs = [a.t[..., a.coff:a.coff + a.c].permute(0, 3, 1, 2).cpu() for a in dxs]
    grads = {n: p.grad.detach().cpu() for n, p in m.named_parameters() if p.grad is not None}
    return outs, dxs, grads
