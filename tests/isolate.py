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


def oracle_alone(blk, rec, dtype, with_sums=False):
    """The oracle block on the captured tensors in `dtype`: -> (outputs, input gradients, {param: grad}, sums or None)."""
    from oracle.somi_ref.testing import AbsTermSums
    for p in blk.parameters():
        p.grad = None
    xs = [t.to(dtype).requires_grad_(True) for t in rec['x']]
    ctx = AbsTermSums(blk, squares=False) if with_sums else None
    if ctx:
        ctx.__enter__()
    out = blk(xs if rec['multi_in'] else xs[0])
    outs = list(out) if isinstance(out, (list, tuple)) else [out]
    keep = [(o, d.to(dtype)) for o, d in zip(outs, rec['dy']) if d is not None]
    torch.autograd.backward([o for o, _ in keep], [d for _, d in keep])
    if ctx:
        ctx.__exit__(None, None, None)
    grads = {n: p.grad.detach().clone() for n, p in blk.named_parameters() if p.grad is not None}
    return [o.detach() for o in outs], [x.grad for x in xs], grads, (ctx.sums if ctx else None)


def hip_alone(m, rec):
    """The HIP block on the same tensors: -> (outputs NCHW / raw, input gradients NCHW or None, {param: grad})."""
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
