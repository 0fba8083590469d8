"""Weight repacking for the NHWC kernels (host side, done once at model load)."""
import torch


def pad4(c):
    """Channel storage width: multiples of 4 always (16 B vectors); widths >= 64 that are not a multiple of 32 go up to the
    next multiple of 32 (177 -> 192, 98 -> 128 in the decoupled head) so the consumer conv takes the uniform-tap fast
    path and whole MFMA tiles; the extra channels are zeros (zero weight rows) and cost no extra tiles."""
    if c >= 64 and c % 32:
        return (c + 31) // 32 * 32
    return (c + 3) // 4 * 4


def pack_conv_weight(w, cin_pad=None, cout_pad=None):
    """(Cout,Cin,kh,kw) [or (S,Cout,Cin,kh,kw)] -> (.., Cout_pad, kh*kw*Cin_pad), k = (r*kw+q)*Cin_pad + c; pads are zero."""
    lead = w.shape[:-4]
    Cout, Cin, kh, kw = w.shape[-4:]
    cin_pad = pad4(Cin) if cin_pad is None else cin_pad
    cout_pad = Cout if cout_pad is None else cout_pad
    out = torch.zeros(*lead, cout_pad, kh, kw, cin_pad, dtype=torch.float32, device=w.device)
    out[..., :Cout, :, :, :Cin] = w.movedim(-3, -1)
    return out.reshape(*lead, cout_pad, kh * kw * cin_pad).contiguous()


def bn_fold(bn):
    """eval-mode BatchNorm as y = x*scale + shift (utils/torch_utils.py:212-219 uses the same two terms)."""
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    return scale, bn.bias - bn.running_mean * scale


def pack_dgrad_weight(w, cin_pad=None, cout_pad=None):
    """(Cout,Cin,kh,kw) -> [Cin_pad][kh*kw*Cout_pad], k = (r*kw+q)*Cout_pad + co (operand of somi_conv2d_dgrad_nhwc_f32)."""
    Cout, Cin, kh, kw = w.shape
    cin_pad = pad4(Cin) if cin_pad is None else cin_pad
    cout_pad = pad4(Cout) if cout_pad is None else cout_pad
    out = torch.zeros(cin_pad, kh, kw, cout_pad, dtype=torch.float32, device=w.device)
    out[:Cin, :, :, :Cout] = w.permute(1, 2, 3, 0)
    return out.reshape(cin_pad, kh * kw * cout_pad).contiguous()
