"""Oracle (test infrastructure): CPU restatement of the SOMI building blocks.

Every class keeps the reference's attribute names so that a reference ``state_dict``
loads unchanged; bodies are written from the reference's behaviour, cited per class.
All paths below are relative to /root/reference.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

__all__ = [
    'autopad', 'Conv', 'ChannelAttentionModule', 'SpatialAttentionModule', 'CBAMBottleneck',
    'C2fCBAM', 'SPPF', 'Swish', 'BiFPN', 'ODConv2d_3rd', 'ODConv_3rd', 'Residual', 'SEAM',
    'Decouple', 'DecoupledDetect', 'fuse_conv_and_bn', 'initialize_weights',
    'check_anchor_order', 'make_divisible',
    'Bottleneck', 'C3', 'SPP', 'Focus', 'Concat', 'Detect', 'DCNv3_YOLO',
]


def make_divisible(x, divisor):
    """utils/general.py:452-454."""
    return math.ceil(x / divisor) * divisor


def autopad(k, p=None, d=1):
    """'same' padding for kernel k / dilation d (models/common.py:43-51)."""
    if p is not None:
        return p
    eff = (lambda v: d * (v - 1) + 1) if d > 1 else (lambda v: v)
    return eff(k) // 2 if isinstance(k, int) else [eff(v) // 2 for v in k]


class Conv(nn.Module):
    """conv2d(bias=False) -> BatchNorm2d -> SiLU (models/common.py:53-70)."""

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, d=1, act=True):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, k, s, autopad(k, p, d), groups=g, dilation=d, bias=False)
        self.bn = nn.BatchNorm2d(c2)
        self.act = nn.SiLU() if act is True else (act if isinstance(act, nn.Module) else nn.Identity())

    def forward(self, x):
        y = self.conv(x)
        if hasattr(self, 'bn'):          # un-fused (models/common.py:64-66)
            y = self.bn(y)
        return self.act(y)               # fused form drops bn (models/common.py:68-70)


class ChannelAttentionModule(nn.Module):
    """sigmoid(MLP(GAP(x)) + MLP(GMP(x))), MLP = c -> c/r -> c with ReLU (models/common.py:339-358)."""

    def __init__(self, c1, reduction=16):
        super().__init__()
        mid = c1 // reduction
        self.shared_MLP = nn.Sequential(nn.Linear(c1, mid), nn.ReLU(), nn.Linear(mid, c1))

    def forward(self, x):
        a = self.shared_MLP(x.mean(dim=(2, 3)))
        m = self.shared_MLP(x.amax(dim=(2, 3)))
        return torch.sigmoid(a + m)[:, :, None, None]


class SpatialAttentionModule(nn.Module):
    """sigmoid(conv_kxk([mean_c(x), max_c(x)])) with bias (models/common.py:392-405)."""

    def __init__(self, kernel_size=7):
        super().__init__()
        assert kernel_size in (3, 5, 7)
        self.cv1 = nn.Conv2d(2, 1, kernel_size, padding=kernel_size // 2)

    def forward(self, x):
        stats = torch.cat([x.mean(dim=1, keepdim=True), x.amax(dim=1, keepdim=True)], dim=1)
        return torch.sigmoid(self.cv1(stats))


class CBAMBottleneck(nn.Module):
    """cv1 3x3 -> channel attention -> spatial attention -> cv2 3x3 (+x) (models/common.py:671-691)."""

    def __init__(self, c1, c2, shortcut=True, g=1, e=1.0, k=(3, 3), ratio=8, kernel_size=3):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, k[0], 1)
        self.cv2 = Conv(c_, c2, k[1], 1, g=1)
        self.add = shortcut and c1 == c2
        self.channel_attention = ChannelAttentionModule(c_, ratio)
        self.spatial_attention = SpatialAttentionModule(kernel_size)

    def forward(self, x):
        t = self.cv1(x)
        t = self.channel_attention(t) * t
        t = self.spatial_attention(t) * t
        t = self.cv2(t)
        return x + t if self.add else t


class C2fCBAM(nn.Module):
    """C2f with CBAM bottlenecks (models/common.py:2671-2695).

    cv1 1x1 -> two halves; the second half feeds a chain of n bottlenecks; all (2+n) pieces
    are concatenated and mixed by cv2 1x1.
    """

    def __init__(self, c1, c2, n=1, shortcut=False, g=1, e=0.5, kernel_size=7):
        super().__init__()
        self.c = int(c2 * e)
        self.cv1 = Conv(c1, 2 * self.c, 1, 1)
        self.cv2 = Conv((2 + n) * self.c, c2, 1)
        self.m = nn.ModuleList(
            CBAMBottleneck(self.c, self.c, shortcut, g, k=(3, 3), e=1.0, ratio=16, kernel_size=kernel_size)
            for _ in range(n))

    def forward(self, x):
        pieces = list(self.cv1(x).chunk(2, 1))
        for blk in self.m:
            pieces.append(blk(pieces[-1]))
        return self.cv2(torch.cat(pieces, 1))


class SPPF(nn.Module):
    """1x1 -> three chained 5x5/s1 max-pools -> cat(4) -> 1x1 (models/common.py:1846-1861)."""

    def __init__(self, c1, c2, k=5):
        super().__init__()
        c_ = c1 // 2
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_ * 4, c2, 1, 1)
        self.m = nn.MaxPool2d(kernel_size=k, stride=1, padding=k // 2)

    def forward(self, x):
        x = self.cv1(x)
        y1 = self.m(x)
        y2 = self.m(y1)
        return self.cv2(torch.cat([x, y1, y2, self.m(y2)], 1))


class Swish(nn.Module):
    """x * sigmoid(x) (models/common.py:8210-8215)."""

    def forward(self, x):
        return x * torch.sigmoid(x)


class BiFPN(nn.Module):
    """Learned weighted sum: w_i / (sum_j swish(w_j) + 1e-4) (models/common.py:3688-3704).

    Note the numerator is the raw weight, not swish(w) - kept as the reference has it.
    """

    def __init__(self, length):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(length, dtype=torch.float32), requires_grad=True)
        self.swish = Swish()
        self.epsilon = 0.0001

    def forward(self, xs):
        w = self.weight / (self.swish(self.weight).sum(dim=0) + self.epsilon)
        return torch.stack([w[i] * xs[i] for i in range(len(xs))], dim=0).sum(dim=0)


class ODConv2d_3rd(nn.Conv2d):
    """Omni-dimensional dynamic conv (models/common.py:4495-4624).

    K candidate kernels; per sample four attentions from a squeezed context
    (filter sigma(C_out), spatial sigma(kh*kw), channel sigma(C_in/g), kernel softmax(K)) build
    one weight tensor per sample, applied as a grouped conv with groups = B*g.
    """

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, K=4, r=1 / 16, save_parameters=False, padding_mode='zeros'):
        self.K, self.r, self.save_parameters = K, r, save_parameters
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                         padding_mode)
        del self.weight
        self.weight = nn.Parameter(torch.empty(K, out_channels, in_channels // groups, *self.kernel_size))
        if bias:
            del self.bias
            self.bias = nn.Parameter(torch.empty(K, out_channels))
        hidden = max(int(in_channels * r), 16)
        self.gap = nn.AdaptiveAvgPool2d(1)
        self.reduction = nn.Linear(in_channels, hidden)      # present (unused) in the reference: :4521
        self.fc = nn.Conv2d(in_channels, hidden, 1, bias=False)
        self.bn = nn.BatchNorm2d(hidden)
        self.act = nn.ReLU(inplace=True)
        kk = self.kernel_size[0] * self.kernel_size[1]
        self.fc_f = nn.Linear(hidden, out_channels)
        if not save_parameters or kk > 1:
            self.fc_s = nn.Linear(hidden, kk)
        if not save_parameters or in_channels // groups > 1:
            self.fc_c = nn.Linear(hidden, in_channels // groups)
        if not save_parameters or K > 1:
            self.fc_w = nn.Linear(hidden, K)
        self.reset_parameters()

    def reset_parameters(self):
        """models/common.py:4538-4543."""
        fan_out = self.kernel_size[0] * self.kernel_size[1] * self.out_channels // self.groups
        for i in range(self.K):
            self.weight.data[i].normal_(0, math.sqrt(2.0 / fan_out))
        if self.bias is not None:
            self.bias.data.zero_()

    def attentions(self, context):
        """The squeezed context and the four attention vectors (models/common.py:4557-4578)."""
        z = self.fc(self.gap(context))
        if z.size(0) > 1:                      # BN skipped for a single sample (:4562)
            z = self.bn(z)
        z = self.act(z.flatten(1))
        a_f = torch.sigmoid(self.fc_f(z))
        a_s = torch.sigmoid(self.fc_s(z)) if hasattr(self, 'fc_s') else None
        a_c = torch.sigmoid(self.fc_c(z)) if hasattr(self, 'fc_c') else None
        a_w = torch.softmax(self.fc_w(z), -1) if hasattr(self, 'fc_w') else None
        return a_f, a_s, a_c, a_w

    def get_weight_bias(self, context):
        B = context.shape[0]
        a_f, a_s, a_c, a_w = self.attentions(context)
        attn = a_f.view(B, 1, -1, 1, 1, 1)
        if a_s is not None:
            attn = attn * a_s.view(B, 1, 1, 1, *self.kernel_size)
        if a_c is not None:
            attn = attn * a_c.view(B, 1, 1, -1, 1, 1)
        if a_w is not None:
            attn = attn * a_w.view(B, -1, 1, 1, 1, 1)
        weight = (attn * self.weight).sum(1).view(-1, self.in_channels // self.groups, *self.kernel_size)
        bias = None
        if self.bias is not None:
            bias = (a_w @ self.bias if a_w is not None else self.bias.tile(B, 1)).view(-1)
        return weight, bias

    def forward(self, x, context=None):
        B, C, H, W = x.shape
        if C != self.in_channels:
            raise ValueError(f'Expected input{[B, C, H, W]} to have {self.in_channels} channels, '
                             f'but got {C} channels instead')
        weight, bias = self.get_weight_bias(x if context is None else context)
        y = F.conv2d(x.reshape(1, B * C, H, W), weight, bias, self.stride, self.padding, self.dilation,
                     B * self.groups)
        return y.view(B, self.out_channels, *y.shape[2:])


class ODConv_3rd(nn.Module):
    """ODConv2d_3rd -> BN -> SiLU (models/common.py:4638-4653)."""

    def __init__(self, c1, c2, k=1, s=1, kerNums=4, g=1, p=None, act=True):
        super().__init__()
        self.conv = ODConv2d_3rd(c1, c2, k, s, autopad(k, p), groups=g, K=kerNums)
        self.bn = nn.BatchNorm2d(c2)
        self.act = nn.SiLU() if act is True else (act if isinstance(act, nn.Module) else nn.Identity())

    def forward(self, x):
        return self.act(self.bn(self.conv(x)))


class Residual(nn.Module):
    """fn(x) + x (models/common.py:7183-7189)."""

    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, x):
        return self.fn(x) + x


class SEAM(nn.Module):
    """Separated-and-enhancement attention (models/common.py:8448-8505).

    depthwise 3x3 -> GELU -> BN, then n x [Residual(dw3x3 -> GELU -> BN) -> 1x1 -> GELU -> BN];
    GAP -> Linear(c, c/r) -> ReLU -> Linear -> Sigmoid; output x * exp(y).
    """

    def __init__(self, c1, c2, n, reduction=16):
        super().__init__()
        if c1 != c2:
            c2 = c1
        stages = []
        for _ in range(n):
            stages.append(nn.Sequential(
                Residual(nn.Sequential(nn.Conv2d(c2, c2, 3, 1, 1, groups=c2), nn.GELU(), nn.BatchNorm2d(c2))),
                nn.Conv2d(c2, c2, 1, 1, 0, groups=1), nn.GELU(), nn.BatchNorm2d(c2)))
        self.DCovN = nn.Sequential(nn.Conv2d(c1, c2, 3, 1, 1, groups=c1), nn.GELU(), nn.BatchNorm2d(c2), *stages)
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Sequential(nn.Linear(c2, c2 // reduction, bias=False), nn.ReLU(inplace=True),
                                nn.Linear(c2 // reduction, c2, bias=False), nn.Sigmoid())
        for m in self.modules():                 # :8492-8498
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight, gain=1)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        # initialize_layer(self.fc) in the reference is a no-op for a Sequential (:8500-8504)

    def forward(self, x):
        b, c = x.shape[:2]
        y = self.DCovN(x).mean(dim=(2, 3))
        y = self.fc(y).view(b, c, 1, 1)
        return x * torch.exp(y)


class Decouple(nn.Module):
    """Decoupled head for one level (models/yolo.py:1042-1073).

    a: 1x1 -> c_=min(c1,256); box/obj branch 3x3 -> 3x3 -> 1x1(na*5); cls branch 1x1 -> 1x1 -> 1x1(na*nc);
    outputs interleaved per anchor as [5 | nc].
    """

    def __init__(self, c1, nc=80, na=3):
        super().__init__()
        c_ = min(c1, 256)
        self.na, self.nc = na, nc
        self.a = Conv(c1, c_, 1)
        c = [int(v + na * 5) for v in (c_ - na * 5) * torch.linspace(1, 0, 4)]
        self.b1, self.b2, self.b3 = Conv(c_, c[1], 3), Conv(c[1], c[2], 3), nn.Conv2d(c[2], na * 5, 1)
        self.c1, self.c2, self.c3 = Conv(c_, c_, 1), Conv(c_, c_, 1), nn.Conv2d(c_, na * nc, 1)

    def forward(self, x):
        bs, _, ny, nx = x.shape
        x = self.a(x)
        b = self.b3(self.b2(self.b1(x)))
        c = self.c3(self.c2(self.c1(x)))
        return torch.cat((b.view(bs, self.na, 5, ny, nx), c.view(bs, self.na, self.nc, ny, nx)), 2).view(bs, -1, ny, nx)


class DecoupledDetect(nn.Module):
    """Anchor-based detection layer over Decouple heads (models/yolo.py:925-980)."""
    stride = None

    def __init__(self, nc=10, anchors=(), ch=(), inplace=False):
        super().__init__()
        self.nc, self.no = nc, nc + 5
        self.nl, self.na = len(anchors), len(anchors[0]) // 2
        self.grid = [torch.zeros(1)] * self.nl
        self.anchor_grid = [torch.zeros(1)] * self.nl
        self.register_buffer('anchors', torch.tensor(anchors).float().view(self.nl, -1, 2))
        self.m = nn.ModuleList(Decouple(c, self.nc, self.na) for c in ch)
        self.inplace = False

    def _make_grid(self, nx, ny, i):
        """Cell origins minus 0.5 and anchors in pixels (models/yolo.py:967-980)."""
        dev, dt = self.anchors[i].device, self.anchors[i].dtype
        yv, xv = torch.meshgrid(torch.arange(ny, device=dev, dtype=dt), torch.arange(nx, device=dev, dtype=dt),
                                indexing='ij')
        shape = (1, self.na, ny, nx, 2)
        grid = torch.stack((xv, yv), 2).expand(shape) - 0.5
        anchor_grid = (self.anchors[i] * self.stride[i]).view(1, self.na, 1, 1, 2).expand(shape)
        return grid, anchor_grid

    def forward(self, x):
        x = list(x)
        z = []
        for i in range(self.nl):
            t = self.m[i](x[i])
            bs, _, ny, nx = t.shape
            x[i] = t.view(bs, self.na, self.no, ny, nx).permute(0, 1, 3, 4, 2).contiguous()
            if not self.training:
                if self.grid[i].shape[2:4] != x[i].shape[2:4]:
                    self.grid[i], self.anchor_grid[i] = self._make_grid(nx, ny, i)
                y = x[i].sigmoid()
                xy = (y[..., 0:2] * 2 + self.grid[i]) * self.stride[i]
                wh = (y[..., 2:4] * 2) ** 2 * self.anchor_grid[i]
                z.append(torch.cat((xy, wh, y[..., 4:]), 4).view(bs, -1, self.no))
        return x if self.training else (torch.cat(z, 1), x)


# ---------------------------------------------------------------------------------------------- stock YOLOv5 module set
class Bottleneck(nn.Module):
    """cv1 (k[0]) -> cv2 (k[1]) with an identity shortcut when shapes allow (models/common.py:1494-1509)."""

    def __init__(self, c1, c2, shortcut=True, g=1, k=(3, 3), e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, k[0], 1)
        self.cv2 = Conv(c_, c2, k[1], 1, g=g)
        self.add = shortcut and c1 == c2

    def forward(self, x):
        y = self.cv2(self.cv1(x))
        return x + y if self.add else y


class C3(nn.Module):
    """CSP bottleneck with three 1x1 convs (models/common.py:1541-1565): cv3(cat(m(cv1(x)), cv2(x))); the inner bottlenecks are
    1x1 -> 3x3 with expansion 1 (kernel sizes are passed as tuples, :1558)."""

    def __init__(self, c1, c2, n=1, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c1, c_, 1, 1)
        self.cv3 = Conv(2 * c_, c2, 1)
        self.m = nn.Sequential(*(Bottleneck(c_, c_, shortcut, g, k=((1, 1), (3, 3)), e=1.0) for _ in range(n)))

    def forward(self, x):
        return self.cv3(torch.cat((self.m(self.cv1(x)), self.cv2(x)), dim=1))


class SPP(nn.Module):
    """1x1 -> parallel k x k / s1 max-pools -> cat -> 1x1 (models/common.py:1806-1826)."""

    def __init__(self, c1, c2, k=(5, 9, 13)):
        super().__init__()
        c_ = c1 // 2
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_ * (len(k) + 1), c2, 1, 1)
        self.m = nn.ModuleList([nn.MaxPool2d(kernel_size=x, stride=1, padding=x // 2) for x in k])

    def forward(self, x):
        x = self.cv1(x)
        return self.cv2(torch.cat([x] + [m(x) for m in self.m], 1))


class Focus(nn.Module):
    """Space-to-depth (row parity, column parity) = (0,0), (1,0), (0,1), (1,1) then Conv (models/common.py:1973-1997)."""

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, act=True):
        super().__init__()
        self.conv = Conv(c1 * 4, c2, k, s, p, g, act=act)

    def forward(self, x):
        return self.conv(torch.cat([x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]], 1))


class Concat(nn.Module):
    """torch.cat along `dimension` (models/common.py:2085-2097)."""

    def __init__(self, dimension=1):
        super().__init__()
        self.d = dimension

    def forward(self, x):
        return torch.cat(x, self.d)


class Detect(nn.Module):
    """The anchor head of stock YOLOv5 (models/yolo.py:46-109): one 1x1 conv per level producing na*(5+nc) channels.
    Eval decode: xy = (2*sigmoid - 0.5 + cell) * stride, wh = (2*sigmoid)^2 * anchor_px."""
    stride = None

    def __init__(self, nc=10, anchors=(), ch=(), inplace=False):
        super().__init__()
        self.nc, self.no = nc, nc + 5
        self.nl, self.na = len(anchors), len(anchors[0]) // 2
        self.grid = [torch.zeros(1)] * self.nl
        self.anchor_grid = [torch.zeros(1)] * self.nl
        self.register_buffer('anchors', torch.tensor(anchors).float().view(self.nl, -1, 2))
        self.m = nn.ModuleList(nn.Conv2d(x, self.no * self.na, 1) for x in ch)
        self.inplace = inplace

    def _make_grid(self, nx, ny, i):
        d = self.anchors[i].device
        yv, xv = torch.meshgrid(torch.arange(ny, device=d), torch.arange(nx, device=d), indexing='ij')
        grid = torch.stack((xv, yv), 2).expand((1, self.na, ny, nx, 2)).float()
        anchor_grid = (self.anchors[i].clone() * self.stride[i]).view((1, self.na, 1, 1, 2)).expand((1, self.na, ny, nx, 2)).float()
        return grid, anchor_grid

    def forward(self, x):
        x = list(x)
        z = []
        for i in range(self.nl):
            t = self.m[i](x[i])
            bs, _, ny, nx = t.shape
            x[i] = t.view(bs, self.na, self.no, ny, nx).permute(0, 1, 3, 4, 2).contiguous()
            if not self.training:
                if self.grid[i].shape[2:4] != x[i].shape[2:4]:
                    self.grid[i], self.anchor_grid[i] = self._make_grid(nx, ny, i)
                y = x[i].sigmoid()
                xy = (y[..., 0:2] * 2 - 0.5 + self.grid[i]) * self.stride[i]
                wh = (y[..., 2:4] * 2) ** 2 * self.anchor_grid[i]
                z.append(torch.cat((xy, wh, y[..., 4:]), -1).view(bs, -1, self.no))
        return x if self.training else (torch.cat(z, 1), x)


# ---------------------------------------------------------------------------------------------- DCNv3 wired into a graph
class DCNv3_YOLO(nn.Module):
    """The build's wiring of the reference's DCNv3 layer into an NCHW YOLO graph (the reference vendors models/ops_dcnv3 but wires
    it into no model, SURVEY fact 3 / section 7 "hard parts"): NCHW -> NHWC -> DCNv3 (modules/dcnv3.py:222-379) -> NCHW ->
    BatchNorm2d -> SiLU, channel-preserving.  `c` is nominal in a yaml (parse_model substitutes the incoming width)."""

    def __init__(self, c, k=3, s=1, g=4, offset_scale=1.0, center_feature_scale=False):
        super().__init__()
        from .dcnv3 import DCNv3
        self.dcnv3 = DCNv3(c, kernel_size=k, stride=s, pad=k // 2, group=g, offset_scale=offset_scale,
                           center_feature_scale=center_feature_scale)
        self.bn = nn.BatchNorm2d(c)
        self.act = nn.SiLU()

    def forward(self, x):
        return self.act(self.bn(self.dcnv3(x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)))


def fuse_conv_and_bn(conv, bn):
    """Fold an eval-mode BN into the preceding conv (utils/torch_utils.py:202-222).

    W' = diag(gamma / sqrt(var + eps)) W ;  b' = gamma (b - mean) / sqrt(var + eps) + beta.
    """
    fused = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding,
                      groups=conv.groups, bias=True).requires_grad_(False).to(conv.weight.device)
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    fused.weight.copy_((conv.weight.flatten(1) * scale[:, None]).view_as(fused.weight))
    b = torch.zeros(conv.out_channels, device=conv.weight.device) if conv.bias is None else conv.bias
    fused.bias.copy_(scale * b + bn.bias - scale * bn.running_mean)
    return fused


def initialize_weights(model):
    """BN eps / momentum the reference sets on every model (utils/torch_utils.py:165-174)."""
    for m in model.modules():
        if type(m) is nn.BatchNorm2d:
            m.eps, m.momentum = 1e-3, 0.03
        elif type(m) in (nn.Hardswish, nn.LeakyReLU, nn.ReLU, nn.ReLU6, nn.SiLU):
            m.inplace = True


def check_anchor_order(m):
    """Flip anchors if their area order disagrees with the stride order (utils/autoanchor.py:16-22)."""
    a = m.anchors.prod(-1).view(-1)
    if (a[-1] - a[0]).sign() != (m.stride[-1] - m.stride[0]).sign():
        m.anchors[:] = m.anchors.flip(0)
