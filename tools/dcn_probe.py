"""One DCNv3 shape, forward + backward a few times (for counter passes: tools/pmc_dcn.sh).  usage: dcn_probe.py N H spread iters"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'yolo-somi_amd'))
import torch  # noqa: E402

from somi_amd.dcnv3 import dcnv3_backward, dcnv3_forward  # noqa: E402

N, H, spread, iters = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
C, G, k = 256, 8, 3
K = k * k
d = torch.device('cuda')
g = torch.Generator(device='cuda').manual_seed(0)
x = torch.randn(N, H, H, C, device=d, generator=g)
off = torch.randn(N, H, H, G * K * 2, device=d, generator=g) * spread
m = torch.softmax(torch.randn(N, H, H, G, K, device=d, generator=g), -1).reshape(N, H, H, G * K).contiguous()
go = torch.randn(N, H, H, C, device=d, generator=g)
args = (k, k, 1, 1, 1, 1, 1, 1, G, C // G, 1.0)
for _ in range(iters):
    dcnv3_forward(x, off, m, *args, 256)
    dcnv3_backward(x, off, m, *args, go, 256)
torch.cuda.synchronize()
