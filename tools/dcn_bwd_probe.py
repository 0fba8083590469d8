#!/usr/bin/env python3
"""Times the windowed DCNv3 backward alone (HIP events) at the two bench sites.  usage: dcn_bwd_probe.py [spread=0.7]
Environment switches: SOMI_DCN_SLAB=1 (round-3 form of backward B: staging slab + combine pass instead of the coloured direct adds),
SOMI_DCN_SLAB_MB (that slab's cap), SOMI_DCN_NEAR (0: no near pass)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'yolo-somi_amd'))
import torch  # noqa: E402

from somi_amd import ops  # noqa: E402
from somi_amd.dcnv3 import dcnv3_backward  # noqa: E402

spread = float(sys.argv[1]) if len(sys.argv) > 1 else 0.7
d = torch.device('cuda')
for N, H in ((32, 80), (32, 160)):
    g = torch.Generator(device='cuda').manual_seed(0)
    C, G, K = 256, 8, 9
    x = torch.randn(N, H, H, C, device=d, generator=g)
    off = torch.randn(N, H, H, G * K * 2, device=d, generator=g) * spread
    m = torch.softmax(torch.randn(N, H, H, G, K, device=d, generator=g), -1).reshape(N, H, H, G * K).contiguous()
    go = torch.randn(N, H, H, C, device=d, generator=g)
    args = (3, 3, 1, 1, 1, 1, 1, 1, G, C // G, 1.0)
    for _ in range(3):
        dcnv3_backward(x, off, m, *args, go, 256)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        dcnv3_backward(x, off, m, *args, go, 256)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(json.dumps({'form': 'slab + combine' if os.environ.get('SOMI_DCN_SLAB') == '1' else 'coloured (direct adds)', 'near': os.environ.get('SOMI_DCN_NEAR', '1'), 'shape': f'N{N} {H}x{H}',
                      'spread': spread, 'ms': round(ms, 4), 'far_taps': ops.dcn_overflow_taps()}), flush=True)
