#!/usr/bin/env python3
"""Per-kernel measurements on the MI355X (HIP events on the launch stream): DCNv3 forward/backward achieved HBM GB/s
against the algorithmic bytes of SURVEY.md section 8d, and the conv-backbone (layers 0-9, 640x640, batch 64) fp32-MFMA TFLOP/s
that BASELINE.json's north_star targets; `val`: the validation-metric kernels at VisDrone-val scale; `augment`: the input-pipeline
kernel at the training batch.  usage: kernel_bench.py [dcn] [backbone] [val] [augment]  - prints one JSON object per line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo-somi_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

from somi_amd import ops  # noqa: E402
from somi_amd.blocks import Act  # noqa: E402
from somi_amd.configs import somi_cfg, SOMI_ANCHORS, fill_state  # noqa: E402
from somi_amd.dcnv3 import dcnv3_forward, dcnv3_backward  # noqa: E402
from somi_amd.model import Model  # noqa: E402

HBM_PEAK = 8000.0      # GB/s spec (MI355X_MICROARCH.md); 6290 GB/s measured float4 copy
F32_PEAK = 157.3


def timeit(fn, warm=3, iters=10):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def dcn(N, H, C=256, G=8, k=3, spread=2.0):
    d = torch.device('cuda')
    g = torch.Generator(device='cuda').manual_seed(0)
    K = k * k
    x = torch.randn(N, H, H, C, device=d, generator=g)
    off = torch.randn(N, H, H, G * K * 2, device=d, generator=g) * spread
    m = torch.softmax(torch.randn(N, H, H, G, K, device=d, generator=g), -1).reshape(N, H, H, G * K).contiguous()
    go = torch.randn(N, H, H, C, device=d, generator=g)
    args = (k, k, 1, 1, 1, 1, 1, 1, G, C // G, 1.0)
    from somi_amd import ops
    tf = timeit(lambda: dcnv3_forward(x, off, m, *args, 256))
    tb = timeit(lambda: dcnv3_backward(x, off, m, *args, go, 256))
    over = ops.dcn_overflow_taps()
    os.environ['SOMI_DCN_GIN'] = 'exact'                         # grad_input by exact list sums instead of the MFMA product
    te = timeit(lambda: dcnv3_backward(x, off, m, *args, go, 256))
    os.environ.pop('SOMI_DCN_GIN')
    ops.DCN_DIRECT = True                                        # the one-kernel form with fp32 atomics into grad_input (the reference's)
    os.environ['SOMI_DCN_DIRECT'] = '1'                          # ... and the tiled gathers from L2 (the library reads this per call)
    td = timeit(lambda: dcnv3_backward(x, off, m, *args, go, 256))
    tft = timeit(lambda: dcnv3_forward(x, off, m, *args, 256))
    ops.DCN_DIRECT = False
    os.environ['SOMI_DCN_DIRECT'] = '0'
    px = N * H * H
    bf, bb = 4 * (2 * C + 3 * G * K) * px, 4 * (4 * C + 6 * G * K) * px
    for name, t, byt in (('dcnv3_fwd (windowed: taps from an LDS window)', tf, bf), ('dcnv3_fwd (tiled: taps from L2)', tft, bf),
                         ('dcnv3_bwd (windowed: A + B + C)', tb, bb), ('dcnv3_bwd (windowed, B as exact list sums)', te, bb),
                         ('dcnv3_bwd (direct, fp32 atomics)', td, bb)):
        rec = {'kernel': name, 'shape': f'N{N} {H}x{H} C{C} G{G} K{K} offsets~N(0,{spread})', 'ms': round(t * 1e3, 4),
               'algorithmic_GB': round(byt / 1e9, 4), 'achieved_GBps': round(byt / t / 1e9, 1), 'frac_of_8TBps': round(byt / t / 1e9 / HBM_PEAK, 4)}
        if 'bwd (windowed: A' in name:
            rec['taps_outside_window'] = over
            rec['taps_total'] = px * G * K * 4
        print(json.dumps(rec), flush=True)


def backbone(B=64, S=640):
    d = torch.device('cuda')
    model = fill_state(Model(somi_cfg(1.0, 1.0, anchors=SOMI_ANCHORS)), 1).to(d).eval()
    imgs = torch.randint(0, 256, (B, 3, S, S), dtype=torch.uint8, device=d)

    def run():
        with torch.no_grad():
            a = Act(ops.image_to_nhwc4(imgs), 0, 3)
            for m in model.model[:10]:
                a = m(a)
        return a
    run()
    ops.PROFILE = prof = []
    run()
    torch.cuda.synchronize()
    ops.PROFILE = None
    flops = sum(p[1] for p in prof)
    secs = sum(p[2].elapsed_time(p[3]) for p in prof) * 1e-3
    t = timeit(run, warm=1, iters=3)
    print(json.dumps({'kernel': 'conv backbone L0-9 (all conv launches)', 'shape': f'B{B} {S}x{S}', 'conv_launches': len(prof),
                      'algorithmic_TFLOP': round(flops / 1e12, 3), 'conv_ms': round(secs * 1e3, 2),
                      'achieved_TFLOPs': round(flops / secs / 1e12, 2), 'frac_of_f32_mfma_peak': round(flops / secs / 1e12 / F32_PEAK, 4),
                      'backbone_wall_ms_incl_attention_kernels': round(t * 1e3, 2)}), flush=True)
    by = {}
    for name, fl, e0, e1, _ in prof:
        v = by.setdefault(name, [0, 0.0, 0.0])
        v[0] += 1; v[1] += fl; v[2] += e0.elapsed_time(e1) * 1e-3
    for name, (c, fl, s) in sorted(by.items(), key=lambda kv: -kv[1][2]):
        print(json.dumps({'kernel': name, 'launches': c, 'ms': round(s * 1e3, 2), 'TFLOPs': round(fl / s / 1e12, 2)}), flush=True)


    if os.environ.get('SOMI_PER_LAUNCH'):
        for name, fl, e0, e1, shp in prof:
            t = e0.elapsed_time(e1) * 1e-3
            print(json.dumps({'launch': name.split('<')[1][:-1], 'B,H,W,Cin,Cout,k,s,ps': shp, 'us': round(t * 1e6, 1),
                              'TFLOPs': round(fl / t / 1e12, 1)}), flush=True)


def val_metrics(nimg=548, nc=10):
    """process_batch + ap_per_class at VisDrone-val scale (548 images, 300 detections and 20-90 labels each) on the device."""
    from somi_amd.metrics import ap_per_class, process_batches
    g = torch.Generator().manual_seed(0)
    dets, labs = [], []
    for _ in range(nimg):
        M, N = int(torch.randint(20, 90, (1,), generator=g)), 300
        lc, lwh = torch.rand(M, 2, generator=g) * 600 + 20, torch.rand(M, 2, generator=g) * 60 + 6
        lab = torch.cat((torch.randint(0, nc, (M, 1), generator=g).float(), lc - lwh / 2, lc + lwh / 2), 1)
        pick = torch.randint(0, M, (N,), generator=g)
        box = lab[pick, 1:] + (torch.rand(N, 4, generator=g) - 0.5) * lwh[pick].repeat(1, 2) * torch.rand(N, 1, generator=g)
        dets.append(torch.cat((box, torch.rand(N, 1, generator=g), lab[pick, :1]), 1))
        labs.append(lab)
    d = torch.device('cuda')
    dd, ll, iv = [x.to(d) for x in dets], [x.to(d) for x in labs], torch.linspace(0.5, 0.95, 10, device=d)
    tpd = torch.cat(process_batches(dd, ll, iv))
    cd, pd, td = torch.cat([x[:, 4] for x in dd]), torch.cat([x[:, 5] for x in dd]), torch.cat([x[:, 0] for x in ll])
    t_match = timeit(lambda: process_batches(dd, ll, iv), warm=1, iters=3)
    t_ap = timeit(lambda: ap_per_class(tpd, cd, pd, td, ncap=nc), warm=1, iters=3)
    print(json.dumps({'kernel': 'val metrics (process_batch + ap_per_class)', 'shape': f'{nimg} images x 300 detections, nc={nc}',
                      'match_ms_incl_host_concat': round(t_match * 1e3, 3), 'ap_per_class_ms': round(t_ap * 1e3, 3)}), flush=True)


def augment(B=32, S=640, nimg=256):
    """somi_augment_u8 at the training batch: mosaic + affine crop + HSV + flip (and with mixup on every sample) from an
    HBM-resident cache of VisDrone-shaped images (640 x 360).  Algorithmic bytes per output pixel: 3 read (each source
    pixel under the crop once; 6 with mixup) + 3 written."""
    import random
    import time
    import numpy as np
    from somi_amd.augment import DeviceImageCache, HYP_VISDRONE_AUGMENT
    rng = np.random.RandomState(0)
    imgs = [rng.randint(0, 256, (360, 640, 3)).astype(np.uint8) for _ in range(nimg)]
    labs = []
    for _ in range(nimg):
        k = int(rng.randint(20, 90))
        wh = rng.uniform(0.01, 0.1, (k, 2))
        labs.append(np.concatenate((rng.randint(0, 10, (k, 1)), rng.uniform(0, 1, (k, 2)) * (1 - wh) + wh / 2, wh), 1).astype(np.float32))
    for tag, over in (('mosaic+hsv+flip', dict(mixup=0.0)), ('mosaic+mixup+hsv+flip', dict(mixup=1.0))):
        ds = DeviceImageCache(imgs, labs, S, dict(HYP_VISDRONE_AUGMENT, **over))
        random.seed(0), np.random.seed(0)
        t0 = time.perf_counter()
        plans = [ds.plan(i)[0] for i in range(B)]
        t_plan = time.perf_counter() - t0
        t0 = time.perf_counter()
        recs = ds.upload(plans)
        torch.cuda.synchronize()
        t_up = time.perf_counter() - t0
        out = ds.launch(recs, B)
        t = timeit(lambda: ds.launch(recs, B, out), warm=3, iters=20)
        px = B * S * S
        alg = px * (9 if over['mixup'] else 6)
        print(json.dumps({'kernel': 'augment_kernel', 'case': tag, 'shape': f'B{B} {S}x{S} from {nimg} cached 360x640 images',
                          'us': round(t * 1e6, 1), 'Gpixel_per_s': round(px / t / 1e9, 2), 'alg_GBps': round(alg / t / 1e9, 1),
                          'frac_hbm_peak': round(alg / t / 1e9 / HBM_PEAK, 4), 'host_plan_ms_per_batch': round(t_plan * 1e3, 2),
                          'record_upload_ms': round(t_up * 1e3, 2), 'images_per_s_kernel_only': round(B / t, 0)}), flush=True)


if __name__ == '__main__':
    which = sys.argv[1:] or ['dcn', 'backbone', 'val', 'augment']
    if 'augment' in which:
        augment()
    if 'val' in which:
        val_metrics()
    if 'dcn' in which:
        dcn(32, 80, spread=0.7)                                  # what the bench graph's DCNv3 sites see (fill_state weights)
        dcn(32, 160, spread=0.7)
        dcn(32, 80, spread=0.3)
        dcn(32, 80, spread=2.0)                                  # a third of the taps beyond the window
    if 'backbone' in which:
        backbone()
