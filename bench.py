#!/usr/bin/env python3
"""bench.py - YOLO-SOMI hot path on MI355X: images/s of the 640x640 inference step (forward + NMS), synthetic data.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched by
torch.distributed.run with one rank per GPU.  A step = one pass of the hot path over one batch of VisDrone-shaped
synthetic uint8 images already resident in HBM: ingest(/255) -> SOMI forward (yolov5l-SOMI, 77.5 M parameters, eval,
Conv+BN folded) -> decode -> NMS(conf 0.001, iou 0.6, multi_label).  Inference shards by image with no data-path
collective ("replicas only"): value = images all ranks processed / max-over-ranks time, weak scaling.
Rank 0 prints ONE JSON line with `roofline` (dominant kernel: the fp32-MFMA implicit-GEMM conv, timed live with HIP
events on the launch stream) and, at N=1, `cpu_baseline` (the CPU oracle on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'yolo-somi_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz
XGMI_LINKS, XGMI_LINK_GBPS = 7, 153  # per GPU: 7 point-to-point xGMI links of ~153 GB/s each (MI355X_MICROARCH.md)


def somi_cfg_full():
    """Layer table of models/modules/YOLO-SOMI.yaml (C2fEACBAM -> C2fCBAM) with the yaml's 16 anchor pairs."""
    from somi_amd.configs import somi_cfg, SOMI_ANCHORS
    return somi_cfg(1.0, 1.0, nc=10, anchors=SOMI_ANCHORS)


def synthetic_images(batch, size, seed, device):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, (batch, 3, size, size), generator=g, dtype=torch.uint8).to(device)


def cpu_baseline_train(size):
    """The oracle's training step (forward in train mode + ComputeLoss + backward + torch Adam) on the host cores, batch 2."""
    from oracle.somi_ref import Model as OracleModel
    from oracle.somi_ref.loss import ComputeLoss as OracleLoss
    from oracle.somi_ref.testing import fill_state, synthetic_batch, HYP_VISDRONE
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    m = fill_state(OracleModel(somi_cfg_full()), 1).train()
    m.hyp = dict(HYP_VISDRONE)
    crit = OracleLoss(m)
    opt = torch.optim.Adam(m.parameters(), lr=3e-4, betas=(0.843, 0.999))
    B = 2
    imgs, targets = synthetic_batch(B, size, seed=0)
    x = imgs.float() / 255
    n, t0 = 0, time.time()
    while True:
        loss, _ = crit(m(x), targets)
        loss.backward()
        opt.step()
        opt.zero_grad()
        n += 1
        if n >= 2 or time.time() - t0 > 30:
            break
    dt = time.time() - t0
    return {'value': round(B * n / dt, 3), 'unit': 'images/s', 'cores': cores, 'kind': 'port',
            'sample': f'{n} x (forward + loss + backward + Adam) of yolov5l-SOMI at batch {B}, {size}x{size}, torch CPU fp32, {cores} threads'}


def cpu_baseline(size, seconds_budget=25.0):
    """The oracle (CPU restatement of the reference path, kind 'port') timed on this box's host cores: forward + NMS."""
    from oracle.somi_ref import Model as OracleModel
    from oracle.somi_ref.nms import non_max_suppression as oracle_nms
    from oracle.somi_ref.testing import fill_state
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    m = fill_state(OracleModel(somi_cfg_full()), 1).eval().fuse()
    B = 2
    x = synthetic_images(B, size, 0, 'cpu').float() / 255
    with torch.no_grad():
        m(x)                                                    # warm-up (also builds the decode grids)
        n, t0 = 0, time.time()
        while True:
            z, _ = m(x)
            oracle_nms(z, 0.001, 0.6, multi_label=True)
            n += 1
            if time.time() - t0 > seconds_budget or n >= 10:
                break
        dt = time.time() - t0
    return {'value': round(B * n / dt, 3), 'unit': 'images/s', 'cores': cores, 'kind': 'port',
            'sample': f'{n} x (forward + NMS) of yolov5l-SOMI at batch {B}, {size}x{size}, torch CPU fp32, {cores} threads'}


def pmc_traffic(kernel_name):
    """HBM bytes per launch of `kernel_name` from the committed rocprofv3 PMC passes of this same command
    (profiles/traffic.json, written by tools/profile_summary.py: separate FETCH_SIZE / WRITE_SIZE runs, gfx950 x2 read
    correction).  PMC counters cannot be read live by the benchmarked process; None if no profile is committed."""
    path = os.path.join(ROOT, 'profiles', 'traffic.json')
    if not os.path.exists(path):
        return None
    key = kernel_name.replace(' ', '')
    for k, v in json.load(open(path)).get('kernels', {}).items():
        if key in k.replace(' ', ''):
            return v['hbm_bytes_per_launch']
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=32, help='images per GPU per step (BASELINE configs[1]: bs=32)')
    ap.add_argument('--size', type=int, default=640)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-nms', action='store_true')
    ap.add_argument('--no-infer', action='store_true', help='skip the side measurement of the inference rate (profiling runs)')
    ap.add_argument('--mode', choices=['train', 'infer'], default='train',
                    help='train: forward(train)+loss+backward+Adam+EMA step (BASELINE configs[1]); infer: forward+NMS')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (the product path has no CPU fallback)')
    backend = os.environ.get('SOMI_DIST_BACKEND', 'nccl')     # 'gloo' + several ranks on one GPU = rehearsal on a 1-GPU box only
    if backend != 'nccl':
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    rehearsal = world == 1 and os.environ.get('SOMI_DDP_SINGLE_RANK') == '1'    # one-rank RCCL group: exercises the N > 1 call pattern
    if world > 1 or rehearsal:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if rehearsal:
            os.environ.setdefault('MASTER_PORT', '29557'), os.environ.setdefault('RANK', '0'), os.environ.setdefault('WORLD_SIZE', '1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)      # RCCL over xGMI, one GPU per rank
        else:
            dist.init_process_group(backend)

    from somi_amd import ops
    from somi_amd.model import Model
    from somi_amd.nms import non_max_suppression

    torch.manual_seed(0)
    model = Model(somi_cfg_full())
    from somi_amd.configs import fill_state
    fill_state(model, 1)                                        # deterministic synthetic weights, BN stats randomised
    model = model.to(dev).eval()
    imgs = synthetic_images(args.batch, args.size, 1000 + rank, dev)

    def infer_step():
        with torch.no_grad():
            z, _ = model(imgs)
            if args.no_nms:
                return None
            return non_max_suppression(z, 0.001, 0.6, multi_label=True)

    # inference throughput (eval mode) is measured beside the training number: a few steps
    infer_ips = None
    if not (args.no_infer and args.mode == 'train'):
        for _ in range(2):
            infer_step()
        torch.cuda.synchronize()
        t_inf = time.time()
        for _ in range(3):
            infer_step()
        torch.cuda.synchronize()
        infer_ips = args.batch * 3 / (time.time() - t_inf)

    if args.mode == 'train':
        from somi_amd.configs import HYP_VISDRONE, synthetic_batch
        from somi_amd.train import TrainStep
        _, targets = synthetic_batch(args.batch, args.size, seed=1000 + rank)
        targets = targets.to(dev)
        trainer = TrainStep(model, dict(HYP_VISDRONE), args.batch, dist=dist)

        def step():
            return trainer.step(imgs, targets)
    else:
        step = infer_step

    from somi_amd.dist import timed_steps, whole_job_rate
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ops.PROFILE = prof = []                                     # per-launch HIP events around every conv launch
    dt = timed_steps(step, args.steps, dist=dist, sync=torch.cuda.synchronize, device=dev)   # barrier+sync both sides, MAX over ranks
    ops.PROFILE = None

    # the gradient exchange on its own (algorithmic and bus bandwidth against the xGMI links) - after the
    # timed region, collective over all ranks
    exchange = None
    if (world > 1 or rehearsal) and args.mode == 'train' and trainer.buckets is not None:
        secs, nbytes = trainer.buckets.measure_exchange(iters=5)
        alg = nbytes / secs / 1e9
        exchange = {'bytes': nbytes, 'buckets': sum(len(c) for c in trainer.buckets.buckets), 'ms': round(secs * 1e3, 3),
                    'algbw_GBps': round(alg, 1), 'busbw_GBps': round(alg * 2 * (world - 1) / world, 1),
                    'xgmi_peak_GBps_per_gpu': XGMI_LINKS * XGMI_LINK_GBPS, 'backend': backend,
                    'note': 'bucketed SUM all-reduce of the flat fp32 gradient buffers, not overlapped with anything'}

    if rank == 0:
        # dominant kernel = the conv tile variant with the largest total time
        by = {}
        for name, flops, e0, e1, _ in prof:
            d = by.setdefault(name, [0, 0.0, 0.0])
            d[0] += 1
            d[1] += flops
            d[2] += e0.elapsed_time(e1) * 1e-3
        name, (cnt, flops, secs) = max(by.items(), key=lambda kv: kv[1][2])
        all_flops, all_secs = sum(v[1] for v in by.values()), sum(v[2] for v in by.values())
        achieved = flops / secs / 1e12
        out = {
            'metric': (f'images/sec train (forward+loss+backward+Adam+EMA) @{args.size}, VisDrone-shaped synthetic, yolov5l-SOMI' if args.mode == 'train'
                       else f'images/sec infer (forward+NMS) @{args.size}, VisDrone-shaped synthetic, yolov5l-SOMI'),
            'value': round(whole_job_rate(args.batch, args.steps, world, dt), 2), 'unit': 'images/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': (f'yolov5l-SOMI training step: uint8 ingest + forward (batch-stat BN) + ComputeLoss + backward + '
                                    f'{"gradient all-reduce + " if world > 1 else ""}Adam + EMA, ' if args.mode == 'train' else
                                    f'yolov5l-SOMI inference step: uint8 ingest + forward + decode'
                                    f'{"" if args.no_nms else " + NMS(conf 0.001, iou 0.6, multi_label)"}, ') +
                                   f'{args.size}x{args.size}, batch {args.batch}/GPU (BASELINE configs[1] shape)',
                       'batch_per_gpu': args.batch, 'imgsz': args.size, 'params': 77537610,
                       'parallelism': (f'dp{world}' if args.mode == 'train' else f'replicas x{world}')},
            'infer_images_per_s_per_gpu': None if infer_ips is None else round(infer_ips, 2),
            'roofline': {'bound': 'mfma', 'kernel': name, 'achieved': round(achieved, 2), 'peak': F32_MFMA_PEAK_TFLOPS,
                         'unit': 'TFLOP/s', 'frac': round(achieved / F32_MFMA_PEAK_TFLOPS, 4), 'traffic': None,
                         'launches': cnt, 'avg_launch_us': round(secs / cnt * 1e6, 2),
                         'avg_launch_gflop': round(flops / cnt / 1e9, 3),
                         'all_conv_tflops': round(all_flops / all_secs / 1e12, 2),
                         'conv_share_of_step': round(all_secs / dt, 3)},
        }
        out['roofline']['traffic'] = pmc_traffic(name)
        if exchange:
            out['allreduce'] = exchange
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline_train(args.size) if args.mode == 'train' else cpu_baseline(args.size)
        print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
