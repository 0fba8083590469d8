#!/usr/bin/env python3
"""bench.py - YOLO-SOMI hot path on MI355X: images/s of the 640x640 TRAINING step (default) or inference step, synthetic data.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched by
torch.distributed.run with one rank per GPU (the process refuses to run when WORLD_SIZE != --gpus).  A step = one pass of the hot
path over one batch of VisDrone-shaped synthetic uint8 images already resident in HBM:

  --mode train (default; BASELINE configs[1] at --batch 32): ingest (/255) -> train-mode forward (batch-statistic BN) -> ComputeLoss
      -> hand-written backward -> (N>1: bucketed gradient all-reduce over RCCL, overlapped) -> fused Adam + EMA
  --mode infer (configs[4] at --batch 128): ingest -> eval forward (Conv+BN folded) -> decode -> NMS(conf 0.001, iou 0.6, multi_label)

  --model somi-dcn (default): yolov5l-SOMI with its two DCNv3 sites ("yolov5l-SOMI (DCNv3 blocks)", BASELINE configs[1])
  --model somi:      the layer table of models/modules/YOLO-SOMI.yaml exactly (the reference wires DCNv3 into no yaml)
  --model yolov5s:   stock YOLOv5s, 80 classes (BASELINE configs[0]; use --batch 2)

Training shards data-parallel (weak scaling, per-rank batch fixed); inference shards by image with no collective ("replicas only").
value = images all ranks processed / max-over-ranks time.  Rank 0 prints ONE JSON line with
  roofline         the dominant kernel = the fp32-MFMA implicit-GEMM conv variant with the largest total time: algorithmic FLOPs of
                   its launches / HIP-event time of those launches inside the timed region (events on the launch stream);
  roofline_dcnv3   (DCN graph) the DCNv3 operator kernels inside the step against the HBM roofline: algorithmic bytes / event time;
  cpu_baseline     (N=1) the CPU oracle on the host cores, bounded sample, protocol of BASELINE.md section 3;
  allreduce        (N>1) the gradient exchange on its own, after the timed region.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC: what RCCL needs between the ranks' processes on this host driver
for p in (ROOT, os.path.join(ROOT, 'yolo-somi_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz
BF16_MFMA_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA peak (the 5 PF headline includes 2:1 sparsity)
HBM_PEAK_GBPS = 8000.0             # MI355X_MICROARCH.md: HBM3E ~8 TB/s
XGMI_LINKS, XGMI_LINK_GBPS = 7, 153  # per GPU: 7 point-to-point xGMI links of ~153 GB/s each (MI355X_MICROARCH.md)
MODELS = {'somi-dcn': 'yolov5l-SOMI (DCNv3 blocks)', 'somi': 'yolov5l-SOMI', 'yolov5s': 'yolov5s'}


def model_cfg(name, nc=None):
    """-> (layer table, number of classes).  somi: models/modules/YOLO-SOMI.yaml (C2fEACBAM -> C2fCBAM) with the yaml's 16 anchor pairs.
    nc: 10 (VisDrone) unless given - 3 for the UAVDT shape of BASELINE configs[3] (`--size 1280 --batch 8 --nc 3` is its per-GPU share)."""
    from somi_amd.configs import somi_cfg, yolov5_cfg, SOMI_ANCHORS
    if name == 'yolov5s':
        return yolov5_cfg(nc=nc or 80), nc or 80
    return somi_cfg(1.0, 1.0, nc=nc or 10, anchors=SOMI_ANCHORS, dcn=(name == 'somi-dcn')), nc or 10


def synthetic_images(batch, size, seed, device):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, (batch, 3, size, size), generator=g, dtype=torch.uint8).to(device)


def host_cpu():
    """(threads to use, CPU model): the physical cores of one socket, capped by what this process may actually run on - its affinity
    mask and its cgroup CPU quota (a 1-GPU box hands out a 16-CPU share of its 64-core socket; more threads than that only thrash)."""
    model, cores = 'unknown', set()
    try:
        phys = core = None
        for line in open('/proc/cpuinfo'):
            k, _, v = line.partition(':')
            k, v = k.strip(), v.strip()
            if k == 'model name':
                model = v
            elif k == 'physical id':
                phys = v
            elif k == 'core id':
                core = v
            elif not k and phys is not None:
                if phys == '0':
                    cores.add(core)
                phys = core = None
    except OSError:
        pass
    allowed = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    quota = None
    try:                                                        # cgroup v2, then v1
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        quota = None if q == 'max' else int(q) // int(per)
    except (OSError, ValueError):
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            quota = None if q <= 0 else q // int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        except (OSError, ValueError):
            pass
    n = min(len(cores) or allowed, allowed, quota or allowed, int(os.environ.get('SOMI_CPU_THREADS', '16')))
    return max(n, 1), model


def cpu_baseline(model_name, mode, size, nc=None, wbf=True):
    """The CPU oracle (restatement of the reference path, kind 'port') timed on this box's host cores.  Protocol of BASELINE.md section 3
    (utils/get_FPS.py:81-101 scaled down): warm-up, then a timed loop, wall clock around the loop; batch 8 for the SOMI graphs, 2 for
    yolov5s; threads = physical cores of one socket.  The protocol's 3 + 10 iterations of an 11 s SOMI training step would be minutes, so
    the loop is cut to a ~25 s budget but never below 3 timed iterations: when the warm-up shows that three steps at batch 8 do not fit,
    the batch drops to 4 (said in `sample`).  Per-iteration spread is reported beside the mean."""
    from oracle.somi_ref import Model as OracleModel
    from oracle.somi_ref.loss import ComputeLoss as OracleLoss
    from oracle.somi_ref.nms import non_max_suppression as oracle_nms
    from oracle.somi_ref.testing import fill_state, synthetic_batch, HYP_VISDRONE
    cores, cpu_model = host_cpu()
    torch.set_num_threads(cores)
    cfg, nc = model_cfg(model_name, nc)
    models = [fill_state(OracleModel(cfg), 1)]
    ensemble = mode == 'infer' and wbf
    if ensemble:
        models.append(fill_state(OracleModel(cfg), 2))

    def make_step(B):
        imgs, targets = synthetic_batch(B, size, nc=nc, seed=0)
        x = imgs.float() / 255
        if mode == 'train':
            m = models[0].train()
            m.hyp = dict(HYP_VISDRONE)
            crit = OracleLoss(m)
            opt = torch.optim.Adam(m.parameters(), lr=3e-4, betas=(0.843, 0.999))

            def step():
                loss, _ = crit(m(x), targets)
                loss.backward()
                opt.step()
                opt.zero_grad()
            return step, 'forward + ComputeLoss + backward + Adam'
        ms = [m.eval().fuse() for m in models]

        def step():
            with torch.no_grad():
                dets = [oracle_nms(m(x)[0], 0.001, 0.6, multi_label=True) for m in ms]
            if ensemble:
                from oracle.somi_ref.wbf import weighted_boxes_fusion as oracle_wbf
                for i in range(B):                                # wbf.py:44-68, image by image
                    oracle_wbf([(d[i][:, :4] / size).clamp(0, 1).tolist() for d in dets], [d[i][:, 4].tolist() for d in dets],
                               [d[i][:, 5].long().tolist() for d in dets], weights=None, iou_thr=0.67, skip_box_thr=0.01)
        return step, ('2 models x (forward + NMS) + WBF' if ensemble else 'forward + NMS')

    B = 2 if model_name == 'yolov5s' else 8
    warm = 3 if model_name == 'yolov5s' else 1
    step, what = make_step(B)
    t0 = time.time()
    for _ in range(warm):
        step()
    t_warm = (time.time() - t0) / warm
    note = ''
    if 3 * t_warm > 25.0 and B > 4:                               # three timed steps would not fit the budget: halve the batch
        B = 4
        step, what = make_step(B)
        step()
        note = f' (batch 8 took {t_warm:.1f} s per step in the warm-up: dropped to batch 4 so that 3 timed iterations fit)'
    times = []
    t0 = time.time()
    while len(times) < 10 and (len(times) < 3 or time.time() - t0 < 25.0):
        t1 = time.time()
        step()
        times.append(time.time() - t1)
    n, dt = len(times), sum(times)
    return {'value': round(B * n / dt, 3), 'unit': 'images/s', 'cores': cores, 'cpu_model': cpu_model, 'kind': 'port',
            'iterations': n, 'images_per_s_min': round(B / max(times), 3), 'images_per_s_max': round(B / min(times), 3),
            'sample': f'{warm} warm-up + {n} timed x ({what}) of {MODELS[model_name]} at batch {B}{note}, {size}x{size}, nc {nc}, torch CPU fp32, '
                      f"{cores} threads (this box's CPU share: min of one socket's physical cores, the cgroup quota and 16); BASELINE.md section 3 "
                      f'protocol (3 + 10 iterations) cut to a 25 s budget, never below 3 timed iterations'}


def pmc_traffic(kernel_name):
    """HBM bytes per launch of `kernel_name` from the committed rocprofv3 PMC passes of this same command
    (profiles/traffic.json, written by tools/profile_summary.py: separate FETCH_SIZE / WRITE_SIZE runs, gfx950 x2 read
    correction).  PMC counters cannot be read live by the benchmarked process; None if no profile is committed."""
    path = os.path.join(ROOT, 'profiles', 'traffic.json')
    if not os.path.exists(path):
        return None
    key = kernel_name.replace(' ', '').rstrip('>')               # (rocprofv3 prints trailing default template arguments: <...,true,0>)
    for k, v in json.load(open(path)).get('kernels', {}).items():
        if key in k.replace(' ', ''):
            return v['hbm_bytes_per_launch']
    return None


# the kernels behind the two DCNv3 operator entries that ops.PROFILE times.  The FIRST pattern of an entry is launched exactly once per
# operator call (forward = that one launch; backward = A once, then B once per colour of tiles + the far / near passes - or, slab form, B / C / D
# once per chunk of images)
DCN_OP_KERNELS = {'dcnv3_fwd_kernel': (r'dcnv3_win_kernel<\d+,0>|dcnv3_fwd_kernel',),
                  'dcnv3_bwd_kernel': (r'dcnv3_win_kernel<\d+,1>|dcnv3_bwd_om_kernel', r'dcnv3_bwd_gin_(mfma_)?kernel', r'dcnv3_bwd_combine_kernel',
                                       r'dcnv3_bwd_near_kernel', r'dcnv3_bwd_far_kernel')}


def pmc_traffic_op(op):
    """HBM bytes per OPERATOR CALL of one DCNv3 entry from the committed counter passes: the bytes of every launch of the entry's kernels
    (both DCN sites, every chunk) divided by the number of operator calls - the average over the step's sites, like `avg_launch_MB`, the
    algorithmic figure it is compared with.  None when the committed profile lacks the entry's first kernel."""
    import re
    path = os.path.join(ROOT, 'profiles', 'traffic.json')
    if not os.path.exists(path) or op not in DCN_OP_KERNELS:
        return None
    kern = {k.replace(' ', ''): v for k, v in json.load(open(path)).get('kernels', {}).items()}
    pats = DCN_OP_KERNELS[op]
    calls = sum(v['launches_sampled'] for k, v in kern.items() if re.search(pats[0], k))
    if not calls:
        return None
    total = sum(v['hbm_bytes_per_launch'] * v['launches_sampled'] for k, v in kern.items() if any(re.search(p_, k) for p_ in pats))
    return round(total / calls)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)       # >= 20 timed iterations after >= 10 warm-ups
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--settle-max', type=int, default=30, help='un-timed steps before the warm-up until the step time is steady (0: none)')
    ap.add_argument('--batch', type=int, default=None, help='images per GPU per step (default 32: BASELINE configs[1]; 2 for yolov5s)')
    ap.add_argument('--size', type=int, default=640)
    ap.add_argument('--nc', type=int, default=None, help='classes (default 10, yolov5s 80; 3 = UAVDT, BASELINE configs[3])')
    ap.add_argument('--model', choices=list(MODELS), default='somi-dcn')
    ap.add_argument('--no-dcn', action='store_true', help='same as --model somi')
    ap.add_argument('--sync-bn', action='store_true', help='N > 1: BatchNorm statistics over all ranks (train.py:165-167; off in the reference by default)')
    ap.add_argument('--amp', choices=['bf16', 'bf16x3'], default=None,
                    help='opt-in reduced precision of the conv products (train.py:263 autocast): bf16 operands, or bf16x3 split; fp32 accumulate, '
                         'fp32 tensors / BN / loss / optimizer.  The default (and the headline) is exact fp32')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-nms', action='store_true')
    ap.add_argument('--no-wbf', action='store_true', help='--mode infer: one model, forward + NMS only (no second model, no fusion step)')
    ap.add_argument('--no-infer', action='store_true', help='skip the side measurement of the inference rate (profiling runs)')
    ap.add_argument('--mode', choices=['train', 'infer'], default='train',
                    help='train: forward(train)+loss+backward+Adam+EMA step (BASELINE configs[1]); infer: forward+NMS (configs[4])')
    args = ap.parse_args()
    if args.no_dcn and args.model == 'somi-dcn':
        args.model = 'somi'
    if args.batch is None:
        args.batch = 2 if args.model == 'yolov5s' else 32

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:                                      # a mis-launch must not print an N=1 number under --gpus 8
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 with python -m torch.distributed.run '
                         f'--nproc-per-node {args.gpus} ... bench.py --gpus {args.gpus}')
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
        assert dist.get_world_size() == world

    from somi_amd import ops
    from somi_amd.model import Model
    from somi_amd.nms import non_max_suppression

    torch.manual_seed(0)
    cfg, nc = model_cfg(args.model, args.nc)
    model = Model(cfg)
    nparams = sum(p.numel() for p in model.parameters())
    from somi_amd.configs import fill_state
    fill_state(model, 1)                                        # deterministic synthetic weights, BN stats randomised
    model = model.to(dev).eval()
    imgs = synthetic_images(args.batch, args.size, 1000 + rank, dev)

    from somi_amd.nms import non_max_suppression_raw
    from somi_amd.wbf import weighted_boxes_fusion_batch
    ensemble = args.mode == 'infer' and not args.no_wbf and not args.no_nms
    model2 = None
    if ensemble:                                                # configs[4]: "fused NMS + WBF" = wbf.py over the label sets of several models
        model2 = fill_state(Model(cfg), 2).to(dev).eval()     # the second synthetic "model": same graph, other weights

    def ensemble_step(m1, m2, batch_imgs):
        """configs[4]: two models x (forward + decode + NMS) + device WBF of their detections, image by image (val.py:148-189, wbf.py:55-68)."""
        with torch.no_grad():
            z, _ = m1(batch_imgs)
            d1 = non_max_suppression_raw(z, 0.001, 0.6, multi_label=True)          # val.sh:1 thresholds; rows stay on the device
            z2, _ = m2(batch_imgs)
            d2 = non_max_suppression_raw(z2, 0.001, 0.6, multi_label=True)
            return weighted_boxes_fusion_batch([d1[0], d2[0]], [d1[1], d2[1]], (args.size, args.size), weights=None,
                                               iou_thr=0.67, skip_box_thr=0.01)       # wbf.py:34-35,68

    def infer_step():
        with torch.no_grad():
            if ensemble:
                return ensemble_step(model, model2, imgs)
            z, _ = model(imgs)
            return None if args.no_nms else non_max_suppression(z, 0.001, 0.6, multi_label=True)

    def single_step():                                          # one model: forward + decode + NMS
        with torch.no_grad():
            z, _ = model(imgs)
            return None if args.no_nms else non_max_suppression(z, 0.001, 0.6, multi_label=True)

    # single-model inference throughput (eval mode) is measured beside the headline number: a few steps
    infer_ips = None
    if not (args.no_infer and args.mode == 'train'):
        for _ in range(2):
            single_step()
        torch.cuda.synchronize()
        t_inf = time.time()
        for _ in range(3):
            single_step()
        torch.cuda.synchronize()
        infer_ips = args.batch * 3 / (time.time() - t_inf)

    if args.mode == 'train':
        from somi_amd.configs import HYP_VISDRONE, synthetic_batch
        from somi_amd.train import TrainStep
        _, targets = synthetic_batch(args.batch, args.size, nc=nc, seed=1000 + rank)
        targets = targets.to(dev)
        trainer = TrainStep(model, dict(HYP_VISDRONE), args.batch, dist=dist, sync_bn=args.sync_bn, amp=args.amp)

        def step():
            return trainer.step(imgs, targets)
    else:
        step = infer_step

    from somi_amd.dist import timed_steps, whole_job_rate
    if args.mode == 'infer':
        ops.CONV_PREC = ops.PREC[args.amp]                          # (training: TrainStep sets it around its forward + backward)
    # settle phase, before the W warm-up steps and disclosed in the line (`settle`): un-timed steps until the step time has stopped moving (the mean of
    # the last three within 2 % of the three before; 6 ... --settle-max steps).  One of round 4's default runs on a fresh box measured 402 ms per step
    # where every later run on the same box measured 329: whatever ramps up on a fresh box (clocks, the host's caches) is not what a training run of
    # thousands of steps sees.  With several ranks the decision is taken on the MAX over ranks, so every rank runs the same number of steps.
    settle = None
    if args.settle_max > 0:
        hist = []
        while len(hist) < args.settle_max:
            torch.cuda.synchronize()
            t0 = time.time()
            step()
            torch.cuda.synchronize()
            ms = torch.tensor([(time.time() - t0) * 1e3], device=dev)
            if world > 1:
                dist.all_reduce(ms, op=dist.ReduceOp.MAX)
            hist.append(ms.item())
            if len(hist) >= 6 and abs(sum(hist[-3:]) - sum(hist[-6:-3])) <= 0.02 * sum(hist[-6:-3]):
                break
        settle = {'steps': len(hist), 'first_ms': round(hist[0], 1), 'last_ms': round(hist[-1], 1)}
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ops.PROFILE = prof = []                                     # per-launch HIP events around every conv / DCNv3 launch
    dt = timed_steps(step, args.steps, dist=dist, sync=torch.cuda.synchronize, device=dev)   # barrier+sync both sides, MAX over ranks
    ops.PROFILE = None

    # the gradient exchange on its own (algorithmic and bus bandwidth against the xGMI links) - after the
    # timed region, collective over all ranks
    exchange = None
    if (world > 1 or rehearsal) and args.mode == 'train' and trainer.buckets is not None:
        # one more, instrumented, step: HIP events around every bucket on the side stream against the end of the backward pass on the
        # compute stream - whether the side-stream overlap works, not only that it ran
        trainer.buckets.timing = True
        step()
        torch.cuda.synchronize()
        ov = trainer.buckets.overlap_stats()
        trainer.buckets.timing = False
        secs, nbytes = trainer.buckets.measure_exchange(iters=5)
        alg = nbytes / secs / 1e9
        exchange = {'bytes': nbytes, 'buckets': sum(len(c) for c in trainer.buckets.buckets), 'ms': round(secs * 1e3, 3),
                    'overlap_frac': None if ov is None else round(ov[0], 4), 'in_step_exchange_ms': None if ov is None else round(ov[1], 3),
                    'algbw_GBps': round(alg, 1), 'busbw_GBps': round(alg * 2 * (world - 1) / world, 1),
                    'xgmi_peak_GBps_per_gpu': XGMI_LINKS * XGMI_LINK_GBPS, 'backend': backend, 'ranks': dist.get_world_size(),
                    'note': 'bucketed SUM all-reduce of the flat fp32 gradient buffers, not overlapped with anything'}

    # BASELINE's metric is "train+infer": after the timed training region of the DEFAULT shape (one GPU), the configs[4] pipeline - batch 128,
    # two synthetic models x (forward + NMS) + device WBF - runs 2 warm + 5 timed steps with the trained weights in eval mode, so the one line
    # the driver records covers both halves.  Inference shards by image with no collective: replicas only, so N = 1 only.
    configs4 = None
    if (args.mode == 'train' and world == 1 and not rehearsal and not args.no_infer and args.model != 'yolov5s' and args.size == 640 and nc == 10
            and args.amp is None):
        torch.cuda.empty_cache()
        model.eval()
        m2 = fill_state(Model(cfg), 2).to(dev).eval()           # the second synthetic "model": same graph, other weights
        imgs128 = synthetic_images(128, args.size, 2000, dev)
        for _ in range(2):
            ensemble_step(model, m2, imgs128)
        torch.cuda.synchronize()
        ops.PROFILE = prof4 = []
        t4 = timed_steps(lambda: ensemble_step(model, m2, imgs128), 5, dist=None, sync=torch.cuda.synchronize, device=dev)
        ops.PROFILE = None
        conv4 = sum(e0.elapsed_time(e1) for nm, _, e0, e1, _ in prof4 if not nm.startswith('dcnv3')) * 1e-3
        configs4 = {'workload': 'BASELINE configs[4]: inference batch 128, 640x640, 2 synthetic models x (forward + decode + NMS(conf 0.001, iou 0.6, '
                                'multi_label)) + WBF(iou 0.67, skip 0.01) per image; 2 warm + 5 timed steps after the training region',
                    'images_per_s': round(128 * 5 / t4, 2), 'latency_ms_per_image': round(t4 / 5 * 1e3 / 128, 4), 'ms_per_step': round(t4 / 5 * 1e3, 2),
                    'conv_frac': round(conv4 / t4, 3), 'steps': 5, 'batch': 128}
        del m2, imgs128
        model.train()

    if rank == 0:
        # dominant kernel = the conv tile variant with the largest total time; DCNv3 launches carry bytes instead of FLOPs
        by, dcn = {}, {}
        for name, work, e0, e1, _ in prof:
            d = (dcn if name.startswith('dcnv3') else by).setdefault(name, [0, 0.0, 0.0])
            d[0] += 1
            d[1] += work
            d[2] += e0.elapsed_time(e1) * 1e-3
        name, (cnt, flops, secs) = max(by.items(), key=lambda kv: kv[1][2])
        all_flops, all_secs = sum(v[1] for v in by.values()), sum(v[2] for v in by.values())
        achieved = flops / secs / 1e12
        # peak of the matrix instruction the products go through, in ALGORITHMIC FLOPs: bf16x3 issues 3 MFMAs per product
        peak_tf = {None: F32_MFMA_PEAK_TFLOPS, 'bf16': BF16_MFMA_PEAK_TFLOPS, 'bf16x3': round(BF16_MFMA_PEAK_TFLOPS / 3, 1)}[args.amp]
        label = MODELS[args.model]
        step_txt = (f'{label} training step: uint8 ingest + forward (batch-stat BN) + ComputeLoss + backward + '
                    f'{"gradient all-reduce + " if world > 1 else ""}Adam + EMA' if args.mode == 'train' else
                    f'{label} inference step: uint8 ingest + forward + decode{"" if args.no_nms else " + NMS(conf 0.001, iou 0.6, multi_label)"}')
        # which BASELINE configuration this run is - decided by the shape actually run (--size / --nc / --batch), not by the model alone
        shaped = {10: 'VisDrone-shaped', 3: 'UAVDT-shaped', 80: 'COCO-shaped'}.get(nc, f'{nc}-class') + ' synthetic'
        if args.model == 'yolov5s':
            which = 'BASELINE configs[0] graph' + ('' if args.batch == 2 and args.size == 640 else f' at batch {args.batch}, {args.size}x{args.size}')
        elif args.size == 1280 and nc == 3 and args.mode == 'train':
            which = f'per-GPU share of BASELINE configs[3] (UAVDT nc 3, 1280x1280, 8 images per GPU){"" if args.batch == 8 else " at batch %d" % args.batch}'
        elif args.size == 640 and nc == 10 and args.mode == 'train':
            which = ('BASELINE configs[1] shape' if args.batch == 32 else f'configs[1] graph at batch {args.batch}') + \
                    ('' if args.model == 'somi-dcn' else " on the reference's shipped yaml (no DCNv3 site)")
        elif args.size == 640 and nc == 10 and args.mode == 'infer':
            which = ('BASELINE configs[4] shape' if args.batch == 128 else f'configs[4] pipeline at batch {args.batch}') + \
                    ('' if args.model == 'somi-dcn' else ', shipped yaml') + ('' if ensemble else ' without the WBF leg')
        else:
            which = f'no BASELINE configuration: nc {nc}, {args.size}x{args.size}, batch {args.batch}'
        if args.mode == 'infer':
            step_txt = (f'{label} inference step: uint8 ingest + ' +
                        ('2 synthetic models x (forward + decode + NMS(conf 0.001, iou 0.6, multi_label)) + WBF(iou 0.67, skip 0.01) of their '
                         'detections per image (wbf.py:34-35,68)' if ensemble else
                         f'forward + decode{"" if args.no_nms else " + NMS(conf 0.001, iou 0.6, multi_label)"}'))
        out = {
            'metric': (f'images/sec train (forward+loss+backward+Adam+EMA) @{args.size}, {shaped}, {label}' if args.mode == 'train'
                       else f'images/sec infer ({"2 models x (forward+NMS) + WBF" if ensemble else "forward+NMS"}) @{args.size}, {shaped}, {label}'),
            'value': round(whole_job_rate(args.batch, args.steps, world, dt), 2), 'unit': 'images/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
            'settle': settle,                                           # un-timed steps before the W warm-up steps until the step time stood still
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': {None: 'f32', 'bf16': 'bf16 (conv products; fp32 accumulate, tensors, BN, loss, optimizer)',
                      'bf16x3': 'bf16x3 (conv products as 3 bf16 MFMAs on hi/lo splits; fp32 everything else)'}[args.amp], 'data': 'synthetic',
            'config': {'workload': f'{step_txt}, {args.size}x{args.size}, batch {args.batch}/GPU ({which})',
                       'batch_per_gpu': args.batch, 'imgsz': args.size, 'params': nparams, 'classes': nc,
                       'parallelism': ((f'dp{world}' + (' sync-bn' if args.sync_bn and world > 1 else '')) if args.mode == 'train'
                                       else f'replicas x{world}')},
            'infer_images_per_s_per_gpu': None if infer_ips is None else round(infer_ips, 2),   # ONE model: forward + decode + NMS
            'latency_ms_per_image': round(dt / args.steps * 1e3 / args.batch, 4),   # device time of one step / images in it (throughput latency)
            'roofline': {'bound': 'mfma', 'kernel': name, 'achieved': round(achieved, 2), 'peak': peak_tf,
                         'unit': 'TFLOP/s', 'frac': round(achieved / peak_tf, 4), 'traffic': None,
                         'launches': cnt, 'avg_launch_us': round(secs / cnt * 1e6, 2),
                         'avg_launch_gflop': round(flops / cnt / 1e9, 3),
                         'all_conv_tflops': round(all_flops / all_secs / 1e12, 2),
                         'conv_share_of_step': round(all_secs / dt, 3)},
        }
        # the committed counter passes are of the DEFAULT command (DCN graph, training, batch 32, 640): null for any other workload
        profiled = args.model == 'somi-dcn' and args.mode == 'train' and args.batch == 32 and args.size == 640 and args.amp is None
        out['roofline']['traffic'] = pmc_traffic(name) if profiled else None
        if dcn:                                                  # the DCNv3 operator kernels of the step against the HBM roofline
            kernels = {}
            for k, (c_, b_, s_) in sorted(dcn.items()):
                kernels[k] = {'launches': c_, 'avg_launch_us': round(s_ / c_ * 1e6, 2), 'avg_launch_MB': round(b_ / c_ / 1e6, 1),
                              'achieved_GBps': round(b_ / s_ / 1e9, 1), 'frac': round(b_ / s_ / 1e9 / HBM_PEAK_GBPS, 4),
                              'traffic': pmc_traffic_op(k) if profiled else None}
            tb, ts = sum(v[1] for v in dcn.values()), sum(v[2] for v in dcn.values())
            out['roofline_dcnv3'] = {'bound': 'hbm', 'achieved': round(tb / ts / 1e9, 1), 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                                     'frac': round(tb / ts / 1e9 / HBM_PEAK_GBPS, 4), 'share_of_step': round(ts / dt, 4), 'kernels': kernels,
                                     'note': 'algorithmic bytes 4(2C+3GK) forward / 4(4C+6GK) backward per output pixel (SURVEY 8d)'}
        if configs4:
            out['configs4'] = configs4
        if exchange:
            out['allreduce'] = exchange
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.model, args.mode, args.size, nc, wbf=ensemble)
        sys.stdout.flush()
        os.write(_REAL_STDOUT if _REAL_STDOUT is not None else 1, (json.dumps(out) + '\n').encode())
    if dist:
        dist.destroy_process_group()


_REAL_STDOUT = None


if __name__ == '__main__':
    # ONE JSON line on stdout, nothing else: libraries print there too (RCCL writes its version banner to stdout when the first communicator comes up),
    # so everything but the final line goes to stderr - file descriptor 1 points at stderr for the run, the line is written to the real stdout.
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    main()
