"""CPU-side checks of the host layer: the C ABI exports every symbol the header declares, the product Model has
the reference's parameter layout (checked against the oracle, which is pinned to the reference), and the product
refuses to run without a GPU instead of falling back."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'somi_hip.h')).read()
    declared = set(re.findall(r'\b(somi_[a-z0-9_]+)\s*\(', hdr))
    declared -= {'somi_stream_t'}
    assert len(declared) >= 20
    from somi_amd import _lib
    assert set(_lib.SIGNATURES) == declared, set(_lib.SIGNATURES) ^ declared
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(L, name), f'{name} is declared in include/somi_hip.h but not exported'
    assert _lib.lib().somi_abi_version() == _lib.ABI_VERSION == 14


def test_one_hip_runtime_in_the_process_whatever_is_imported_first():
    """The binding loaded BEFORE torch used to map /opt/rocm's libamdhip64 next to the copy torch's wheel carries: two HIP runtimes, and the library's
    launches failed with "no ROCm-capable device" (seen on the GPU box through `python __graft_entry__.py smoke`, whose build() loads the library
    first).  _lib.lib() imports torch first; a fresh process that touches the binding before anything else must end up with ONE mapped runtime."""
    code = ("import sys; sys.path.insert(0, %r); from somi_amd import _lib; _lib.lib(); import torch; "
            "paths = {l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l}; print(len(paths), sorted(paths))" % os.path.join(ROOT, 'yolo-somi_amd'))
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split()[0] == '1', out.stdout


def test_struct_layouts_match_header():
    from somi_amd._lib import ConvDesc, LossDesc
    assert ctypes.sizeof(ConvDesc) == 9 * 8 + 22 * 4 + 8 + 8 + 8 + 3 * 8 + 4 + 4      # ... + prec + tail padding
    from somi_amd import _lib
    assert _lib.lib().somi_sizeof_desc(0) == ctypes.sizeof(ConvDesc) and _lib.lib().somi_sizeof_desc(1) == ctypes.sizeof(LossDesc)
    assert ctypes.sizeof(LossDesc) == 4 * 8 + 4 * 8 + 4 * 4 + 4 * 4 + 5 * 4 + 4 + 2 * 8 + 4 * 4 + 13 * 4 + 4


@pytest.mark.parametrize('width,depth,anchors', [(0.25, 0.33, 4), (1.0, 1.0, 'visdrone')])
def test_model_parameter_layout_matches_reference(width, depth, anchors):
    from oracle.somi_ref import Model as OModel
    from oracle.somi_ref.testing import somi_cfg, SOMI_ANCHORS
    from somi_amd.model import Model
    anchors = SOMI_ANCHORS if anchors == 'visdrone' else anchors
    cfg = somi_cfg(width, depth, anchors=anchors)
    ref, mine = OModel(cfg), Model(cfg)
    rs, ms = ref.state_dict(), mine.state_dict()
    assert set(rs) == set(ms), sorted(set(rs) ^ set(ms))[:10]
    for k in rs:
        assert rs[k].shape == ms[k].shape, k
    assert torch.equal(ref.stride, mine.stride)
    assert torch.equal(ref.model[-1].anchors, mine.model[-1].anchors)
    assert ref.save == mine.save
    mine.load_state_dict(rs)
    if width == 1.0:
        assert sum(p.numel() for p in mine.parameters()) == 77537610


@pytest.mark.parametrize('version', ['6.0', '5.0'])
def test_stock_yolov5_parameter_layout_matches_reference(version):
    """BASELINE configs[0]: the stock graphs (C3 / Bottleneck / Concat / SPP(F) / Focus / Detect) build with the reference's
    state_dict layout, strides, scaled anchors, bias priors and save list."""
    from oracle.somi_ref import Model as OModel
    from oracle.somi_ref.testing import yolov5_cfg
    from somi_amd.model import Model
    cfg = yolov5_cfg(version=version)
    ref, mine = OModel(cfg), Model(cfg)
    rs, ms = ref.state_dict(), mine.state_dict()
    assert list(rs) == list(ms)
    for k in rs:
        assert rs[k].shape == ms[k].shape, k
    assert torch.equal(ref.stride, mine.stride) and torch.equal(ref.model[-1].anchors, mine.model[-1].anchors)
    assert ref.save == mine.save
    mine.load_state_dict(rs)
    for a, b in zip(ref.model[-1].m, mine.model[-1].m):
        assert torch.equal(a.bias, b.bias)
    if version == '6.0':
        assert sum(p.numel() for p in mine.parameters()) == 7235389


def test_product_refuses_cpu_tensors():
    from oracle.somi_ref.testing import somi_cfg
    from somi_amd.model import Model
    m = Model(somi_cfg(0.25, 0.33)).eval()
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m(torch.zeros(1, 3, 64, 64))


def test_unknown_module_is_rejected():
    from oracle.somi_ref.testing import somi_cfg
    from somi_amd.model import Model
    cfg = somi_cfg(0.25, 0.33)
    cfg['backbone'][0][2] = 'C3TR'
    with pytest.raises(NotImplementedError, match='outside the SOMI hot path'):
        Model(cfg)


def test_product_config_data_matches_oracle_copy():
    """somi_amd.configs (product) and oracle.somi_ref.testing (test infrastructure) carry the same data and fill."""
    import torch.nn as nn
    from oracle.somi_ref import testing as O
    from somi_amd import configs as P
    assert O.somi_cfg(0.5, 0.67) == P.somi_cfg(0.5, 0.67)
    assert O.SOMI_ANCHORS == P.SOMI_ANCHORS and O.HYP_VISDRONE == P.HYP_VISDRONE
    assert O.tiny_somi_cfg() == P.tiny_somi_cfg() and O.somi_cfg(dcn=True) == P.somi_cfg(dcn=True)
    assert O.COCO_ANCHORS == P.COCO_ANCHORS and all(O.yolov5_cfg(version=v) == P.yolov5_cfg(version=v) for v in ('6.0', '5.0'))
    a, b = nn.Sequential(nn.Conv2d(3, 8, 3), nn.BatchNorm2d(8)), nn.Sequential(nn.Conv2d(3, 8, 3), nn.BatchNorm2d(8))
    O.fill_state(a, 3), P.fill_state(b, 3)
    for (k, u), (_, v) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(u, v), k
    ia, ta = O.synthetic_batch(2, 32, seed=4)
    ib, tb = P.synthetic_batch(2, 32, seed=4)
    assert torch.equal(ia, ib) and torch.equal(ta, tb)


def test_read_checkpoint_without_the_checkpoints_code_base():
    """A reference-style checkpoint (train.py:310-317: whole modules pickled, fp16) is read back with stand-in classes for
    every class of the code base that wrote it; the oracle's module tree plays the reference's here."""
    import copy
    import io
    from oracle.somi_ref import Model as OModel
    from oracle.somi_ref.testing import SOMI_ANCHORS, fill_state, somi_cfg
    from somi_amd.checkpoint import ForeignModule, read_checkpoint
    ref = fill_state(OModel(somi_cfg(0.25, 0.33, anchors=SOMI_ANCHORS)), 3)
    buf = io.BytesIO()
    torch.save({'epoch': 7, 'model': copy.deepcopy(ref).half(), 'ema': copy.deepcopy(ref).half(), 'updates': 11, 'optimizer': None}, buf)
    ck = read_checkpoint(buf.getvalue(), foreign_prefixes=('oracle',))
    assert ck['epoch'] == 7 and ck['updates'] == 11
    m = ck['ema']
    assert isinstance(m, ForeignModule) and all(isinstance(x, (ForeignModule, torch.nn.Module)) for x in m.modules())
    assert not any(type(x).__module__.startswith('oracle') and not isinstance(x, ForeignModule) for x in m.modules())
    sd, want = m.state_dict(), ref.half().state_dict()
    assert list(sd) == list(want) and all(torch.equal(sd[k], want[k]) for k in sd)
    assert isinstance(m.yaml, dict) and m.yaml['nc'] == 10


def test_warmup_and_one_cycle_schedule():
    """train.py:146,250-256: one_cycle lambda and the per-batch warm-up of lr (biases from warmup_bias_lr) and accumulate."""
    import math
    import types
    from somi_amd.train import one_cycle, scheduler_step, warmup_lr
    hyp = dict(warmup_bias_lr=0.1, warmup_momentum=0.8, momentum=0.843, lrf=0.12)
    lf = one_cycle(1, hyp['lrf'], 100)
    assert abs(lf(0) - 1.0) < 1e-12 and abs(lf(100) - 0.12) < 1e-12 and abs(lf(50) - (0.5 * (0.12 - 1) + 1)) < 1e-12
    opt = types.SimpleNamespace(param_groups=[dict(lr=3e-4, initial_lr=3e-4), dict(lr=3e-4, initial_lr=3e-4, weight_decay=1e-4),
                                              dict(lr=3e-4, initial_lr=3e-4)])
    nw = 1000
    assert warmup_lr(opt, 0, nw, 0, lf, hyp, batch_size=16) == 1
    assert [g['lr'] for g in opt.param_groups] == [0.0, 0.0, 0.1]
    acc = warmup_lr(opt, 500, nw, 2, lf, hyp, batch_size=16)
    assert acc == 2                                              # halfway between 1 and nbs / batch = 4, rounded like numpy (2.5 -> 2)
    want = 0.5 * 3e-4 * lf(2)
    assert abs(opt.param_groups[0]['lr'] - want) < 1e-15 and abs(opt.param_groups[2]['lr'] - (0.05 + want)) < 1e-12
    assert warmup_lr(opt, nw, nw, 3, lf, hyp, batch_size=16) == 4
    assert all(abs(g['lr'] - 3e-4 * lf(3)) < 1e-15 for g in opt.param_groups)
    scheduler_step(opt, 40, lf)
    assert all(abs(g['lr'] - 3e-4 * lf(40)) < 1e-18 for g in opt.param_groups)
    assert math.isclose(lf(100), hyp['lrf'])


def test_reduction_chunking_covers_every_pixel_and_bounds_the_partial_rows():
    """somi_red_nchunk (host arithmetic of the BatchNorm / column-sum reductions; sizes every caller's workspace): the chunks cover all pixels,
    small maps get enough workgroups (>= 32-pixel chunks: a 20x20 map at batch 32 must not leave most of the 256 CUs idle), and the number of
    partial rows a stage-2 fold walks stays <= 1024 up to 4 M pixels."""
    from somi_amd import _lib
    L = _lib.lib()
    for npix in (1, 31, 32, 33, 400, 12800, 51200, 204800, 819200, 3276800, 4194304, 13107200):
        n = L.somi_red_nchunk(npix)
        assert n >= 1
        chunk = -(-npix // n)                                     # pixels per chunk is at least this
        assert n * 4096 >= npix, (npix, n)                        # chunks are capped at 4096 pixels
        if npix <= 4194304:
            assert n <= 1024, (npix, n)
        if npix >= 12800:
            assert n >= 256, (npix, n)                            # every CU gets a workgroup
        assert chunk >= 1


def test_header_is_plain_c():
    """include/somi_hip.h is the drop-in boundary: it must compile as C (no C++ or torch types) and the INTEGRATION.md conv
    example must compile against it."""
    import shutil
    import subprocess
    import tempfile
    gcc = shutil.which('gcc')
    if gcc is None:
        pytest.skip('no gcc')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = '''#include <stdio.h>
#include "somi_hip.h"
int run(const float *x, const float *w_packed, const float *bias, float *y, void *scratch, somi_stream_t stream) {
    somi_conv_desc d = {0};
    d.x = x; d.w = w_packed; d.bias = bias; d.y = y;
    d.B = 32; d.H = d.W = 160; d.Cin = 128; d.x_cs = 128;
    d.Ho = d.Wo = 160; d.Cout = 128; d.y_cs = 128;
    d.kh = d.kw = 3; d.stride = 1; d.pad = 1; d.dil = 1; d.act = SOMI_ACT_SILU;
    d.workspace = scratch; d.workspace_bytes = somi_conv2d_workspace_bytes();
    if (somi_conv2d_nhwc_f32(&d, stream)) { fprintf(stderr, "%s\\n", somi_last_error()); return 1; }
    return 0;
}
'''
    with tempfile.NamedTemporaryFile('w', suffix='.c', delete=False) as f:
        f.write(src)
    try:
        r = subprocess.run([gcc, '-std=c99', '-Wall', '-Werror', '-fsyntax-only', '-I', os.path.join(root, 'include'), f.name],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    finally:
        os.unlink(f.name)


def test_fitness_uses_the_forks_weights():
    """utils/metrics.py:15-18: 0.1 P + 0.1 R + 0.1 mAP@0.5 + 0.7 mAP@0.5:0.95."""
    import numpy as np
    import torch
    from somi_amd.metrics import fitness
    x = np.array([[0.5, 0.4, 0.3, 0.2, 9.0], [1.0, 1.0, 1.0, 1.0, 0.0]])
    assert np.allclose(fitness(x), [0.5 * 0.1 + 0.4 * 0.1 + 0.3 * 0.1 + 0.2 * 0.7, 1.0])
    assert torch.allclose(fitness(torch.from_numpy(x)), torch.tensor([0.26, 1.0], dtype=torch.float64))


def test_ctypes_mirrors_match_the_c_layout():
    """The structs of include/somi_hip.h, compiled as C, have exactly the sizes and field offsets of their ctypes mirrors (the
    sample records of the input pipeline are filled on the host and read by the kernel byte for byte)."""
    import ctypes
    import shutil
    import subprocess
    import tempfile
    from somi_amd import _lib
    gcc = shutil.which('gcc')
    if gcc is None:
        pytest.skip('no gcc')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    probes = [('somi_conv_desc', _lib.ConvDesc, ['y', 'B', 'per_sample_w', 'res2_cs', 'workspace', 'workspace_bytes', 'stat_pivot']),
              ('somi_loss_desc', _lib.LossDesc, ['grad', 'nl', 'targets', 'balance', 'gr']),
              ('somi_aug_source', _lib.AugSource, ['pixels', 'h', 'x1', 'dy']),
              ('somi_aug_canvas', _lib.AugCanvas, ['src', 'nsrc', 'warp', 'minv']),
              ('somi_aug_sample', _lib.AugSample, ['canvas', 'mix', 'fliplr', 'mix_r', 'lut'])]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "somi_hip.h"', 'int main(void) {']
    for cname, _, fields in probes:
        lines.append(f'  printf("{cname} %zu", sizeof({cname}));')
        lines += [f'  printf(" %zu", offsetof({cname}, {f}));' for f in fields]
        lines.append('  printf("\\n");')
    lines += ['  return 0;', '}']
    with tempfile.TemporaryDirectory() as tmp:
        src, exe = os.path.join(tmp, 'layout.c'), os.path.join(tmp, 'layout')
        with open(src, 'w') as f:
            f.write('\n'.join(lines))
        r = subprocess.run([gcc, '-std=c99', '-I', os.path.join(root, 'include'), src, '-o', exe], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout.strip().splitlines()
    for (cname, ct, fields), line in zip(probes, out):
        got = [int(v) for v in line.split()[1:]]
        want = [ctypes.sizeof(ct)] + [getattr(ct, f).offset for f in fields]
        assert got == want, (cname, got, want)


def test_bench_refuses_a_mislaunch():
    """bench.py --gpus N must not print an N=1 number when it was not launched with N ranks: WORLD_SIZE != --gpus exits
    non-zero before anything touches the GPU."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--steps', '1', '--warmup', '0'], capture_output=True, text=True,
                       timeout=300, env=env, cwd=root)
    assert r.returncode != 0 and 'WORLD_SIZE=1' in (r.stderr + r.stdout) and '{' not in r.stdout
    assert r.stdout == ''        # stdout is reserved for the one JSON line (file descriptor 1 points at stderr for the run)


def test_custom_ops_are_registered_device_only():
    """north_star: 'surfaced to Python through PyTorch-ROCm custom ops'.  torch.ops.somi.* exist with the reference extension's
    signatures, have shape (fake) kernels, and have NO CPU kernel - a CPU tensor fails loudly instead of falling back."""
    import DCNv3                                                  # yolo-somi_amd/DCNv3.py: what `import DCNv3` (dcnv3_func.py:16) finds
    import somi_amd.torch_ops  # noqa: F401
    from torch._subclasses.fake_tensor import FakeTensorMode
    assert callable(DCNv3.dcnv3_forward) and callable(DCNv3.dcnv3_backward)
    for name in ('dcnv3_forward', 'dcnv3_backward', 'conv2d_nhwc', 'nms', 'yolo_loss'):
        assert hasattr(torch.ops.somi, name), name
    schema = str(torch.ops.somi.dcnv3_forward.default._schema)
    assert 'Int kernel_h' in schema and 'float offset_scale' in schema and 'Int im2col_step' in schema      # int / SymInt
    with FakeTensorMode():
        x, o, m = torch.empty(2, 9, 7, 64), torch.empty(2, 5, 4, 72), torch.empty(2, 5, 4, 36)
        assert torch.ops.somi.dcnv3_forward(x, o, m, 3, 3, 2, 2, 1, 1, 1, 1, 4, 16, 1.0, 256).shape == (2, 5, 4, 64)
    with pytest.raises(NotImplementedError, match="'CPU' backend"):
        torch.ops.somi.nms(torch.zeros(1, 10, 15), 0.25, 0.45, False, False, 300)
    with pytest.raises(RuntimeError, match='CUDA tensor'):
        DCNv3.dcnv3_forward(torch.zeros(1, 4, 4, 16), torch.zeros(1, 4, 4, 72), torch.zeros(1, 4, 4, 36), 3, 3, 1, 1, 1, 1, 1, 1, 4, 4, 1.0, 256)


def test_read_checkpoint_pickled_by_the_reference_itself(golden):
    """tests/golden/checkpoint_ref.npz holds the bytes of a checkpoint written the way train.py:310-317 writes it, from the reference's
    OWN classes (models.yolo.Model, models.common.*, Conv under ultralytics.nn.modules.conv - SURVEY fact 5).  Read back with the
    default foreign_prefixes and none of that code importable: every class becomes a stand-in, the weights are the half-rounded
    fill_state values, the architecture dict and the side attributes survive."""
    from oracle.somi_ref import Model as OModel
    from oracle.somi_ref.testing import fill_state, tiny_somi_cfg, yolov5_cfg
    from somi_amd.checkpoint import ForeignModule, read_checkpoint
    g = golden('checkpoint_ref')
    ck = read_checkpoint(g['bytes'].tobytes())
    assert ck['epoch'] == 12 and ck['updates'] == 345 and ck['optimizer'] is None and abs(float(ck['best_fitness'][0]) - 0.4321) < 1e-12
    for key, cfg, seed in (('ema', tiny_somi_cfg(), 4), ('model', yolov5_cfg(0.125, 0.33, nc=80), 5)):
        m = ck[key]
        assert isinstance(m, ForeignModule) and type(m).__module__ == 'models.yolo' and type(m).__name__ == 'Model'
        mods = {type(x).__module__ for x in m.modules()}
        assert 'ultralytics.nn.modules.conv' in mods and all(isinstance(x, (ForeignModule, torch.nn.Module)) for x in m.modules())
        want = fill_state(OModel(cfg), seed).half().state_dict()
        got = m.state_dict()
        assert list(got) == list(want)
        for k in want:
            assert got[k].dtype == want[k].dtype and torch.equal(got[k], want[k]), k
        assert isinstance(m.yaml, dict) and m.yaml['nc'] == cfg['nc']
    assert ck['ema'].names[3] == 'class3' and ck['ema'].hyp['anchor_t'] == 3
