"""Whole-model parity on the MI355X: somi_amd.Model (HIP kernels through the C ABI) against the golden outputs
of the reference model (tests/golden/model_*.npz) and against the CPU oracle at other shapes.
Bar: 1e-3 relative in fp32 (BASELINE.json north_star)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def rel_close(got, want, rel=1e-3, what=''):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = (got - want).abs().max().item()
    scale = want.abs().max().item() + 1e-12
    assert err <= rel * scale, f'{what}: max err {err:.3e} vs scale {scale:.3e}'


def build(width, depth, anchors, seed=1):
    from oracle.somi_ref import Model as OModel
    from oracle.somi_ref.testing import fill_state, somi_cfg
    from somi_amd.model import Model
    cfg = somi_cfg(width, depth, anchors=anchors)
    ref = fill_state(OModel(cfg), seed).eval()
    mine = Model(cfg)
    mine.load_state_dict(ref.state_dict())
    return ref, mine.cuda().eval()


@pytest.mark.parametrize('tag,width,depth,anch', [('w025_anch4', 0.25, 0.33, 4), ('w025_visdrone', 0.25, 0.33, 'v'),
                                                  ('full', 1.0, 1.0, 'v')])
def test_model_matches_reference_vectors(golden, tag, width, depth, anch):
    from oracle.somi_ref.testing import SOMI_ANCHORS
    g = golden('model_' + tag)
    _, mine = build(width, depth, SOMI_ANCHORS if anch == 'v' else anch)
    x = T(g['x']).cuda()
    with torch.no_grad():
        z, raw = mine(x)
    torch.cuda.synchronize()
    rel_close(z, T(g['z']), what='z')
    rel_close(z, T(g['z_fused']), what='z vs Model.fuse() output')
    for i, r in enumerate(raw):
        rel_close(r, T(g[f'raw{i}']), what=f'raw{i}')


def test_model_uint8_batch_matches_oracle():
    """The reference's batch contract: uint8 images, /255 in the loop (train.py:249, val.py:150-152)."""
    from oracle.somi_ref.testing import SOMI_ANCHORS, synthetic_batch
    ref, mine = build(0.25, 0.33, SOMI_ANCHORS, seed=2)
    imgs, _ = synthetic_batch(3, 128, seed=5)
    with torch.no_grad():
        zr, rr = ref(imgs.float() / 255)
        z, raw = mine(imgs.cuda())
    rel_close(z, zr, what='z')
    for a, b in zip(raw, rr):
        rel_close(a, b, what='raw')


def test_single_image_skips_odconv_bn():
    """ODConv drops its squeeze BN for a batch of one (models/common.py:4562)."""
    from oracle.somi_ref.testing import SOMI_ANCHORS
    ref, mine = build(0.25, 0.33, SOMI_ANCHORS, seed=3)
    x = torch.rand(1, 3, 64, 96, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        rel_close(mine(x.cuda())[0], ref(x)[0], what='z (B=1, non-square)')
