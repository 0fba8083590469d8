"""Where the torch-level glue of one training step comes from: aten::copy_ / fill_ / zero_ / add_ call sites (python stacks) of one
TrainStep.step on the bench graph.  usage: glue_probe.py [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'yolo-somi_amd'))
import torch  # noqa: E402

from somi_amd.configs import HYP_VISDRONE, SOMI_ANCHORS, fill_state, somi_cfg, synthetic_batch  # noqa: E402
from somi_amd.model import Model  # noqa: E402
from somi_amd.train import TrainStep  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda')
model = fill_state(Model(somi_cfg(1.0, 1.0, anchors=SOMI_ANCHORS, dcn=True)), 1).to(dev)
tr = TrainStep(model, dict(HYP_VISDRONE), B)
imgs, targets = synthetic_batch(B, 640, seed=0)
imgs, targets = imgs.to(dev), targets.to(dev)
for _ in range(2):
    tr.step(imgs, targets)
torch.cuda.synchronize()
import collections  # noqa: E402
import traceback  # noqa: E402

sites = collections.Counter()


def spy(name):
    orig = getattr(torch.Tensor, name)

    def wrapped(self, *a, **k):
        fr = [f for f in traceback.extract_stack(limit=8)[:-1] if 'somi_amd' in f.filename or 'bench' in f.filename]
        where = ' <- '.join(f'{os.path.basename(f.filename)}:{f.lineno}' for f in reversed(fr[-3:]))
        sites[(name, tuple(self.shape) if self.dim() < 3 else self.dim(), where)] += 1
        return orig(self, *a, **k)
    setattr(torch.Tensor, name, wrapped)


for n in ('copy_', 'clone', 'fill_', 'zero_', 'add_', 'contiguous'):
    spy(n)
for fn in ('zeros', 'zeros_like', 'full', 'cat', 'tensor'):
    orig = getattr(torch, fn)

    def mk(orig, fn):
        def wrapped(*a, **k):
            fr = [f for f in traceback.extract_stack(limit=8)[:-1] if 'somi_amd' in f.filename or 'bench' in f.filename]
            where = ' <- '.join(f'{os.path.basename(f.filename)}:{f.lineno}' for f in reversed(fr[-3:]))
            sites[('torch.' + fn, '', where)] += 1
            return orig(*a, **k)
        return wrapped
    setattr(torch, fn, mk(orig, fn))
tr.step(imgs, targets)
torch.cuda.synchronize()
for (name, shape, where), n in sites.most_common(45):
    print(f'{name:14s} x{n:4d} {str(shape):14s} {where}')
