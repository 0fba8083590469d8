"""Where the training step's SMALL launches come from: device-to-device copies, fills and torch element-wise kernels, grouped by the Python
line that issued them (the torch entry points wrapped for one step of the bench graph at a small batch - the launch COUNT per step does not
depend on the batch).

    python tools/small_launch_sites.py [--model somi_dcn] [--top 25]
"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'yolo-somi_amd'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--top', type=int, default=25)
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--size', type=int, default=320)
    args = ap.parse_args()
    from somi_amd.configs import HYP_VISDRONE, somi_cfg, synthetic_batch
    from somi_amd.model import Model
    from somi_amd.train import TrainStep
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    model = Model(somi_cfg(nc=10, dcn=True), ch=3, nc=10).to(dev).train()
    imgs, targets = synthetic_batch(args.batch, args.size, nc=10, seed=1)
    imgs, targets = imgs.to(dev), targets.to(dev)
    trainer = TrainStep(model, dict(HYP_VISDRONE), args.batch)
    for _ in range(2):
        trainer.step(imgs, targets)
    torch.cuda.synchronize()
    # count the callers of the torch entry points that launch small kernels (copies, fills, element-wise arithmetic)
    sites = collections.Counter()
    here = os.path.abspath(ROOT) + '/'

    def wrap(owner, name):
        orig = getattr(owner, name)

        def counted(*a, **k):
            f = sys._getframe(1)
            while f is not None and 'somi_amd' not in f.f_code.co_filename:
                f = f.f_back
            if f is not None:
                sites[(name, f'{f.f_code.co_filename.replace(here, "")}:{f.f_lineno}')] += 1
            return orig(*a, **k)
        setattr(owner, name, counted)
        return orig

    saved = []
    for owner, names in ((torch.Tensor, ['copy_', 'clone', 'contiguous', 'to', 'float', 'add_', 'mul_', 'zero_', 'fill_', '__add__', '__mul__',
                                          '__sub__', '__truediv__', '__iadd__', '__imul__', 'sum', 'mean', 'item', 'tolist', '__getitem__',
                                          '__setitem__']),
                         (torch, ['zeros', 'zeros_like', 'ones', 'full', 'cat', 'stack', 'tensor', 'arange', 'empty_like'])):
        for n in names:
            saved.append((owner, n, wrap(owner, n)))
    trainer.step(imgs, targets)
    torch.cuda.synchronize()
    for owner, n, orig in saved:
        setattr(owner, n, orig)
    kinds = collections.Counter()
    for (name, _), n in sites.items():
        kinds[name] += n
    print('calls by kind:', dict(kinds.most_common(30)))
    for (name, site), n in sites.most_common(args.top):
        print(f'{n:5d}  {name:14s} {site}')


if __name__ == '__main__':
    main()
