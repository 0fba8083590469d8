"""Reading the reference's checkpoints without the reference's code.

train.py:310-317 pickles whole module objects (`{'model': Model(...).half(), 'ema': ema.ema.half(), 'updates', 'optimizer', ...}`)
and `attempt_load` (models/experimental.py:90-122) needs every class of `models/` importable to unpickle them.  Here the unpickler
substitutes an empty nn.Module for any class it cannot (or is told not to) import: nn.Module's own state (`_parameters`,
`_buffers`, `_modules`, plain attributes such as `yaml`, `names`, `stride`) is restored by pickle as usual, which is all that
is needed to read the weights and the architecture description back.

    model, info = attempt_load('best.pt')          # somi_amd.Model in eval mode, EMA weights when present (like attempt_load)
"""
import io
import pickle
import types

import torch
import torch.nn as nn


class ForeignModule(nn.Module):
    """Stand-in for a module class of the checkpoint's code base that is not importable here."""

    def forward(self, *a, **k):
        raise RuntimeError('this module is a container read from a foreign checkpoint; load its state_dict into somi_amd.Model')


def _pickle_module(foreign_prefixes):
    stubs = {}

    class Unpickler(pickle.Unpickler):
        def find_class(self, module, name):
            foreign = any(module == p or module.startswith(p + '.') for p in foreign_prefixes)
            if not foreign:
                try:
                    return super().find_class(module, name)
                except (ImportError, AttributeError):
                    pass
            key = (module, name)
            if key not in stubs:
                stubs[key] = type(name, (ForeignModule,), {'__module__': module})
            return stubs[key]

    mod = types.ModuleType('somi_amd_ckpt_pickle')
    mod.Unpickler = Unpickler
    mod.load = lambda f, **kw: Unpickler(f, **kw).load()
    mod.__name__ = 'pickle'
    return mod


def read_checkpoint(path_or_bytes, foreign_prefixes=('models', 'utils')):
    """torch.load of a reference checkpoint with stand-ins for its code base -> the checkpoint dict (modules are ForeignModule trees)."""
    f = io.BytesIO(path_or_bytes) if isinstance(path_or_bytes, (bytes, bytearray)) else path_or_bytes
    return torch.load(f, map_location='cpu', pickle_module=_pickle_module(tuple(foreign_prefixes)), weights_only=False)


def attempt_load(weights, device='cuda', foreign_prefixes=('models', 'utils')):
    """models/experimental.py:90-122 for one weights file: EMA module if present else 'model', float, eval -> (somi_amd.Model, info).
    The architecture comes from the pickled module's own `.yaml` (models/yolo.py:1176-1187 keeps it), the weights from its
    state_dict; `.names`, `.stride`, `.hyp` are carried over when present."""
    from .model import Model
    ckpt = read_checkpoint(weights, foreign_prefixes)
    src = ckpt['ema'] if isinstance(ckpt, dict) and ckpt.get('ema') is not None else (ckpt['model'] if isinstance(ckpt, dict) else ckpt)
    if isinstance(src, dict):
        raise RuntimeError('the checkpoint holds a bare state_dict: build somi_amd.Model(cfg) and call load_state_dict on it')
    cfg = getattr(src, 'yaml', None)
    if not isinstance(cfg, dict):
        raise RuntimeError('the pickled model carries no .yaml architecture dict')
    state = {k: v.float() if v.is_floating_point() else v for k, v in src.state_dict().items()}
    model = Model(dict(cfg))
    missing, unexpected = model.load_state_dict(state, strict=False)
    missing = [k for k in missing if 'anchor_grid' not in k]
    if missing or unexpected:
        raise RuntimeError(f'checkpoint does not match the SOMI graph: missing {missing[:5]}, unexpected {list(unexpected)[:5]}')
    for attr in ('names', 'hyp'):
        if hasattr(src, attr):
            setattr(model, attr, getattr(src, attr))
    info = {k: ckpt.get(k) for k in ('epoch', 'best_fitness', 'updates', 'date')} if isinstance(ckpt, dict) else {}
    info['used'] = 'ema' if isinstance(ckpt, dict) and ckpt.get('ema') is not None else 'model'
    return model.to(device).eval(), info
