"""CPU restatement of the YOLO-SOMI hot path (TEST INFRASTRUCTURE ONLY).

This package is the *oracle*: a plain-PyTorch, CPU-only restatement of the reference's
arithmetic for the path SURVEY.md section 8 scopes (Model.forward, ComputeLoss, NMS, WBF,
DCNv3).  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  The product path (``yolo-somi_amd/``) never does: it fails
loudly when the HIP library is missing.

Pinning: every function here is checked in ``tests/test_oracle_golden.py``
against vectors in ``tests/golden/`` that were produced by running the reference's own
Python (``/root/reference``) through the stub-import harness ``oracle/gen_golden.py``.
Two pieces of arithmetic live in third-party packages that are not in the reference tree
and are therefore "parity unpinned": ``torchvision.ops.nms`` (torchvision==0.14.1) and
``ensemble_boxes.weighted_boxes_fusion`` (ensemble-boxes==1.0.9); their published
algorithms are restated in ``nms.py`` / ``wbf.py`` and anchored on the reference call
sites (utils/general.py:694, wbf.py:68).
"""
from .blocks import *  # noqa: F401,F403
from .model import Model, parse_model  # noqa: F401
