"""somi_amd: host-side mirror of the YOLO-SOMI hot path over libsomi_hip.so (MI355X / gfx950).

Python is the reference's host language for this path (models/yolo.py, utils/loss.py, utils/general.py), so the
host layer is Python too; every device operation goes through the C ABI in include/somi_hip.h.
"""
from . import _lib  # noqa: F401

__all__ = ['_lib']
