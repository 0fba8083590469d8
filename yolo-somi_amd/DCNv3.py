"""Drop-in for the reference's compiled extension module `DCNv3` (models/ops_dcnv3/src/vision.cpp:14-17).

`models/ops_dcnv3/functions/dcnv3_func.py:16` does `import DCNv3` and calls `DCNv3.dcnv3_forward(...)` /
`DCNv3.dcnv3_backward(...)` (:39-43, :54-58).  With this directory on sys.path that import resolves here and the reference's
`DCNv3Function` / `DCNv3` nn.Module run on libsomi_hip.so unchanged: same positional signatures (src/dcnv3.h:20-26,40-47), the same
argument checks surfacing as RuntimeError (src/cuda/dcnv3_cuda.cu:29-53), float / double / half like AT_DISPATCH_FLOATING_TYPES_AND_HALF
(:69), outputs freshly allocated, launched on the current stream.  CPU tensors raise, as in the reference ("Not implemented on the
CPU", src/dcnv3.h:37,58).
"""
from somi_amd.dcnv3 import dcnv3_backward, dcnv3_forward  # noqa: F401

__all__ = ['dcnv3_forward', 'dcnv3_backward']
