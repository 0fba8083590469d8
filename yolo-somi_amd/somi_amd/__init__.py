"""somi_amd: host-side mirror of the YOLO-SOMI hot path over libsomi_hip.so (MI355X / gfx950).

Python is the reference's host language for this path (models/yolo.py, utils/loss.py, utils/general.py, train.py, val.py), so
the host layer is Python too; every device operation goes through the C ABI in include/somi_hip.h.  Modules, by the reference
file they stand in for:

    model, blocks    models/yolo.py (Model, parse_model, DecoupledDetect), models/common.py (Conv, C2fCBAM, ODConv_3rd, SPPF, BiFPN, SEAM)
    dcnv3            models/ops_dcnv3 (extension functions, DCNv3Function, DCNv3 module)
    loss             utils/loss.py ComputeLoss, utils/RepulsionLoss.py repulsion_loss
    nms, wbf         utils/general.py non_max_suppression, wbf.py / ensemble_boxes weighted_boxes_fusion
    metrics, val     val.py process_batch / loop body, utils/metrics.py ap_per_class
    train, optim, ddp   train.py step (forward, loss, backward, Adam + ModelEMA), DDP gradient exchange
    augment          utils/datasets.py LoadImagesAndLabels.__getitem__ / collate_fn, utils/augmentations.py (device input pipeline)
    checkpoint       models/experimental.py attempt_load (reads the reference's pickled checkpoints)
    graph            hipGraph replay of the inference forward

Importing the package only loads the library binding; the submodules import torch-facing code on demand.
"""
from . import _lib  # noqa: F401

__all__ = ['_lib']
