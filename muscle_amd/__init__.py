"""muscle_amd — MI355X-native MCL / MuSCLe training hot path.

Public names mirror the reference's `src/__init__.py:1-6` for the path that is built: the MuSCLe model (CAM-encoder
and decoder modes), the MCL loss callables, `edge.FieldLoss`, the loop bodies (`mcl_step`, `muscle_step`) and the fused
optimiser; `muscle_amd.infer` (CAM generation), `muscle_amd.evaluation` (per-epoch mIoU sweep) and
`muscle_amd.indexing` (IRN random walk) cover the scripts around the training loop.  Importing the package never needs
a GPU; running anything does.
"""
from .MuSCLe import MuSCLe  # noqa: F401
from .loss_multilabel import (FocalLoss, Log_Sum_Exp_Pairwise_Loss, MultiLabelSoftMarginLoss,  # noqa: F401
                              image_level_contrast)
from .optim import FusedAdam  # noqa: F401
from .train_step import cam_softmaxnorm, er_loss, mcl_step, muscle_step  # noqa: F401
from . import edge  # noqa: F401
from .phase2 import EMD, PixPro, cam_maxnorm, get_dynamic_crops  # noqa: F401
from . import phase2 as torchutils  # noqa: F401  (reference name of the module holding get_dynamic_crops)

__all__ = ["MuSCLe", "FocalLoss", "Log_Sum_Exp_Pairwise_Loss", "MultiLabelSoftMarginLoss", "image_level_contrast",
           "FusedAdam", "cam_softmaxnorm", "er_loss", "mcl_step", "EMD", "PixPro", "cam_maxnorm", "get_dynamic_crops",
           "torchutils", "edge", "muscle_step"]
from .graph import GraphedStep  # noqa: F401,E402
from .ops import get_gemm_mode, set_gemm_mode  # noqa: F401,E402
from . import data  # noqa: F401,E402  (input path: two-view sampler + device-side color_norm / crop stage)
