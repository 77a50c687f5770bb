"""pointcept.models on MI355X: registers only what the PTv3 / Swin3D window-attention path implements.

The reference's pointcept/models/__init__.py:1-45 imports every backbone (spconv, MinkowskiEngine,
ocnn, torch_cluster, peft ...) and cannot be imported on a ROCm box; this package replaces it.
"""
from .builder import MODELS, MODULES, build_model  # noqa: F401
from .default import DefaultSegmentor, DefaultSegmentorV2  # noqa: F401
from .modules import PointModule, PointSequential, PointModel  # noqa: F401
from .point_transformer_v3 import *  # noqa: F401,F403
from .offset_keypoint_ptv3 import OffsetKeypointPTv3  # noqa: F401
from .swin3d import Swin3DUNet  # noqa: F401
from .offset_keypoint_swin3d import OffsetKeypointSwin3D  # noqa: F401
