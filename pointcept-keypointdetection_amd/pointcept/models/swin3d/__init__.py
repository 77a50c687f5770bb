from .swin3d_layers import (WindowAttention, Mlp, SwinTransformerBlock, WindowStage, window_attn_args,  # noqa: F401
                            sparse_self_attention)
from .swin3d_v1m1_base import Swin3DUNet, BasicLayer, GridKNNDownsample, Upsample  # noqa: F401
