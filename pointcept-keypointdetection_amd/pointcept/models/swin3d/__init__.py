from .swin3d_layers import (WindowAttention, Mlp, SwinTransformerBlock, WindowStage, window_attn_args,  # noqa: F401
                            sparse_self_attention)
