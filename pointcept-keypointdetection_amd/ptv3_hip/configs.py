"""Model configurations of the path as plain dicts (data only): the fork's keypoint-offset config and the small
plumbing configs the tests / smoke run use.  Keys are the constructor keywords of "PT-v3m1" / "PT-v3m2"."""
ORDERS = ["z", "z-trans", "hilbert", "hilbert-trans"]

# plumbing-size "PT-v3m1"
TINY_CFG = dict(
    in_channels=4, order=ORDERS, stride=(2, 2, 2, 2),
    enc_depths=(1, 1, 1, 2, 1), enc_channels=(16, 16, 32, 32, 64), enc_num_head=(1, 1, 2, 2, 4),
    enc_patch_size=(64,) * 5, dec_depths=(1, 1, 1, 1), dec_channels=(16, 16, 32, 32),
    dec_num_head=(1, 1, 2, 2), dec_patch_size=(64,) * 4, mlp_ratio=4, qkv_bias=True,
    drop_path=0.3, shuffle_orders=True, pre_norm=True, enable_rpe=False, enable_flash=False,
    upcast_attention=False, upcast_softmax=False,
)

# the fork's config (configs/my_dataset/offset_keypoint_ptv3.py:11-46)
FORK_CFG = dict(
    in_channels=4, order=ORDERS, stride=(2, 2, 2, 2),
    enc_depths=(2, 2, 2, 6, 2), enc_channels=(32, 64, 128, 256, 512), enc_num_head=(2, 4, 8, 16, 32),
    enc_patch_size=(1024,) * 5, dec_depths=(2, 2, 2, 2), dec_channels=(64, 64, 128, 256),
    dec_num_head=(4, 4, 8, 16), dec_patch_size=(1024,) * 4, mlp_ratio=4, qkv_bias=True, qk_scale=None,
    attn_drop=0.0, proj_drop=0.0, drop_path=0.3, shuffle_orders=True, pre_norm=True, enable_rpe=False,
    enable_flash=False, upcast_attention=False, upcast_softmax=False,
)

# the upstream PTv3 semantic-segmentation backbone (configs/scannet/semseg-pt-v3m1-0-base.py:14-44,
# configs/nuscenes/semseg-pt-v3m1-0-base.py): same widths as the fork, enable_flash=True, 1024-point patches
SEMSEG_CFG = dict(FORK_CFG, enable_flash=True)

# "PT-v3m2" (point_transformer_v3m2_sonata.py) plumbing-size config: GridPooling, LayerScale, LayerNorm stem
TINY_M2_CFG = dict(
    in_channels=4, order=ORDERS, stride=(2, 2, 2, 2),
    enc_depths=(1, 1, 1, 2, 1), enc_channels=(16, 16, 32, 32, 64), enc_num_head=(1, 1, 2, 2, 4),
    enc_patch_size=(64,) * 5, dec_depths=(1, 1, 1, 1), dec_channels=(16, 16, 32, 32),
    dec_num_head=(1, 1, 2, 2), dec_patch_size=(64,) * 4, mlp_ratio=4, qkv_bias=True,
    drop_path=0.3, layer_scale=0.5, shuffle_orders=True, pre_norm=True, enable_rpe=False, enable_flash=False,
    upcast_attention=False, upcast_softmax=False,
)

# Swin3D-S backbone of configs/s3dis/semseg-swin3d-v1m1-0-small.py:11-30 (built there under "DefaultSegmentor")
SWIN3D_S3DIS_CFG = dict(
    type="Swin3D-v1m1", in_channels=9, num_classes=13, base_grid_size=0.02, depths=[2, 4, 9, 4, 4],
    channels=[48, 96, 192, 384, 384], num_heads=[6, 6, 12, 24, 24], window_sizes=[5, 7, 7, 7, 7], quant_size=4,
    drop_path_rate=0.3, up_k=3, num_layers=5, stem_transformer=True, down_stride=3, upsample="linear_attn",
    knn_down=True, cRSE="XYZ_RGB_NORM", fp16_mode=1,
)

# the fork's Swin3D offset model (configs/my_dataset/offset_keypoint_swin3d.py:11-40)
OFFSET_SWIN3D_CFG = dict(
    type="OffsetKeypointSwin3D", num_keypoints=6, hidden_dim=256,
    backbone_conf=dict(
        type="Swin3D-v1m1", in_channels=4, num_classes=64, base_grid_size=0.02, quant_size=50, num_layers=4,
        depths=[2, 2, 6, 2], channels=[64, 128, 256, 512], num_heads=[4, 8, 16, 32], window_sizes=[5, 7, 7, 7],
        up_k=3, drop_path_rate=0.2, stem_transformer=True, down_stride=2, upsample="linear", knn_down=True,
        cRSE="XYZ_RGB", fp16_mode=1,
    ),
)

# plumbing-size Swin3D: three levels, both head widths the kernel is built for (8 and 16)
TINY_SWIN3D_CFG = dict(
    type="Swin3D-v1m1", in_channels=9, num_classes=13, base_grid_size=0.02, depths=[2, 2, 2], channels=[16, 32, 32],
    num_heads=[2, 2, 2], window_sizes=[5, 7, 7], quant_size=4, drop_path_rate=0.3, up_k=3, num_layers=3,
    stem_transformer=True, down_stride=3, upsample="linear_attn", knn_down=True, cRSE="XYZ_RGB_NORM", fp16_mode=1,
)

# the two constructor variants no shipped config uses: GridDownsample (knn_down=False) and the residual stem
# (stem_transformer=False) - tests/golden/state_dict_swin3d_tiny_grid_resstem.txt lists the reference class built this way
TINY_SWIN3D_GRID_RESSTEM_CFG = dict(TINY_SWIN3D_CFG, knn_down=False, stem_transformer=False)
