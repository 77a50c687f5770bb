"""Point Transformer V3 (mode 2, "Sonata" backbone) on MI355X: registry name "PT-v3m2".

Counterpart of the reference's pointcept/models/point_transformer_v3/point_transformer_v3m2_sonata.py:
same class names (LayerScale, RPE, SerializedAttention, MLP, Block, GridPooling, GridUnpooling, Embedding,
PointTransformerV3), constructor keywords and defaults (:545-574), module tree and state_dict keys.
Differences from "PT-v3m1" and where they run:
  * Block carries LayerScale after attention and MLP (:349, :357): a per-channel scale, ptv3_affine_act;
  * Embedding is Linear -> LayerNorm -> GELU (:520-540) and the cloud is serialized AFTER it (:722-725);
  * GridPooling (:402-470) clusters by (batch, grid_coord // stride) in lexicographic order - one 64-bit key per
    point, ptv3_argsort_i64 + ptv3_pool_segments (= torch.unique(dim=0) with inverse and counts), segment max of the
    projected features, mean coordinate, then LayerNorm / GELU and a fresh serialization of the pooled level;
  * GridUnpooling (:497-512) refreshes the sparse tensor after the skip add (no stale-skip quirk of v3m1).
Attention, MLP, xCPE conv and all their kernels are the v3m1 ones.
"""
import torch
import torch.nn as nn

from ptv3_hip import ops
from ptv3_hip import autograd as A
from pointcept.models.builder import MODELS
from pointcept.models.utils.structure import Point
from pointcept.models.utils.sparse import SubMConv3d
from pointcept.models.utils.hip_layers import Linear, LayerNorm, GELU, DropPath, check_sync_batchnorm
from pointcept.models.modules import PointModule, PointSequential
from .point_transformer_v3m1_base import RPE, SerializedAttention, MLP  # noqa: F401  (same classes, same kernels)


class LayerScale(nn.Module):
    def __init__(self, dim, init_values=1e-5, inplace=False):
        super().__init__()
        self.inplace = inplace
        self.gamma = nn.Parameter(init_values * torch.ones(dim))

    def forward(self, x):
        if self.training:
            return x * self.gamma.to(x.dtype)          # taped elementwise scale
        g = self.gamma.detach().float().contiguous()
        return ops.affine_act(x, g, torch.zeros_like(g), ops.ACT_NONE)


class Block(PointModule):
    def __init__(self, channels, num_heads, patch_size=48, mlp_ratio=4.0, qkv_bias=True, qk_scale=None,
                 attn_drop=0.0, proj_drop=0.0, drop_path=0.0, layer_scale=None, norm_layer=LayerNorm,
                 act_layer=GELU, pre_norm=True, order_index=0, cpe_indice_key=None, enable_rpe=False,
                 enable_flash=True, upcast_attention=True, upcast_softmax=True):
        super().__init__()
        self.channels = channels
        self.pre_norm = pre_norm
        self.cpe = PointSequential(
            SubMConv3d(channels, channels, kernel_size=3, bias=True, indice_key=cpe_indice_key),
            Linear(channels, channels),
            norm_layer(channels),
        )
        self.norm1 = PointSequential(norm_layer(channels))
        self.ls1 = PointSequential(LayerScale(channels, init_values=layer_scale) if layer_scale is not None
                                   else nn.Identity())
        self.attn = SerializedAttention(
            channels=channels, patch_size=patch_size, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale,
            attn_drop=attn_drop, proj_drop=proj_drop, order_index=order_index, enable_rpe=enable_rpe,
            enable_flash=enable_flash, upcast_attention=upcast_attention, upcast_softmax=upcast_softmax)
        self.norm2 = PointSequential(norm_layer(channels))
        self.ls2 = PointSequential(LayerScale(channels, init_values=layer_scale) if layer_scale is not None
                                   else nn.Identity())
        self.mlp = PointSequential(MLP(in_channels=channels, hidden_channels=int(channels * mlp_ratio),
                                       out_channels=channels, act_layer=act_layer, drop=proj_drop))
        self.drop_path = PointSequential(DropPath(drop_path) if drop_path > 0.0 else nn.Identity())

    def _residual_branch(self, point, norm, body, scale):
        """One residual branch of the block in either norm placement (reference :343-360):
        pre_norm:  x <- x + drop_path(scale(body(norm(x))));   post-norm:  x <- norm(x + drop_path(scale(body(x))))."""
        skip = point.feat
        branch_in = norm(point) if self.pre_norm else point
        point = self.drop_path(scale(body(branch_in)))
        point.feat = skip + point.feat
        return point if self.pre_norm else norm(point)

    def forward(self, point: Point):
        skip = point.feat
        point = self.cpe(point)                                 # xCPE: conv -> linear -> norm, added to its input (:339-341)
        point.feat = skip + point.feat
        point = self._residual_branch(point, self.norm1, self.attn, self.ls1)
        point = self._residual_branch(point, self.norm2, self.mlp, self.ls2)
        point.sparse_conv_feat = point.sparse_conv_feat.replace_feature(point.feat)
        return point


class GridPooling(PointModule):
    def __init__(self, in_channels, out_channels, stride=2, norm_layer=None, act_layer=None, reduce="max",
                 shuffle_orders=True, traceable=True):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.stride = stride
        assert reduce in ["sum", "mean", "min", "max"]
        if reduce != "max":
            raise NotImplementedError("GridPooling on the HIP path implements reduce='max' (every PTv3 config)")
        if stride < 1 or stride & (stride - 1):
            raise NotImplementedError("GridPooling on the HIP path: stride must be a power of two")
        self.reduce = reduce
        self.shuffle_orders = shuffle_orders
        self.traceable = traceable
        self.proj = Linear(in_channels, out_channels)
        # the reference only defines these attributes when the layer is given (:389-392)
        if norm_layer is not None:
            self.norm = PointSequential(norm_layer(out_channels))
        if act_layer is not None:
            self.act = PointSequential(act_layer())

    def forward(self, point: Point):
        point._ensure_grid_coord()
        shift = self.stride.bit_length() - 1
        gc = point.grid_coord.long().contiguous()
        batch = point.batch.long().contiguous()
        # one lexicographic key per point: (batch, x', y', z') with x' = x // stride  (index plumbing; the
        # reference packs batch << 48 into every column and lets torch.unique(dim=0) sort the rows, :413-421)
        g = gc >> shift
        key = (batch << 48) | (g[:, 0] << 32) | (g[:, 1] << 16) | g[:, 2]
        order, _ = ops.argsort_codes(key.view(1, -1), 64)
        order = order[0].contiguous()
        nb = len(point.offset)
        cluster, seg_start, n_out, pooled_offset, pooled_offset_host = ops.pool_segments(
            key, order, 0, batch=batch, num_scenes=nb)
        proj = self.proj(point.feat)
        feat, coord, grid_coord, batch_out, _ = ops.pool_reduce(
            proj.detach(), point.coord.float().contiguous(), gc, batch, key.view(1, -1), order, seg_start, n_out, shift)
        if self.training:
            feat = A.segment_max(proj, order, seg_start, n_out)
        point_dict = Point(feat=feat, coord=coord, grid_coord=grid_coord, batch=batch_out, offset=pooled_offset,
                           _offset_host=pooled_offset_host)
        if "_grid_max_host" in point.keys():
            point_dict["_grid_max_host"] = [v >> shift for v in point["_grid_max_host"]]
        for key_ in ("condition", "context", "name", "split"):
            if key_ in point.keys():
                point_dict[key_] = point[key_]
        if "grid_size" in point.keys():
            point_dict["grid_size"] = point.grid_size * self.stride
        for key_ in ("origin_coord", "color"):
            if key_ in point.keys():
                raise NotImplementedError(f"GridPooling on the HIP path: mean-pooled '{key_}' (pre-training extras)")
        if self.traceable:
            point_dict["pooling_inverse"] = cluster
            point_dict["pooling_parent"] = point
            point_dict["idx_ptr"] = seg_start.long()
            point_dict["_pool_segments"] = (order, seg_start)
        order_names = point.order
        point = point_dict
        if getattr(self, "norm", None) is not None:
            point = self.norm(point)
        if getattr(self, "act", None) is not None:
            point = self.act(point)
        point.serialization(order=order_names, shuffle_orders=self.shuffle_orders)
        point.sparsify()
        return point


class GridUnpooling(PointModule):
    def __init__(self, in_channels, skip_channels, out_channels, norm_layer=None, act_layer=None, traceable=False):
        super().__init__()
        self.proj = PointSequential(Linear(in_channels, out_channels))
        self.proj_skip = PointSequential(Linear(skip_channels, out_channels))
        if norm_layer is not None:
            self.proj.add(norm_layer(out_channels))
            self.proj_skip.add(norm_layer(out_channels))
        if act_layer is not None:
            self.proj.add(act_layer())
            self.proj_skip.add(act_layer())
        self.traceable = traceable

    def forward(self, point):
        """reference :490-504: the coarse features, projected, are added to the projected skip features of the level
        they were pooled from (row gather by `pooling_inverse`)."""
        for key in ("pooling_parent", "pooling_inverse"):
            assert key in point.keys(), f"GridUnpooling: point without {key}"
        coarse_feat = point.feat                                # kept for traceable=True: proj() overwrites point.feat
        fine = self.proj_skip(point.pop("pooling_parent"))
        up = self.proj(point).feat
        segments = point.get("_pool_segments")
        if self.training and segments is not None:
            carried = A.cluster_gather(up, point.pooling_inverse, segments[0], segments[1])
        else:
            carried = up[point.pooling_inverse]                 # row gather: index plumbing
        fine.feat = fine.feat + carried
        fine.sparse_conv_feat = fine.sparse_conv_feat.replace_feature(fine.feat)
        if self.traceable:
            point.feat = coarse_feat
            fine["unpooling_parent"] = point
        return fine


class Embedding(PointModule):
    def __init__(self, in_channels, embed_channels, norm_layer=None, act_layer=None, mask_token=False):
        super().__init__()
        self.in_channels = in_channels
        self.embed_channels = embed_channels
        self.stem = PointSequential(linear=Linear(in_channels, embed_channels))
        if norm_layer is not None:
            self.stem.add(norm_layer(embed_channels), name="norm")
        if act_layer is not None:
            self.stem.add(act_layer(), name="act")
        self.mask_token = nn.Parameter(torch.zeros(1, embed_channels)) if mask_token else None

    def forward(self, point: Point):
        point = self.stem(point)
        if "mask" in point.keys():
            point.feat = torch.where(point.mask.unsqueeze(-1), self.mask_token.to(point.feat.dtype), point.feat)
        return point


@MODELS.register_module("PT-v3m2")
class PointTransformerV3(PointModule):
    def __init__(
        self,
        in_channels=6,
        order=("z", "z-trans"),
        stride=(2, 2, 2, 2),
        enc_depths=(2, 2, 2, 6, 2),
        enc_channels=(32, 64, 128, 256, 512),
        enc_num_head=(2, 4, 8, 16, 32),
        enc_patch_size=(48, 48, 48, 48, 48),
        dec_depths=(2, 2, 2, 2),
        dec_channels=(64, 64, 128, 256),
        dec_num_head=(4, 4, 8, 16),
        dec_patch_size=(48, 48, 48, 48),
        mlp_ratio=4,
        qkv_bias=True,
        qk_scale=None,
        attn_drop=0.0,
        proj_drop=0.0,
        drop_path=0.3,
        layer_scale=None,
        pre_norm=True,
        shuffle_orders=True,
        enable_rpe=False,
        enable_flash=True,
        upcast_attention=False,
        upcast_softmax=False,
        traceable=False,
        mask_token=False,
        enc_mode=False,
        freeze_encoder=False,
    ):
        super().__init__()
        self.num_stages = len(enc_depths)
        self.order = [order] if isinstance(order, str) else order
        self.shuffle_orders = shuffle_orders
        self.enc_mode = enc_mode
        self.freeze_encoder = freeze_encoder
        self.compute_dtype = None   # None: follow torch autocast (bf16) else fp32; or force a dtype

        assert self.num_stages == len(stride) + 1
        assert self.num_stages == len(enc_depths) == len(enc_channels) == len(enc_num_head) == len(enc_patch_size)
        if not self.enc_mode:
            assert self.num_stages == len(dec_depths) + 1 == len(dec_channels) + 1
            assert self.num_stages == len(dec_num_head) + 1 == len(dec_patch_size) + 1

        ln_layer, act_layer = LayerNorm, GELU
        self.embedding = Embedding(in_channels=in_channels, embed_channels=enc_channels[0], norm_layer=ln_layer,
                                   act_layer=act_layer, mask_token=mask_token)
        block_kw = dict(mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                        proj_drop=proj_drop, layer_scale=layer_scale, norm_layer=ln_layer, act_layer=act_layer,
                        pre_norm=pre_norm, enable_rpe=enable_rpe, enable_flash=enable_flash,
                        upcast_attention=upcast_attention, upcast_softmax=upcast_softmax)
        k = len(self.order)

        rates = [x.item() for x in torch.linspace(0, drop_path, sum(enc_depths))]
        self.enc = PointSequential()
        for s in range(self.num_stages):
            stage_rates = rates[sum(enc_depths[:s]): sum(enc_depths[: s + 1])]
            enc = PointSequential()
            if s > 0:
                enc.add(GridPooling(in_channels=enc_channels[s - 1], out_channels=enc_channels[s],
                                    stride=stride[s - 1], norm_layer=ln_layer, act_layer=act_layer), name="down")
            for i in range(enc_depths[s]):
                enc.add(Block(channels=enc_channels[s], num_heads=enc_num_head[s], patch_size=enc_patch_size[s],
                              drop_path=stage_rates[i], order_index=i % k, cpe_indice_key=f"stage{s}", **block_kw),
                        name=f"block{i}")
            if len(enc) != 0:
                self.enc.add(module=enc, name=f"enc{s}")

        if not self.enc_mode:
            rates = [x.item() for x in torch.linspace(0, drop_path, sum(dec_depths))]
            self.dec = PointSequential()
            dec_channels = list(dec_channels) + [enc_channels[-1]]
            for s in reversed(range(self.num_stages - 1)):
                stage_rates = rates[sum(dec_depths[:s]): sum(dec_depths[: s + 1])]
                stage_rates.reverse()
                dec = PointSequential()
                dec.add(GridUnpooling(in_channels=dec_channels[s + 1], skip_channels=enc_channels[s],
                                      out_channels=dec_channels[s], norm_layer=ln_layer, act_layer=act_layer,
                                      traceable=traceable), name="up")
                for i in range(dec_depths[s]):
                    dec.add(Block(channels=dec_channels[s], num_heads=dec_num_head[s], patch_size=dec_patch_size[s],
                                  drop_path=stage_rates[i], order_index=i % k, cpe_indice_key=f"stage{s}",
                                  **block_kw), name=f"block{i}")
                self.dec.add(module=dec, name=f"dec{s}")
        if self.freeze_encoder:
            for p in list(self.embedding.parameters()) + list(self.enc.parameters()):
                p.requires_grad = False
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(module):
        if isinstance(module, (nn.Linear, SubMConv3d)):
            nn.init.trunc_normal_(module.weight, std=0.02)
            if module.bias is not None:
                nn.init.zeros_(module.bias)

    def resolve_dtype(self):
        if self.compute_dtype is not None:
            return self.compute_dtype
        if torch.is_autocast_enabled():
            # float16 autocast (the reference's default amp_dtype) is taken as a request for 16-bit matmuls and served
            # in bfloat16 (see PT-v3m1's resolve_dtype)
            dt = torch.get_autocast_gpu_dtype()
            return torch.bfloat16 if dt in (torch.bfloat16, torch.float16) else torch.float32
        return torch.float32

    def forward(self, data_dict):
        check_sync_batchnorm(self)
        with torch.set_grad_enabled(self.training and torch.is_grad_enabled()):
            point = Point(data_dict)
            feat = point.feat
            if feat.dtype not in (torch.float32, torch.bfloat16):
                feat = feat.float()
            point.feat = ops.cast(feat.contiguous(), self.resolve_dtype())
            point = self.embedding(point)
            point.serialization(order=self.order, shuffle_orders=self.shuffle_orders)
            point.sparsify()
            point = self.enc(point)
            if not self.enc_mode:
                point = self.dec(point)
        return point
