"""Point Transformer V3 (mode 1) on MI355X: registry name "PT-v3m1".

Drop-in counterpart of the reference's
pointcept/models/point_transformer_v3/point_transformer_v3m1_base.py: same class names
(RPE, SerializedAttention, MLP, Block, SerializedPooling, SerializedUnpooling, Embedding,
PointTransformerV3), same constructor keywords and defaults (:520-552), same module tree and
state_dict keys (:595-697).  All arithmetic runs in libptv3_hip.so (include/ptv3_hip.h):
serialization + radix argsort, pad plan, fused window attention, implicit-GEMM linear / sparse conv,
LayerNorm, folded BatchNorm, segmented pooling.  Both `enable_flash` settings run the same fused window
kernel.  enable_flash=False: the patch shrinks to the smallest scene (:173-176), every window has K slots.
enable_flash=True: K is fixed, a scene with fewer than K points is ONE short window (:131-133) and the
windows go through the ragged form of the kernel (ptv3_window_attn_varlen_fwd: the semantics of the
flash_attn_varlen_qkvpacked_func call at :207-215, computed in the model's dtype).  Attention dropout
(`attn_drop` > 0 in training: nn.Dropout on the probabilities :203, flash_attn's dropout_p :211) runs in
ptv3_window_attn_drop_fwd / _bwd with a hash-generated keep mask (same distribution as the reference's, not
its Philox values); together with enable_rpe it raises.
"""
import math
from functools import partial

import torch
import torch.nn as nn

from ptv3_hip import ops
from ptv3_hip import autograd as A
from ptv3_hip import engine as _engine
from pointcept.models.builder import MODELS
from pointcept.models.utils.misc import offset2bincount  # noqa: F401  (reference import surface)
from pointcept.models.utils.structure import Point
from pointcept.models.utils.sparse import SubMConv3d, _ParamCache
from pointcept.models.utils.hip_layers import (Linear, LayerNorm, BatchNorm1d, GELU, DropPath, _no_training,
                                                check_sync_batchnorm)
from pointcept.models.modules import PointModule, PointSequential


class RPE(nn.Module):
    """Relative position bias table (:29-48).  Off in the fork's configs (enable_rpe=False)."""

    def __init__(self, patch_size, num_heads):
        super().__init__()
        self.patch_size = patch_size
        self.num_heads = num_heads
        self.pos_bnd = int((4 * patch_size) ** (1 / 3) * 2)
        self.rpe_num = 2 * self.pos_bnd + 1
        self.rpe_table = nn.Parameter(torch.zeros(3 * self.rpe_num, num_heads))
        nn.init.trunc_normal_(self.rpe_table, std=0.02)

    def forward(self, coord):
        # dense fallback only (windows too large for the resident-window kernel): index plumbing that materialises
        # the (windows, H, K, K) bias; the normal path is ops.window_attention_rpe (table lookup inside the kernel)
        idx = (coord.clamp(-self.pos_bnd, self.pos_bnd) + self.pos_bnd
               + torch.arange(3, device=coord.device) * self.rpe_num)
        out = self.rpe_table.float().index_select(0, idx.reshape(-1))
        out = out.view(idx.shape + (-1,)).sum(3)
        return out.permute(0, 3, 1, 2).contiguous()


class SerializedAttention(PointModule):
    def __init__(self, channels, num_heads, patch_size, qkv_bias=True, qk_scale=None, attn_drop=0.0,
                 proj_drop=0.0, order_index=0, enable_rpe=False, enable_flash=True, upcast_attention=True,
                 upcast_softmax=True):
        super().__init__()
        assert channels % num_heads == 0
        self.channels = channels
        self.num_heads = num_heads
        self.scale = qk_scale or (channels // num_heads) ** -0.5
        self.order_index = order_index
        self.upcast_attention = upcast_attention
        self.upcast_softmax = upcast_softmax
        self.enable_rpe = enable_rpe
        self.enable_flash = enable_flash
        if enable_flash:
            assert enable_rpe is False, "Set enable_rpe to False when enable Flash Attention"
            assert upcast_attention is False, "Set upcast_attention to False when enable Flash Attention"
            assert upcast_softmax is False, "Set upcast_softmax to False when enable Flash Attention"
            self.patch_size = patch_size
            self.attn_drop = attn_drop
        else:
            # the reference avoids masks: the patch shrinks to the smallest scene (:91-96)
            self.patch_size_max = patch_size
            self.patch_size = 0
            self.attn_drop = nn.Dropout(attn_drop)
        self.qkv = Linear(channels, channels * 3, bias=qkv_bias)
        self.proj = Linear(channels, channels)
        self.proj_drop = nn.Dropout(proj_drop)
        self.softmax = nn.Softmax(dim=-1)
        self.rpe = RPE(patch_size, num_heads) if self.enable_rpe else None

    @torch.no_grad()
    def get_rel_pos(self, point, order):
        K = self.patch_size
        rel_pos_key = f"rel_pos_{self.order_index}"
        if rel_pos_key not in point.keys():
            grid_coord = point.grid_coord[order].reshape(-1, K, 3)
            point[rel_pos_key] = grid_coord.unsqueeze(2) - grid_coord.unsqueeze(1)
        return point[rel_pos_key]

    @torch.no_grad()
    def get_padding_and_inverse(self, point):
        """pad / unpad / cu_seqlens, cached on the Point like the reference (:114-170)."""
        pad_key, unpad_key, cu_seqlens_key = "pad", "unpad", "cu_seqlens_key"
        if (pad_key not in point.keys() or unpad_key not in point.keys()
                or cu_seqlens_key not in point.keys() or point.get("_pad_patch") != self.patch_size):
            pad, unpad, cu = ops.pad_plan(point.offset.long().contiguous(), point.offset_host(), self.patch_size)
            point[pad_key], point[unpad_key], point[cu_seqlens_key] = pad, unpad, cu
            point["_pad_patch"] = self.patch_size
            for k in [k for k in point.keys() if isinstance(k, str) and k.startswith("_win_maps_")]:
                del point[k]
        return point[pad_key], point[unpad_key], point[cu_seqlens_key]

    @torch.no_grad()
    def window_maps(self, point):
        """order[pad] and unpad[inverse] (:184-185) as int32, cached per (Point, order_index)."""
        key = f"_win_maps_{self.order_index}"
        pad, unpad, _ = self.get_padding_and_inverse(point)
        if key not in point.keys():
            point[key] = ops.window_maps(point.serialized_order[self.order_index],
                                         point.serialized_inverse[self.order_index], pad, unpad)
        return point[key]

    def resolve_patch_size(self, point):
        if not self.enable_flash:
            host = point.offset_host()
            counts = [b - a for a, b in zip([0] + host[:-1], host)]
            self.patch_size = min(min(counts), self.patch_size_max)
        return self.patch_size

    def window_cu(self, point):
        """cu_seqlens of the pad plan when some window is short (a scene with fewer points than the fixed patch of
        enable_flash=True, :131-133), else None: all windows then have K slots."""
        if not self.enable_flash:
            return None
        host = point.offset_host()
        if all(b - a >= self.patch_size for a, b in zip([0] + host[:-1], host)):
            return None
        return self.get_padding_and_inverse(point)[2]

    def _p_attn_drop(self):
        return self.attn_drop.p if isinstance(self.attn_drop, nn.Dropout) else float(self.attn_drop)

    def _check_attn_drop(self):
        if self._p_attn_drop() > 0.0:  # RPE bias and attention dropout together: no kernel (no config of the reference)
            raise NotImplementedError("SerializedAttention: attn_drop > 0 together with enable_rpe has no training "
                                      "kernel on the HIP path")

    def attention_core(self, point, qkv):
        """gather -> softmax(QK^T)V -> scatter (:188-216) in one kernel."""
        K = self.resolve_patch_size(point)
        wo, wi = self.window_maps(point)
        bias = None
        if self.enable_rpe:
            gkey = "_grid_coord_i32"
            if gkey not in point.keys():
                point[gkey] = point.grid_coord.int().contiguous()
            if self.training:
                self._check_attn_drop()
                return A.window_attention_rpe(qkv, self.rpe.rpe_table, wo, wi, point[gkey], self.num_heads, K,
                                              self.scale, self.rpe.pos_bnd)
            out = ops.window_attention_rpe(qkv, wo, wi, self.num_heads, K, self.scale, point[gkey],
                                           self.rpe.rpe_table.detach().float().contiguous(), self.rpe.pos_bnd)
            if out is not None:
                return out
            bias = self.rpe(self.get_rel_pos(point, wo.long()))
        if self.training:
            p_drop = self._p_attn_drop()
            if p_drop > 0.0:
                # dropout on the attention probabilities (:203 nn.Dropout, :211 flash_attn dropout_p): the mask is a hash
                # of (query slot, head, key slot) and a seed drawn here from torch's CPU generator (torch.manual_seed)
                seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
                return A.window_attention_drop(qkv, wo, wi, self.num_heads, K, self.scale, p_drop, seed,
                                               self.window_cu(point))
            return A.window_attention(qkv, wo, wi, self.num_heads, K, self.scale, self.window_cu(point))
        cu = self.window_cu(point)
        if cu is not None:
            return ops.window_attention_varlen(qkv, wo, wi, cu, self.num_heads, K, self.scale)
        return ops.window_attention(qkv, wo, wi, self.num_heads, K, self.scale, rpe_bias=bias)

    def forward(self, point):
        qkv = self.qkv(point.feat)
        feat = self.attention_core(point, qkv)
        feat = self.proj(feat)
        feat = self.proj_drop(feat)   # nn.Dropout: identity at the configs' proj_drop = 0 and in eval
        point.feat = feat
        return point


class MLP(nn.Module):
    def __init__(self, in_channels, hidden_channels=None, out_channels=None, act_layer=GELU, drop=0.0):
        super().__init__()
        out_channels = out_channels or in_channels
        hidden_channels = hidden_channels or in_channels
        self.fc1 = Linear(in_channels, hidden_channels)
        self.act = act_layer()
        self.fc2 = Linear(hidden_channels, out_channels)
        self.drop = nn.Dropout(drop)

    def _act_id(self):
        if isinstance(self.act, nn.GELU):
            return ops.ACT_GELU
        if isinstance(self.act, nn.ReLU):
            return ops.ACT_RELU
        return None

    def forward(self, x, res=None):
        act = self._act_id()
        if self.training:
            x = self.drop(self.act(self.fc1(x)))
            x = self.drop(self.fc2(x))
            return x if res is None else x + res
        if act is None:
            x = self.act(self.fc1(x))
        else:
            x = self.fc1(x, act=act)  # activation in the GEMM epilogue
        return self.fc2(x, res=res)


class Block(PointModule):
    def __init__(self, channels, num_heads, patch_size=48, mlp_ratio=4.0, qkv_bias=True, qk_scale=None,
                 attn_drop=0.0, proj_drop=0.0, drop_path=0.0, norm_layer=LayerNorm, act_layer=GELU,
                 pre_norm=True, order_index=0, cpe_indice_key=None, enable_rpe=False, enable_flash=True,
                 upcast_attention=True, upcast_softmax=True):
        super().__init__()
        self.channels = channels
        self.pre_norm = pre_norm
        self.cpe = PointSequential(
            SubMConv3d(channels, channels, kernel_size=3, bias=True, indice_key=cpe_indice_key),
            Linear(channels, channels),
            norm_layer(channels),
        )
        self.norm1 = PointSequential(norm_layer(channels))
        self.attn = SerializedAttention(
            channels=channels, patch_size=patch_size, num_heads=num_heads, qkv_bias=qkv_bias,
            qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=proj_drop, order_index=order_index,
            enable_rpe=enable_rpe, enable_flash=enable_flash, upcast_attention=upcast_attention,
            upcast_softmax=upcast_softmax)
        self.norm2 = PointSequential(norm_layer(channels))
        self.mlp = PointSequential(MLP(in_channels=channels, hidden_channels=int(channels * mlp_ratio),
                                       out_channels=channels, act_layer=act_layer, drop=proj_drop))
        self.drop_path = PointSequential(DropPath(drop_path) if drop_path > 0.0 else nn.Identity())

    def _fusable(self):
        return (self.pre_norm and isinstance(self.cpe[2], LayerNorm) and isinstance(self.norm1[0], LayerNorm)
                and isinstance(self.norm2[0], LayerNorm) and isinstance(self.cpe[0], SubMConv3d)
                and isinstance(self.cpe[1], Linear))

    def _rows_path(self, feat):
        """the executor's rule (csrc/engine.hip Run::rows_path): every linear of the block served by ptv3_rows_linear and
        enough rows for its grid of 64 | 128-row workgroups"""
        mlp = self.mlp[0]
        n, c = feat.shape
        return (n >= ops.rows_linear_rows() and isinstance(mlp.act, nn.GELU) and self.attn.qkv.bias is not None
                and self.cpe[2].eps == self.norm1[0].eps == self.norm2[0].eps
                and all(ops.rows_linear_capable(c, co, feat.dtype, n) for co in (3 * c, c, mlp.fc1.out_features)))

    def folded_cpe(self, dtype):
        """xCPE conv followed by its Linear (:277-285) has no nonlinearity in between, so the Linear is folded
        into the 27 kernel taps once per weight version:  W'_d = W_lin @ W_d,  b' = W_lin @ b_conv + b_lin.
        Saves one GEMM launch and one (N, C) round trip per block; fp32 reassociation only."""
        conv, lin = self.cpe[0], self.cpe[1]
        cache = self.__dict__.setdefault("_fold_cache", _ParamCache())

        def make():
            wc = conv.weight.detach().float().reshape(conv.out_channels, -1, conv.in_channels)
            wl = lin.weight.detach().float()
            w = torch.einsum("oc,ckd->okd", wl, wc).reshape(lin.out_features, -1)
            b = wl @ conv.bias.detach().float() + lin.bias.detach().float()
            return w.to(dtype).contiguous(), b.contiguous()
        return cache.get(("cpe", dtype), [conv.weight, conv.bias, lin.weight, lin.bias], make)

    def chain_weights(self, dtype):
        """(qkv, proj, fc1, fc2) weight matrices in `dtype`; the three GEMMs fed from registers by the fused block
        kernels get their input channels permuted (ops.chain_permute), once per weight version."""
        mlp = self.mlp[0]
        cache = self.__dict__.setdefault("_fold_cache", _ParamCache())

        def make():
            cast = lambda w: w.detach().to(dtype).contiguous()  # noqa: E731
            perm = ((lambda w: ops.chain_permute(w, dtype))
                    if ops.block_fusable(self.channels, mlp.fc1.out_features, dtype) == 1 else (lambda w: w))
            return (perm(cast(self.attn.qkv.weight)), cast(self.attn.proj.weight),
                    perm(cast(mlp.fc1.weight)), perm(cast(mlp.fc2.weight)))
        return cache.get(("chain", dtype), [self.attn.qkv.weight, self.attn.proj.weight, mlp.fc1.weight,
                                            mlp.fc2.weight], make)

    def _forward_fused_kernels(self, point: Point):
        """conv -> block_head -> window attention -> block_tail: 4 launches."""
        mlp = self.mlp[0]
        spt = point.sparse_conv_feat
        dtype = spt.features.dtype
        wf, bf = self.folded_cpe(dtype)
        wqkv, wproj, w1, w2 = self.chain_weights(dtype)
        nbr = spt.neighbors(3, self.cpe[0].indice_key)
        g0, b0 = self.cpe[2].affine_f32()
        g1, b1 = self.norm1[0].affine_f32()
        g2, b2 = self.norm2[0].affine_f32()
        eps = self.cpe[2].eps
        slabs = ops.conv_slabs(spt.features, wf, nbr, 27, spt.row_order)
        if slabs is not None:
            f1, qkv = ops.block_head(None, slabs[0], slabs[1], bf, point.feat, g0, b0, g1, b1, wqkv,
                                     self.attn.qkv.bias_f32(), eps)
        else:
            x = ops.gemm(spt.features, wf, bias=bf, nbr=nbr, kvol=27, row_order=spt.row_order)
            f1, qkv = ops.block_head(x, None, 0, None, point.feat, g0, b0, g1, b1, wqkv, self.attn.qkv.bias_f32(), eps)
        x = self.attn.attention_core(point, qkv)
        feat = ops.block_tail(x, f1, wproj, self.attn.proj.bias_f32(), g2, b2, w1, mlp.fc1.bias_f32(), w2,
                              mlp.fc2.bias_f32(), eps)
        point.feat = feat
        point.sparse_conv_feat = spt.replace_feature(feat)
        return point

    def _train_fusable(self):
        mlp = self.mlp[0]
        drops = [self.attn.proj_drop, mlp.drop] + ([self.attn.attn_drop] if isinstance(self.attn.attn_drop, nn.Dropout) else [])
        flash_drop = (not isinstance(self.attn.attn_drop, nn.Dropout)) and float(self.attn.attn_drop) > 0.0
        return (self._fusable() and isinstance(mlp.act, nn.GELU) and not self.attn.enable_rpe and not flash_drop
                and all(d.p == 0.0 for d in drops) and self.cpe[0].bias is not None
                and self.cpe[2].eps == self.norm1[0].eps == self.norm2[0].eps
                and isinstance(self.drop_path[0], (DropPath, nn.Identity)))

    def _drop_draw(self, point):
        """(uniform draw per point (fp32), keep probability) of one DropPath, or None.  The draws of a level come from one
        torch.rand pool on the Point (16 rows at a time): one launch per eight blocks instead of three per DropPath; the
        factor u < keep ? 1 / keep : 0 (timm drop_path, scale_by_keep) is formed inside the block kernels."""
        dp = self.drop_path[0]
        if not isinstance(dp, DropPath) or dp.drop_prob == 0.0:
            return None
        if not dp.scale_by_keep:
            raise NotImplementedError("DropPath(scale_by_keep=False) on the fused training path")
        pool = point.get("_drop_pool")
        if pool is None or pool[1] >= pool[0].shape[0]:
            pool = [torch.rand((16, point.feat.shape[0]), dtype=torch.float32, device=point.feat.device), 0]
            point["_drop_pool"] = pool
        row = pool[0][pool[1]]
        pool[1] += 1
        return row, 1.0 - dp.drop_prob

    def _forward_train(self, point: Point):
        """Training: the whole block as ONE taped Function (ptv3_hip.autograd.BlockFn) - same kernels and statement
        order as _forward_generic, without a trip through the autograd engine per layer."""
        spt = point.sparse_conv_feat
        feat = point.feat
        conv_feat = None if spt.features is feat else spt.features
        mlp = self.mlp[0]
        K = self.attn.resolve_patch_size(point)
        wo, wi = self.attn.window_maps(point)
        params = (self.cpe[0].weight, self.cpe[0].bias, self.cpe[1].weight, self.cpe[1].bias, self.cpe[2].weight,
                  self.cpe[2].bias, self.norm1[0].weight, self.norm1[0].bias, self.attn.qkv.weight, self.attn.qkv.bias,
                  self.attn.proj.weight, self.attn.proj.bias, self.norm2[0].weight, self.norm2[0].bias,
                  mlp.fc1.weight, mlp.fc1.bias, mlp.fc2.weight, mlp.fc2.bias)
        # the two DropPath draws in the reference's order: attention branch, then MLP branch
        mask1 = self._drop_draw(point)
        mask2 = self._drop_draw(point)
        out = A.block(feat, conv_feat, params, spt.neighbors(3, self.cpe[0].indice_key), spt.row_order, wo, wi,
                      self.attn.num_heads, K, self.attn.scale, mask1, mask2, self.cpe[2].eps,
                      self.attn.window_cu(point))
        point.feat = out
        point.sparse_conv_feat = spt.replace_feature(out)
        return point

    def forward(self, point: Point):
        if self.training and self._train_fusable() and point.feat.shape[1] % ops.k_granule(point.feat.dtype) == 0 \
                and self.attn.qkv.bias is not None:
            return self._forward_train(point)
        if self.training or not self._fusable():
            return self._forward_generic(point)
        mlp = self.mlp[0]
        if (ops.block_fusable(self.channels, mlp.fc1.out_features, point.feat.dtype, point.feat.shape[0]) and isinstance(mlp.act, nn.GELU)
                and self.cpe[2].eps == self.norm1[0].eps == self.norm2[0].eps):
            return self._forward_fused_kernels(point)
        # ---- fused eval path: 9 launches per block, residual adds and norms folded into epilogues
        shortcut = point.feat
        spt = point.sparse_conv_feat                         # xCPE conv reads the sparse tensor's features
        wf, bf = self.folded_cpe(spt.features.dtype)
        nbr = spt.neighbors(3, self.cpe[0].indice_key)
        g1, b1 = self.cpe[2].affine_f32()
        g2, b2 = self.norm1[0].affine_f32()
        slabs = ops.conv_slabs(spt.features, wf, nbr, 27, spt.row_order)
        # whole-row linears with their LayerNorms folded in (ptv3_rows_linear; the executor takes the same branches)
        rows = self._rows_path(point.feat)
        dt = spt.features.dtype
        if slabs is not None:   # deep levels: the conv splits over K; the LayerNorm kernel sums the slabs
            feat, x = ops.layernorm_slabs(slabs[0], slabs[1], nbr.shape[0], wf.shape[0], bf, spt.features.dtype, g1, b1,
                                          self.cpe[2].eps, res=shortcut, gamma2=g2, beta2=b2)
            qkv = self.attn.qkv(x)
        elif rows:
            x = ops.gemm(spt.features, wf, bias=bf, nbr=nbr, kvol=27, row_order=spt.row_order)
            feat, qkv = ops.rows_linear(x, self.attn.qkv.weight_for(dt), self.attn.qkv.bias_f32(), ln=(g2, b2), ln0=(g1, b1),
                                        shortcut=shortcut, eps=self.cpe[2].eps)
        else:
            x = ops.gemm(spt.features, wf, bias=bf, nbr=nbr, kvol=27, row_order=spt.row_order)
            feat, x = ops.layernorm(x, g1, b1, self.cpe[2].eps, res=shortcut, gamma2=g2, beta2=b2)
            qkv = self.attn.qkv(x)
        x = self.attn.attention_core(point, qkv)
        if rows:
            feat = ops.rows_linear(x, self.attn.proj.weight_for(dt), self.attn.proj.bias_f32(), res=feat)
            g3, b3 = self.norm2[0].affine_f32()
            h = ops.rows_linear(feat, mlp.fc1.weight_for(dt), mlp.fc1.bias_f32(), act=ops.ACT_GELU, ln=(g3, b3),
                                eps=self.norm2[0].eps)
            feat = mlp.fc2(h, res=feat)
        else:
            feat = self.attn.proj(x, res=feat)                  # + shortcut
            x = self.norm2[0](feat)
            feat = self.mlp[0](x, res=feat)                     # fc1+GELU, fc2 + shortcut
        point.feat = feat
        point.sparse_conv_feat = point.sparse_conv_feat.replace_feature(feat)
        return point

    def _forward_generic(self, point: Point):
        """The reference's statement order (:318-338) on the unfused layer ops."""
        shortcut = point.feat
        point = self.cpe(point)
        point.feat = _add(shortcut, point.feat)
        shortcut = point.feat
        if self.pre_norm:
            point = self.norm1(point)
        point = self.drop_path(self.attn(point))
        point.feat = _add(shortcut, point.feat)
        if not self.pre_norm:
            point = self.norm1(point)
        shortcut = point.feat
        if self.pre_norm:
            point = self.norm2(point)
        point = self.drop_path(self.mlp(point))
        point.feat = _add(shortcut, point.feat)
        if not self.pre_norm:
            point = self.norm2(point)
        point.sparse_conv_feat = point.sparse_conv_feat.replace_feature(point.feat)
        return point


def _add(a, b):
    # residual add outside a fused epilogue: (m, c) elementwise, index-free -> torch plumbing
    return a + b


class SerializedPooling(PointModule):
    def __init__(self, in_channels, out_channels, stride=2, norm_layer=None, act_layer=None, reduce="max",
                 shuffle_orders=True, traceable=True):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        assert stride == 2 ** (math.ceil(stride) - 1).bit_length()  # 2, 4, 8
        self.stride = stride
        assert reduce in ["sum", "mean", "min", "max"]
        if reduce != "max":
            raise NotImplementedError("SerializedPooling on the HIP path implements reduce='max' (every PTv3 config)")
        self.reduce = reduce
        self.shuffle_orders = shuffle_orders
        self.traceable = traceable
        self.proj = Linear(in_channels, out_channels)
        self.norm = PointSequential(norm_layer(out_channels)) if norm_layer is not None else None
        self.act = PointSequential(act_layer()) if act_layer is not None else None

    def _pooling_depth(self, serialized_depth):
        pooling_depth = (math.ceil(self.stride) - 1).bit_length()
        return 0 if pooling_depth > serialized_depth else pooling_depth

    def plan_geometry(self, geo):
        """Everything of this stage that depends on coordinates only: clusters, the pooled coordinates / codes and
        their serialized order.  `geo`: mapping with serialized_code / _order / _depth, grid_coord, batch, offset,
        optionally coord.  A training forward computes the plans of ALL stages before the first feature kernel
        (PointTransformerV3._plan_pooling): the host sync inside pool_segments then meets a near-empty queue instead of
        draining the feature pipeline once per stage.  Draws the order shuffle (:408-412) - same CPU RNG sequence as
        computing the stages one after the other."""
        depth_in = geo["serialized_depth"]
        pooling_depth = self._pooling_depth(depth_in)
        code = geo["serialized_code"]
        order0 = geo["serialized_order"][0]
        nb = len(geo["offset"])
        batch = geo["batch"].long().contiguous()
        cluster, seg_start, n_out, pooled_offset, pooled_offset_host = ops.pool_segments(
            code[0], order0, pooling_depth * 3, batch=batch, num_scenes=nb)
        perm = torch.randperm(code.shape[0]).tolist() if self.shuffle_orders else None
        coord, grid_coord, batch_out, code_out = ops.pool_geometry(
            geo["coord"].float().contiguous() if geo.get("coord") is not None else None,
            geo["grid_coord"].long().contiguous(), batch, code, order0, seg_start, n_out, pooling_depth, row_perm=perm)
        depth = depth_in - pooling_depth
        end_bit = max(1, depth * 3 + max(nb - 1, 0).bit_length())
        order, inverse = ops.argsort_codes(code_out, end_bit)
        return dict(cluster=cluster, seg_start=seg_start, n_out=n_out, order0=order0, pooling_depth=pooling_depth,
                    child=dict(coord=coord, grid_coord=grid_coord, batch=batch_out, serialized_code=code_out,
                               serialized_order=order, serialized_inverse=inverse, serialized_depth=depth,
                               offset=pooled_offset, _offset_host=pooled_offset_host))

    def _forward_planned(self, point: Point, plans):
        """Training forward on a precomputed geometry plan: projection, taped segment max, norm / act."""
        plan, child = plans[0], plans[0]["child"]
        feat = A.segment_max(self.proj(point.feat), plan["order0"], plan["seg_start"], plan["n_out"])
        point_dict = Point(feat=feat, **{k: v for k, v in child.items() if v is not None})
        if "_grid_max_host" in point.keys():
            point_dict["_grid_max_host"] = [g >> plan["pooling_depth"] for g in point["_grid_max_host"]]
        for key in ("condition", "context"):
            if key in point.keys():
                point_dict[key] = point[key]
        if self.traceable:
            point_dict["pooling_inverse"] = plan["cluster"]
            point_dict["pooling_parent"] = point
            point_dict["_pool_segments"] = (plan["order0"], plan["seg_start"])
        if len(plans) > 1:
            point_dict["_pool_plans"] = plans[1:]
        point = point_dict
        if self.norm is not None:
            point = self.norm(point)
        if self.act is not None:
            point = self.act(point)
        point.sparsify()
        return point

    def forward(self, point: Point):
        if self.training and "_pool_plans" in point.keys():
            return self._forward_planned(point, point.pop("_pool_plans"))
        pooling_depth = self._pooling_depth(point.serialized_depth)
        assert {"serialized_code", "serialized_order", "serialized_inverse", "serialized_depth"}.issubset(
            point.keys()), "Run point.serialization() point cloud before SerializedPooling"
        code = point.serialized_code
        order0 = point.serialized_order[0]
        # clusters = runs of equal (code[0] >> 3*pooling_depth) along serialized order 0
        nb = len(point.offset)
        cluster, seg_start, n_out, pooled_offset, pooled_offset_host = ops.pool_segments(
            code[0], order0, pooling_depth * 3, batch=point.batch.long().contiguous(), num_scenes=nb)
        k = code.shape[0]
        if self.shuffle_orders:
            # same CPU-RNG draw as the reference (:408-412); applied to the source rows so the pooled codes
            # come out already permuted
            perm = torch.randperm(k).tolist()
        else:
            perm = None
        # folded BN + activation ride on the segmented max when they are the standard eval layers
        bn = self.norm[0] if self.norm is not None and len(self.norm) == 1 and isinstance(self.norm[0], BatchNorm1d) else None
        act_id = ops.ACT_NONE
        fuse = (self.norm is None or bn is not None) and (self.act is None or isinstance(self.act[0], nn.GELU))
        fuse = fuse and not self.training   # training: BN needs the batch statistics of the pooled rows
        if fuse and self.act is not None:
            act_id = ops.ACT_GELU
        scale = shift = None
        if fuse and bn is not None:
            scale, shift = bn.folded()
        proj = self.proj(point.feat)
        feat, coord, grid_coord, batch, code_out = ops.pool_reduce(
            proj.detach(), point.coord.float().contiguous() if "coord" in point.keys() else None,
            point.grid_coord.long().contiguous(), point.batch.long().contiguous(), code, order0, seg_start,
            n_out, pooling_depth, bn_scale=scale, bn_shift=shift, act=act_id if fuse else ops.ACT_NONE,
            row_perm=perm)
        if self.training:
            feat = A.segment_max(proj, order0, seg_start, n_out)   # taped twin of the max above
        depth = point.serialized_depth - pooling_depth
        end_bit = max(1, depth * 3 + max(nb - 1, 0).bit_length())
        order, inverse = ops.argsort_codes(code_out, end_bit)
        point_dict = Point(
            feat=feat, coord=coord, grid_coord=grid_coord, serialized_code=code_out, serialized_order=order,
            serialized_inverse=inverse, serialized_depth=depth, batch=batch, offset=pooled_offset,
            _offset_host=pooled_offset_host,
        )
        if coord is None:
            del point_dict["coord"]
        if "_grid_max_host" in point.keys():
            point_dict["_grid_max_host"] = [g >> pooling_depth for g in point["_grid_max_host"]]
        if "condition" in point.keys():
            point_dict["condition"] = point.condition
        if "context" in point.keys():
            point_dict["context"] = point.context
        if self.traceable:
            point_dict["pooling_inverse"] = cluster
            point_dict["pooling_parent"] = point
            point_dict["_pool_segments"] = (order0, seg_start)   # for the deterministic gather backward
        point = point_dict
        if not fuse:
            if self.norm is not None:
                point = self.norm(point)
            if self.act is not None:
                point = self.act(point)
        point.sparsify()
        return point


class SerializedUnpooling(PointModule):
    def __init__(self, in_channels, skip_channels, out_channels, norm_layer=None, act_layer=None,
                 traceable=False):
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

    @staticmethod
    def _epilogue(seq):
        """(bn_scale, bn_shift, act) when `seq` is Linear [+ eval BatchNorm1d] [+ GELU], else None."""
        mods = list(seq._modules.values())
        if not isinstance(mods[0], Linear):
            return None
        scale = shift = None
        act = ops.ACT_NONE
        rest = mods[1:]
        if rest and isinstance(rest[0], BatchNorm1d):
            scale, shift = rest[0].folded()
            rest = rest[1:]
        if rest and isinstance(rest[0], nn.GELU):
            act = ops.ACT_GELU
            rest = rest[1:]
        if rest:
            return None
        return scale, shift, act

    def forward(self, point):
        assert "pooling_parent" in point.keys()
        assert "pooling_inverse" in point.keys()
        parent = point.pop("pooling_parent")
        inverse = point.pop("pooling_inverse")
        segments = point.pop("_pool_segments", None)
        e1, e2 = (None, None) if self.training else (self._epilogue(self.proj), self._epilogue(self.proj_skip))
        if e1 is not None and e2 is not None:
            # two GEMMs: up-branch, then skip-branch whose epilogue gathers the up-branch rows by cluster id
            up = self.proj[0](point.feat, bn_scale=e1[0], bn_shift=e1[1], act=e1[2])
            point.feat = up
            skip, fused = self.proj_skip[0](parent.feat, bn_scale=e2[0], bn_shift=e2[1], act=e2[2], res=up,
                                            res_index=inverse.int(), dual=True)
            # The reference refreshes parent.sparse_conv_feat inside proj_skip (modules.py:97-103) but NOT
            # after `parent.feat = parent.feat + point.feat[inverse]` (:478): the next Block's xCPE conv
            # therefore sees the skip branch alone.  Reproduced on purpose.
            parent.sparse_conv_feat = parent.sparse_conv_feat.replace_feature(skip)
            parent.feat = fused
        else:
            point = self.proj(point)
            parent = self.proj_skip(parent)
            if self.training and segments is not None:
                parent.feat = parent.feat + A.cluster_gather(point.feat, inverse, segments[0], segments[1])
            else:
                parent.feat = parent.feat + point.feat[inverse]
        if self.traceable:
            parent["unpooling_parent"] = point
        return parent


class Embedding(PointModule):
    def __init__(self, in_channels, embed_channels, norm_layer=None, act_layer=None):
        super().__init__()
        self.in_channels = in_channels
        self.embed_channels = embed_channels
        self.stem = PointSequential(conv=SubMConv3d(in_channels, embed_channels, kernel_size=5, padding=1,
                                                    bias=False, indice_key="stem"))
        if norm_layer is not None:
            self.stem.add(norm_layer(embed_channels), name="norm")
        if act_layer is not None:
            self.stem.add(act_layer(), name="act")

    def forward(self, point: Point):
        mods = self.stem._modules
        bn, act = mods.get("norm"), mods.get("act")
        if not self.training and (bn is None or isinstance(bn, BatchNorm1d)) and (act is None or isinstance(act, nn.GELU)):
            scale, shift = bn.folded() if bn is not None else (None, None)
            sp = mods["conv"](point.sparse_conv_feat, bn_scale=scale, bn_shift=shift,
                              act=ops.ACT_GELU if act is not None else ops.ACT_NONE)
            point.sparse_conv_feat = sp
            point.feat = sp.features
            return point
        return self.stem(point)


@MODELS.register_module("PT-v3m1")
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
        pre_norm=True,
        shuffle_orders=True,
        enable_rpe=False,
        enable_flash=True,
        upcast_attention=False,
        upcast_softmax=False,
        enc_mode=False,
        pdnorm_bn=False,
        pdnorm_ln=False,
        pdnorm_decouple=True,
        pdnorm_adaptive=False,
        pdnorm_affine=True,
        pdnorm_conditions=("ScanNet", "S3DIS", "Structured3D"),
    ):
        super().__init__()
        self.num_stages = len(enc_depths)
        self.order = [order] if isinstance(order, str) else order
        self.enc_mode = enc_mode
        self.shuffle_orders = shuffle_orders
        # None: follow torch autocast (bf16) else fp32; or force torch.float32 / torch.bfloat16
        self.compute_dtype = None
        # True: one ptv3_forward call per forward (native executor) whenever the tree is the standard eval
        # configuration and no forward hooks are attached; False: module-by-module (same kernels, same results)
        self.use_engine = True
        # True: the caller guarantees grid_coord / batch / offset were materialised before the call (not
        # produced on the current stream just now); consecutive forwards then pipeline (geometry of call i+1
        # under the feature tail of call i).  Default False = fully stream-ordered.
        self.inputs_resident = False
        # True (with inputs_resident, native executor only): throughput mode - the whole feature pipeline runs on
        # executor-owned streams so consecutive forwards overlap (deep, latency-bound levels of call i under the
        # level-0 kernels of call i+1).  Outputs then live in a ring of three executor-owned buffers: consume the
        # result of call i before issuing call i+2 (ptv3_forward_io.overlap_calls in include/ptv3_hip.h).
        self.overlap_calls = False

        assert self.num_stages == len(stride) + 1
        assert self.num_stages == len(enc_depths)
        assert self.num_stages == len(enc_channels)
        assert self.num_stages == len(enc_num_head)
        assert self.num_stages == len(enc_patch_size)
        assert self.enc_mode or self.num_stages == len(dec_depths) + 1
        assert self.enc_mode or self.num_stages == len(dec_channels) + 1
        assert self.enc_mode or self.num_stages == len(dec_num_head) + 1
        assert self.enc_mode or self.num_stages == len(dec_patch_size) + 1

        if pdnorm_bn or pdnorm_ln:
            raise NotImplementedError("PDNorm (pdnorm_bn / pdnorm_ln) is off in every target config and is not "
                                      "part of the MI355X path (SURVEY.md section 2a row 10)")
        bn_layer = partial(BatchNorm1d, eps=1e-3, momentum=0.01)
        ln_layer = LayerNorm
        act_layer = GELU

        self.embedding = Embedding(in_channels=in_channels, embed_channels=enc_channels[0], norm_layer=bn_layer,
                                   act_layer=act_layer)

        enc_drop_path = [x.item() for x in torch.linspace(0, drop_path, sum(enc_depths))]
        self.enc = PointSequential()
        for s in range(self.num_stages):
            enc_drop_path_ = enc_drop_path[sum(enc_depths[:s]): sum(enc_depths[: s + 1])]
            enc = PointSequential()
            if s > 0:
                enc.add(SerializedPooling(in_channels=enc_channels[s - 1], out_channels=enc_channels[s],
                                          stride=stride[s - 1], norm_layer=bn_layer, act_layer=act_layer,
                                          shuffle_orders=True), name="down")
            for i in range(enc_depths[s]):
                enc.add(Block(channels=enc_channels[s], num_heads=enc_num_head[s], patch_size=enc_patch_size[s],
                              mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                              proj_drop=proj_drop, drop_path=enc_drop_path_[i], norm_layer=ln_layer,
                              act_layer=act_layer, pre_norm=pre_norm, order_index=i % len(self.order),
                              cpe_indice_key=f"stage{s}", enable_rpe=enable_rpe, enable_flash=enable_flash,
                              upcast_attention=upcast_attention, upcast_softmax=upcast_softmax),
                        name=f"block{i}")
            if len(enc) != 0:
                self.enc.add(module=enc, name=f"enc{s}")

        if not self.enc_mode:
            dec_drop_path = [x.item() for x in torch.linspace(0, drop_path, sum(dec_depths))]
            self.dec = PointSequential()
            dec_channels = list(dec_channels) + [enc_channels[-1]]
            for s in reversed(range(self.num_stages - 1)):
                dec_drop_path_ = dec_drop_path[sum(dec_depths[:s]): sum(dec_depths[: s + 1])]
                dec_drop_path_.reverse()
                dec = PointSequential()
                dec.add(SerializedUnpooling(in_channels=dec_channels[s + 1], skip_channels=enc_channels[s],
                                            out_channels=dec_channels[s], norm_layer=bn_layer,
                                            act_layer=act_layer), name="up")
                for i in range(dec_depths[s]):
                    dec.add(Block(channels=dec_channels[s], num_heads=dec_num_head[s],
                                  patch_size=dec_patch_size[s], mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                                  qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=proj_drop,
                                  drop_path=dec_drop_path_[i], norm_layer=ln_layer, act_layer=act_layer,
                                  pre_norm=pre_norm, order_index=i % len(self.order),
                                  cpe_indice_key=f"stage{s}", enable_rpe=enable_rpe, enable_flash=enable_flash,
                                  upcast_attention=upcast_attention, upcast_softmax=upcast_softmax),
                            name=f"block{i}")
                self.dec.add(module=dec, name=f"dec{s}")

    _warned_fp16 = False

    def resolve_dtype(self):
        if self.compute_dtype is not None:
            self.effective_dtype = self.compute_dtype
            return self.compute_dtype
        if torch.is_autocast_enabled():
            # amp_dtype="float16" is the reference's default (configs/_base_/default_runtime.py:19) and means "16-bit
            # matmuls": the MI355X kernels take the request as bfloat16 - the same storage width with fp32's exponent
            # range, so the GradScaler the trainer wraps around fp16 runs (engines/train.py:201-241) never sees an
            # overflow - and hand results back in fp32 at the model boundary (pred / seg_logits / loss).
            dt = torch.get_autocast_gpu_dtype()
            if dt == torch.float16 and not PointTransformerV3._warned_fp16:
                # said once, and visible afterwards as `backbone.effective_dtype` (what a trainer can log): the arithmetic
                # is NOT what the config names - 8 mantissa bits instead of 11 - and a GradScaler has nothing to protect
                import warnings
                warnings.warn("PT-v3m1 (MI355X HIP path): autocast float16 is computed in bfloat16 (same width, fp32 exponent "
                              "range); set amp_dtype='bfloat16' to make the config say what runs", stacklevel=2)
                PointTransformerV3._warned_fp16 = True
            self.effective_dtype = torch.bfloat16 if dt in (torch.bfloat16, torch.float16) else torch.float32
            return self.effective_dtype
        self.effective_dtype = torch.float32
        return torch.float32

    def _plan_pooling(self, point):
        """geometry of every pooling stage of the encoder, chained, before any feature kernel is queued"""
        pools = [m for m in self.enc.modules() if isinstance(m, SerializedPooling)]
        geo = {k: point[k] for k in ("serialized_code", "serialized_order", "serialized_depth", "grid_coord", "batch",
                                     "offset")}
        geo["coord"] = point["coord"] if "coord" in point.keys() else None
        plans = []
        for m in pools:
            plans.append(m.plan_geometry(geo))
            geo = plans[-1]["child"]
        if plans:
            point["_pool_plans"] = plans

    def forward(self, data_dict, _head=None):
        check_sync_batchnorm(self)  # sync_bn=True: torch converted the BatchNorm1d modules, take them back
        # eval: no autograd tape (fused kernels / native executor); train: one taped Function per layer
        with torch.set_grad_enabled(self.training and torch.is_grad_enabled()):
            use_engine = not self.training and self.use_engine and _engine.eligible(self, _head)
            if use_engine and "batch" not in data_dict and "offset" in data_dict and not isinstance(data_dict, Point):
                # the executor derives the batch ids on its geometry stream: hand it an uninitialised buffer
                data_dict = dict(data_dict)
                data_dict["batch"] = torch.empty(data_dict["feat"].shape[0], dtype=torch.long,
                                                 device=data_dict["feat"].device)
                data_dict["_batch_pending"] = True
            point = Point(data_dict)
            dtype = self.resolve_dtype()
            feat = point.feat
            if feat.dtype not in (torch.float32, torch.bfloat16):
                feat = feat.float()
            overlap = use_engine and self.overlap_calls and self.inputs_resident
            point.feat = feat.contiguous() if overlap else ops.cast(feat.contiguous(), dtype)
            if use_engine:
                point._ensure_grid_coord()
                point, head_out = _engine.forward(self, point, dtype, _head)
                if _head is not None:
                    point["_head_out"] = head_out
                return point
            point.serialization(order=self.order, shuffle_orders=self.shuffle_orders)
            if self.training:
                self._plan_pooling(point)
            point.sparsify()
            point = self.embedding(point)
            point = self.enc(point)
            if not self.enc_mode:
                point = self.dec(point)
        return point
