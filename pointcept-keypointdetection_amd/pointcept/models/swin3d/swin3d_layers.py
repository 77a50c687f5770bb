"""Swin3D window attention with cRSE on MI355X: the attention stage only.

Counterpart of the window-attention half of the reference's pointcept/models/swin3d/swin3d_layers.py:
`WindowAttention` (:384-577), `Mlp` (:155-178), `SwinTransformerBlock` (:580-631) keep the reference's constructor
keywords, parameter names and forward signatures; `window_attn_args` is `BasicLayer.get_index01` (:797-824) on plain
tensors instead of MinkowskiEngine sparse tensors, and `WindowStage` is the block loop of `BasicLayer` (:653-699,
846-866: blocks alternate between the regular and the half-window-shifted partition).

PARITY UNPINNED: the reference runs this arithmetic in MinkowskiEngine and microsoft/Swin3D, neither of which is in
its tree; oracle/swin3d.py restates it and is the only checker (see its header for what is an assumption).

Training: every layer is a taped Function with a HIP backward (ptv3_swin_attn_bwd for the attention, the PTv3 path's
Functions for Linear / LayerNorm / GELU).  `attn_drop` is accepted and, as in the reference (the nn.Dropout of :476 is
never applied in its forward), has no effect.
"""
import numpy as np
import torch
import torch.nn as nn

from ptv3_hip import ops
from ptv3_hip import autograd as A
from pointcept.models.utils.hip_layers import Linear, LayerNorm, GELU, DropPath


def sparse_self_attention(w_sizes):
    """The part of sparse_self_attention(..., protocol="v2") (:78-152) the attention kernel consumes: the windows'
    offsets in token space (w2n) and in pair space (w2m).  The per-pair lists (x_offset, y_offset, m2w_indices; sum of
    squared window sizes entries) exist in the reference only to feed SelfAttnAIOFunction's pair-parallel CUDA
    kernel; this kernel walks pairs per window and never materialises them (they are returned as None)."""
    w2n = torch.cumsum(w_sizes, 0) - w_sizes
    sq = w_sizes * w_sizes
    w2m = torch.cumsum(sq, 0) - sq
    return None, None, None, w_sizes, w2n, w2m


def window_attn_args(coords, stride, window_size, local_xyz, signals, shift=0):
    """BasicLayer.get_index01 (:797-824).  coords (n,4) int32 [batch,x,y,z] of the stage's voxels at tensor stride
    `stride`; local_xyz (n,3) fp32 sub-voxel offset of the voxel's averaged point (:855-858); signals (n,3|6) fp32
    colour (and normal); shift in voxels (window_size // 2 for the shifted partition, :826-840).
    -> the reference's attn_args tuple (x_offset, y_offset, m2w_indices, w_sizes, w2n_indices, n2n_indices,
    w2m_indices, n_coords), with the three per-pair lists None."""
    _, w_w_xyz, w_sizes, n2n, _, _ = ops.swin_window_mapping(coords, stride, window_size, shift)
    n_coords = torch.cat([w_w_xyz.float() + local_xyz[n2n], signals[n2n]], dim=1)
    x_off, y_off, m2w, w_sizes, w2n, w2m = sparse_self_attention(w_sizes)
    # the largest window, fetched once per partition (one sync): the attention kernel sizes its LDS by it instead of by
    # window_size^3 occupied cells, which surfaces never reach
    w_sizes._ptv3_max_tokens = int(w_sizes.max()) if w_sizes.numel() else 1
    return x_off, y_off, m2w, w_sizes, w2n, n2n, w2m, n_coords


class WindowAttention(nn.Module):
    """:384-577.  Parameters: qkv, proj, {query,key,value}_{xyz,rgb,norm}_table of shape (3, 2 L, heads, head_dim)."""

    def __init__(self, dim, window_size, quant_size, num_heads, qkv_bias=True, qk_scale=None, attn_drop=0.0,
                 proj_drop=0.0, cRSE="XYZ_RGB", fp16_mode=0):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.color_windowsize = self.normal_windowsize = 2     # colour and normal live in [-1, 1]
        self.fp16_mode, self.cRSE, self.quant_size = fp16_mode, cRSE, quant_size
        self.table_offsets = []
        groups = (("XYZ", "xyz", quant_size, window_size), ("RGB", "rgb", quant_size * 2, self.color_windowsize),
                  ("NORM", "norm", quant_size * 2, self.normal_windowsize))
        self._groups = []
        for key, name, quant, extent in groups:
            if key not in cRSE:
                continue
            setattr(self, {"xyz": "xyz_quant_size", "rgb": "color_quant_size", "norm": "normal_quant_size"}[name], quant)
            shape = (3, 2 * extent * quant, num_heads, head_dim)
            for kind in ("query", "key", "value"):
                p = nn.Parameter(torch.zeros(shape))
                nn.init.trunc_normal_(p, std=0.02)
                setattr(self, f"{kind}_{name}_table", p)
            self.table_offsets += [int(np.prod(shape[1:]))] * 3
            self._groups.append((name, quant))
        self.qkv = Linear(dim, dim * 3, bias=qkv_bias)
        # the reference constructs this module (:476) and never calls it: SelfAttnAIOFunction (:556-569) has no dropout
        # argument, so attn_drop > 0 changes nothing there - nor here
        self.attn_drop = nn.Dropout(attn_drop, inplace=True)
        self.proj = Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop, inplace=True)

    def _tables(self, kind):
        """the concatenated table of :503-528 (a differentiable view of the parameters in training)"""
        parts = [getattr(self, f"{kind}_{name}_table").reshape(-1) for name, _ in self._groups]
        if not self.training:
            parts = [p.detach().float() for p in parts]
        return torch.cat(parts)

    def forward(self, feats, attn_args):
        (_, _, _, w_sizes, w2n, n2n, _, n_coords) = attn_args
        num_v = feats.shape[0]
        hd = self.dim // self.num_heads
        qkv = self.qkv(feats).view(num_v, 3, self.num_heads, hd).permute(1, 0, 2, 3).contiguous()
        query = qkv[0] * self.scale                                                   # :499
        n_crse = torch.cat([n_coords[:, 3 * i:3 * i + 3] * float(q) for i, (_, q) in enumerate(self._groups)],
                           dim=1).float().contiguous()                                # :505-530
        w_start = torch.cat([w2n, w2n.new_tensor([num_v])]).int()
        max_tokens = getattr(w_sizes, "_ptv3_max_tokens", self.window_size ** 3)
        attend = A.swin_attention if self.training else ops.swin_attention      # training: taped, with its HIP backward
        out = attend(query.contiguous(), qkv[1], qkv[2], self._tables("query"), self._tables("key"),
                     self._tables("value"), self.table_offsets, n2n, w_start, n_crse, max_tokens)
        return self.proj_drop(self.proj(out.reshape(num_v, self.dim)))


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=GELU, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x, res=None):
        if isinstance(self.act, nn.GELU) and not self.training:
            return self.fc2(self.fc1(x, act=ops.ACT_GELU), res=res)     # activation / residual in the GEMM epilogue
        x = self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))
        return x if res is None else x + res


class SwinTransformerBlock(nn.Module):
    """:580-631."""

    def __init__(self, dim, num_heads, window_size, quant_size, drop_path=0.0, mlp_ratio=4.0, qkv_bias=True,
                 qk_scale=None, act_layer=GELU, norm_layer=LayerNorm, cRSE="XYZ_RGB", fp16_mode=0):
        super().__init__()
        self.window_size = window_size
        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention(dim, window_size=window_size, quant_size=quant_size, num_heads=num_heads,
                                    qkv_bias=qkv_bias, qk_scale=qk_scale, cRSE=cRSE, fp16_mode=fp16_mode)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer)

    def forward(self, feats, attn_args):
        feats = feats + self.drop_path(self.attn(self.norm1(feats), attn_args))
        if self.training:
            return feats + self.drop_path(self.mlp(self.norm2(feats)))
        return self.mlp(self.norm2(feats), res=feats)


class WindowStage(nn.Module):
    """The attention part of BasicLayer (:653-699 constructor, :846-866 forward) on plain tensors: `blocks`
    alternate between the regular and the shifted window partition.  No downsample (not built)."""

    def __init__(self, dim, depth, num_heads, window_size, quant_size, mlp_ratio=4.0, qkv_bias=True, qk_scale=None,
                 drop_path=0.0, norm_layer=LayerNorm, cRSE="XYZ_RGB", fp16_mode=0):
        super().__init__()
        self.window_size, self.depth, self.dim, self.num_heads = window_size, depth, dim, num_heads
        self.quant_size, self.cRSE, self.fp16_mode = quant_size, cRSE, fp16_mode
        self.shift_size = window_size // 2
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim, num_heads, window_size, quant_size,
                                 drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path,
                                 mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, norm_layer=norm_layer,
                                 cRSE=cRSE, fp16_mode=fp16_mode) for i in range(depth)])

    def forward(self, feats, coords, stride, local_xyz, signals):
        args = window_attn_args(coords, stride, self.window_size, local_xyz, signals, 0)
        args_shift = window_attn_args(coords, stride, self.window_size, local_xyz, signals, self.shift_size)
        for i, blk in enumerate(self.blocks):
            feats = blk(feats, args if i % 2 == 0 else args_shift)
        return feats
