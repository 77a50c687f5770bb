"""Submanifold sparse convolution on MI355X: the stand-in for `spconv.pytorch` on the PTv3 path.

The reference builds `spconv.SparseConvTensor` in Point.sparsify (models/utils/structure.py:111-146)
and runs `spconv.SubMConv3d` in Embedding / Block.cpe (point_transformer_v3m1_base.py:277-284,499-506).
spconv 2.3.6 is a CUDA-only wheel; here the same two names are backed by libptv3_hip.so:
a site hash + neighbour table per (indice_key, kernel_size) and the implicit-GEMM kernel.
Weight layout (out, k, k, k, in) and correlation offset order follow spconv 2.x ("parity
unpinned" for loading real spconv checkpoints - the wheel cannot be run here).
"""
import math

import torch
import torch.nn as nn

from ptv3_hip import ops
from ptv3_hip import autograd as A


class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size, _shared=None):
        self.features = features
        self.indices = indices  # (n, 4) int32 [batch, x, y, z]
        self.spatial_shape = spatial_shape
        self.batch_size = batch_size
        # neighbour tables are shared by every tensor derived through replace_feature (spconv's indice_dict)
        self._shared = _shared if _shared is not None else {"table": None, "nbr": {}, "row_order": None}

    def replace_feature(self, feature):
        return SparseConvTensor(feature, self.indices, self.spatial_shape, self.batch_size, self._shared)

    def neighbors(self, ksize, indice_key=None):
        key = (indice_key, ksize) if indice_key is not None else ("_k", ksize)
        if key not in self._shared["nbr"]:
            nbr, table = ops.subm_neighbors(self.indices, ksize, self._shared["table"])
            self._shared["table"] = table
            self._shared["nbr"][key] = nbr
        return self._shared["nbr"][key]

    @property
    def row_order(self):
        return self._shared["row_order"]

    @row_order.setter
    def row_order(self, v):
        self._shared["row_order"] = v


class _ParamCache:
    """Casts / re-lays-out parameters once per parameter version (eval: once)."""

    def __init__(self):
        self._c = {}

    def get(self, key, params, fn):
        ver = tuple((p.data_ptr(), p._version) for p in params)
        hit = self._c.get(key)
        if hit is None or hit[0] != ver:
            with torch.no_grad():
                hit = (ver, fn())
            self._c[key] = hit
        return hit[1]


class SubMConv3d(nn.Module):
    """spconv.pytorch.SubMConv3d(in, out, kernel_size, bias=..., indice_key=...) on the HIP path."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, indice_key=None, algo=None, **kwargs):
        super().__init__()
        assert stride == 1 and dilation == 1 and groups == 1, "only the configuration PTv3 uses"
        assert isinstance(kernel_size, int) and kernel_size % 2 == 1
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.indice_key = indice_key
        k = kernel_size
        self.weight = nn.Parameter(torch.empty(out_channels, k, k, k, in_channels))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()
        self._cache = _ParamCache()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.in_channels * self.kernel_size ** 3
            bound = 1 / math.sqrt(fan_in)
            nn.init.uniform_(self.bias, -bound, bound)

    def _weight_for(self, dtype, cin_pad):
        def make():
            w = self.weight.detach()
            if cin_pad != self.in_channels:
                w = torch.nn.functional.pad(w, (0, cin_pad - self.in_channels))
            return w.reshape(self.out_channels, -1).to(dtype).contiguous()
        return self._cache.get(("w", dtype, cin_pad), [self.weight], make)

    def forward(self, x: SparseConvTensor, bn_scale=None, bn_shift=None, act=ops.ACT_NONE):
        if self.training:
            if bn_scale is not None or act != ops.ACT_NONE:
                raise NotImplementedError("training-mode SubMConv3d: BN / activation epilogues are eval-only fusions")
            nbr = x.neighbors(self.kernel_size, self.indice_key)
            return x.replace_feature(A.subm_conv(x.features, self.weight, self.bias, nbr, x.row_order))
        feat = x.features
        cin = feat.shape[1]
        gran = ops.k_granule(feat.dtype)
        cin_pad = (cin + gran - 1) // gran * gran
        if cin_pad != cin:  # kernel wants 16-byte K granularity: zero-pad features and weights alike
            feat = torch.nn.functional.pad(feat, (0, cin_pad - cin)).contiguous()
        nbr = x.neighbors(self.kernel_size, self.indice_key)
        w = self._weight_for(feat.dtype, cin_pad)
        bias = None if self.bias is None else self.bias.detach().float()
        out = ops.gemm(feat, w, bias=bias, nbr=nbr, kvol=self.kernel_size ** 3, row_order=x.row_order,
                       bn_scale=bn_scale, bn_shift=bn_shift, act=act)
        return x.replace_feature(out)


def is_spconv_module(module):
    return isinstance(module, SubMConv3d)
