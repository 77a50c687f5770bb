"""torch.nn layer classes whose forward runs in libptv3_hip.so.

They subclass the torch classes the reference instantiates (nn.Linear, nn.LayerNorm, nn.BatchNorm1d,
nn.GELU, nn.ReLU) so parameters, buffers, state_dict keys and isinstance checks are unchanged; only
the arithmetic moves to the HIP kernels (ptv3_gemm / ptv3_layernorm / ptv3_affine_act).
Inference only in this round: training-mode statistics / autograd are SURVEY section 8 row f1.
"""
import torch
import torch.nn as nn

from ptv3_hip import ops
from .sparse import _ParamCache


def _f32(p):
    return None if p is None else p.detach().float().contiguous()


def bn_fold(bn, cache):
    """eval BatchNorm1d -> per-channel (scale, shift), fp32."""
    def make():
        scale = bn.weight.detach().float() * torch.rsqrt(bn.running_var.float() + bn.eps)
        shift = bn.bias.detach().float() - bn.running_mean.float() * scale
        return scale.contiguous(), shift.contiguous()
    return cache.get("bn", [bn.weight, bn.bias, bn.running_mean, bn.running_var], make)


def _no_training(m):
    if m.training:
        raise NotImplementedError(
            f"{type(m).__name__}: the HIP path implements the eval-mode forward only (call model.eval()); "
            "training statistics and backward are SURVEY.md section 8 row f1")


class Linear(nn.Linear):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self._cache = _ParamCache()

    def weight_for(self, dtype):
        return self._cache.get(("w", dtype), [self.weight], lambda: self.weight.detach().to(dtype).contiguous())

    def bias_f32(self):
        if self.bias is None:
            return None
        return self._cache.get("b", [self.bias], lambda: _f32(self.bias))

    def forward(self, x, **epilogue):
        pad = (-x.shape[1]) % ops.k_granule(x.dtype)
        w = self.weight_for(x.dtype)
        if pad:  # 16-byte K granularity of the kernel
            x = torch.nn.functional.pad(x, (0, pad)).contiguous()
            w = torch.nn.functional.pad(w, (0, pad)).contiguous()
        return ops.gemm(x, w, bias=self.bias_f32(), **epilogue)


class LayerNorm(nn.LayerNorm):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self._cache = _ParamCache()

    def affine_f32(self):
        return self._cache.get("gb", [self.weight, self.bias], lambda: (_f32(self.weight), _f32(self.bias)))

    def forward(self, x, res=None):
        g, b = self.affine_f32()
        return ops.layernorm(x, g, b, self.eps, res=res)


class BatchNorm1d(nn.BatchNorm1d):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self._cache = _ParamCache()

    def folded(self):
        return bn_fold(self, self._cache)

    def forward(self, x, act=ops.ACT_NONE):
        _no_training(self)
        s, t = self.folded()
        return ops.affine_act(x, s, t, act)


class GELU(nn.GELU):
    def forward(self, x):
        return ops.affine_act(x, None, None, ops.ACT_GELU)


class ReLU(nn.ReLU):
    def forward(self, x):
        return ops.affine_act(x, None, None, ops.ACT_RELU)


class DropPath(nn.Module):
    """timm.layers.DropPath: identity in eval mode (the only mode of this round)."""

    def __init__(self, drop_prob=0.0, scale_by_keep=True):
        super().__init__()
        self.drop_prob, self.scale_by_keep = drop_prob, scale_by_keep

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        _no_training(self)

    def extra_repr(self):
        return f"drop_prob={round(self.drop_prob, 3):0.3f}"
