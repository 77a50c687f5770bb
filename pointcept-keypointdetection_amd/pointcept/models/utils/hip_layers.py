"""torch.nn layer classes whose forward runs in libptv3_hip.so.

They subclass the torch classes the reference instantiates (nn.Linear, nn.LayerNorm, nn.BatchNorm1d,
nn.GELU, nn.ReLU) so parameters, buffers, state_dict keys and isinstance checks are unchanged; only
the arithmetic moves to the HIP kernels (ptv3_gemm / ptv3_layernorm / ptv3_affine_act).
`.eval()`: the fused inference kernels (no autograd tape).  `.train()`: one ptv3_hip.autograd Function per layer
(forward and backward kernels of libptv3_hip.so, batch-statistic BatchNorm, DropPath) - SURVEY section 8 f1.
"""
import torch
import torch.nn as nn

from ptv3_hip import ops
from ptv3_hip import autograd as A
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
    """Guard for the few layer variants that exist only as eval-mode kernels (e.g. RPE bias in attention)."""
    if m.training:
        raise NotImplementedError(
            f"{type(m).__name__}: this configuration has an eval-mode forward only on the HIP path")


def _train_epilogue(y, epilogue):
    """The GEMM epilogues the eval kernels fuse, as separate taped ops in training."""
    extra = set(epilogue) - {"act", "res"}
    if extra:
        raise NotImplementedError(f"training-mode Linear: epilogue {sorted(extra)} is an eval-only fusion")
    act = epilogue.get("act", ops.ACT_NONE)
    if act != ops.ACT_NONE:
        y = A.activation(y, act)
    if epilogue.get("res") is not None:
        y = y + epilogue["res"]
    return y


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
        if self.training:
            return _train_epilogue(A.linear(x, self.weight, self.bias), epilogue)
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
        if self.training:
            y = A.layer_norm(x, self.weight, self.bias, self.eps)
            return y if res is None else y + res
        g, b = self.affine_f32()
        return ops.layernorm(x, g, b, self.eps, res=res)


class BatchNorm1d(nn.BatchNorm1d):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self._cache = _ParamCache()

    def folded(self):
        return bn_fold(self, self._cache)

    def forward(self, x, act=ops.ACT_NONE):
        if self.training:
            if self.track_running_stats and self.num_batches_tracked is not None:
                self.num_batches_tracked.add_(1)
            return A.batch_norm_act(x, self, act)
        s, t = self.folded()
        return ops.affine_act(x, s, t, act)


def adopt_sync_batchnorm(root):
    """`tools/train.py` with `sync_bn=True` runs `nn.SyncBatchNorm.convert_sync_batchnorm(model)` (engines/train.py:
    256-257), which replaces every BatchNorm1d of this package by torch's SyncBatchNorm.  Put the HIP BatchNorm1d back -
    around the SAME Parameter / buffer objects, so optimizer, DistributedDataParallel and state_dict keep their
    references - flagged with the process group: its taped Function then all-reduces the batch statistics
    (ptv3_hip/autograd.py BatchNormActFn).  Called by the model wrappers on their first forward and on every training
    forward; returns the number of modules adopted."""
    count = 0
    for parent in list(root.modules()):
        for name, child in list(parent._modules.items()):
            if isinstance(child, nn.SyncBatchNorm):
                bn = BatchNorm1d(child.num_features, eps=child.eps, momentum=child.momentum, affine=child.affine,
                                 track_running_stats=child.track_running_stats)
                bn._parameters, bn._buffers = child._parameters, child._buffers
                bn.sync_group = child.process_group if child.process_group is not None else True
                bn.train(child.training)
                parent._modules[name] = bn
                count += 1
    return count


def check_sync_batchnorm(model):
    """Guard for the model wrappers.  The first forward walks the whole tree (adopt_sync_batchnorm) and remembers the
    slots that hold a BatchNorm; training forwards afterwards only look at those slots - convert_sync_batchnorm swaps
    modules in place, slot by slot - instead of walking ~600 modules per step (1.7 ms of host time per training step of
    the fork model)."""
    state = model.__dict__.get("_sync_bn_slots")
    if state is None:
        adopt_sync_batchnorm(model)
        model.__dict__["_sync_bn_slots"] = [(parent, name) for parent in model.modules()
                                            for name, child in parent._modules.items()
                                            if isinstance(child, nn.modules.batchnorm._BatchNorm)]
        return
    if model.training and any(isinstance(parent._modules.get(name), nn.SyncBatchNorm) for parent, name in state):
        adopt_sync_batchnorm(model)


class GELU(nn.GELU):
    def forward(self, x):
        if self.training:
            return A.activation(x, ops.ACT_GELU)
        return ops.affine_act(x, None, None, ops.ACT_GELU)


class ReLU(nn.ReLU):
    def forward(self, x):
        if self.training:
            return A.activation(x, ops.ACT_RELU)
        return ops.affine_act(x, None, None, ops.ACT_RELU)


class DropPath(nn.Module):
    """timm.layers.DropPath (timm 1.0.22 drop_path): identity in eval mode; in training one Bernoulli draw per
    ROW of the (N, C) feature matrix - PointSequential hands DropPath the point features, so the reference drops
    per point, not per scene (point_transformer_v3m1_base.py:314-316) - scaled by 1/keep_prob."""

    def __init__(self, drop_prob=0.0, scale_by_keep=True):
        super().__init__()
        self.drop_prob, self.scale_by_keep = drop_prob, scale_by_keep

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)  # RNG + mask: torch plumbing
        if keep > 0.0 and self.scale_by_keep:
            mask.div_(keep)
        return x * mask

    def extra_repr(self):
        return f"drop_prob={round(self.drop_prob, 3):0.3f}"
