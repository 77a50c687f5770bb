"""Build-container-only helper: import the reference's hot-path Python files in place.

Used ONLY by tests/golden/make_golden.py (which writes the committed *.npz fixtures) and by
tests that are skipped when /root/reference is absent.  Nothing from /root/reference is
copied; the files are imported where they lie.  The reference's package __init__ files pull
in CUDA-only wheels (spconv, MinkowskiEngine, ocnn, pointops, torch_scatter ...), so bare
package objects are pre-registered and four tiny third-party stubs are installed:

  addict.Dict                      attribute dict
  spconv.pytorch.SubMConv3d        -> oracle.ptv3.subm_conv3d   (our restatement; parity unpinned)
  spconv.pytorch.SparseConvTensor  plain container
  torch_scatter.segment_csr        -> torch.segment_reduce
  timm.layers.DropPath             identity in eval mode
  flash_attn.flash_attn_varlen_qkvpacked_func -> oracle.ptv3.varlen_attention (our restatement of the
                                   published function; returns its input dtype like the library does)
  torch_cluster, peft              bare modules (imported by models/default.py, unused by DefaultSegmentorV2)
"""
import importlib
import os
import sys
import types

import torch
import torch.nn as nn

REF = os.environ.get("PTV3_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF, "pointcept", "models"))


class _Dict(dict):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __delattr__(self, k):
        del self[k]


def _install_stubs():
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(os.path.dirname(here))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import ptv3 as oracle_ptv3

    addict = types.ModuleType("addict")
    addict.Dict = _Dict
    sys.modules["addict"] = addict

    class SparseConvTensor:
        def __init__(self, features, indices, spatial_shape, batch_size):
            self.features, self.indices = features, indices
            self.spatial_shape, self.batch_size = spatial_shape, batch_size

        def replace_feature(self, feat):
            return SparseConvTensor(feat, self.indices, self.spatial_shape, self.batch_size)

    class SubMConv3d(nn.Module):
        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0,
                     dilation=1, groups=1, bias=True, indice_key=None, **kw):
            super().__init__()
            k = kernel_size
            self.weight = nn.Parameter(torch.empty(out_channels, k, k, k, in_channels))
            nn.init.normal_(self.weight, std=(1.0 / (in_channels * k ** 3)) ** 0.5)
            self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
            if bias:
                nn.init.normal_(self.bias, std=0.02)

        def forward(self, x):
            out = oracle_ptv3.subm_conv3d(x.features, x.indices, self.weight, self.bias)
            return x.replace_feature(out)

    spconv = types.ModuleType("spconv")
    sp = types.ModuleType("spconv.pytorch")
    spm = types.ModuleType("spconv.pytorch.modules")
    sp.SubMConv3d, sp.SparseConvTensor = SubMConv3d, SparseConvTensor
    spm.is_spconv_module = lambda m: isinstance(m, SubMConv3d)
    sp.modules = spm
    spconv.pytorch = sp
    sys.modules.update({"spconv": spconv, "spconv.pytorch": sp, "spconv.pytorch.modules": spm})

    ts = types.ModuleType("torch_scatter")

    def segment_csr(src, indptr, reduce="sum"):
        return torch.segment_reduce(src, reduce, lengths=indptr[1:] - indptr[:-1], axis=0, unsafe=True)

    ts.segment_csr = segment_csr
    sys.modules["torch_scatter"] = ts

    timm = types.ModuleType("timm")
    tl = types.ModuleType("timm.layers")

    class DropPath(nn.Module):
        def __init__(self, drop_prob=0.0, scale_by_keep=True):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            assert not self.training, "stub DropPath is eval-only"
            return x

    tl.DropPath = DropPath
    timm.layers = tl
    sys.modules.update({"timm": timm, "timm.layers": tl})

    fa = types.ModuleType("flash_attn")

    def flash_attn_varlen_qkvpacked_func(qkv, cu_seqlens, max_seqlen, dropout_p=0.0, softmax_scale=None, **kw):
        assert dropout_p == 0
        T, three, H, D = qkv.shape
        assert int((cu_seqlens[1:] - cu_seqlens[:-1]).max()) <= max_seqlen
        out = oracle_ptv3.varlen_attention(qkv.float(), cu_seqlens, H, softmax_scale if softmax_scale is not None
                                           else D ** -0.5)
        return out.to(qkv.dtype)

    fa.flash_attn_varlen_qkvpacked_func = flash_attn_varlen_qkvpacked_func
    sys.modules["flash_attn"] = fa
    for name in ("torch_cluster", "peft"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["peft"].LoraConfig = sys.modules["peft"].get_peft_model = None


def _bare_pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


_loaded = {}


def load():
    """Returns a namespace with the reference's hot-path modules."""
    if _loaded:
        return _loaded["ns"]
    assert available(), "reference tree not present"
    _install_stubs()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    pc = os.path.join(REF, "pointcept")
    _bare_pkg("pointcept", pc)
    _bare_pkg("pointcept.models", os.path.join(pc, "models"))
    _bare_pkg("pointcept.engines", os.path.join(pc, "engines"))
    hooks = _bare_pkg("pointcept.engines.hooks", os.path.join(pc, "engines", "hooks"))

    class HookBase:
        pass

    hooks.HookBase = HookBase
    _bare_pkg("pointcept.models.point_transformer_v3", os.path.join(pc, "models", "point_transformer_v3"))
    ppt = _bare_pkg("pointcept.models.point_prompt_training", os.path.join(pc, "models", "point_prompt_training"))

    class PDNorm(nn.Module):
        pass

    ppt.PDNorm = PDNorm
    ns = types.SimpleNamespace()
    ns.utils = importlib.import_module("pointcept.models.utils")
    ns.structure = importlib.import_module("pointcept.models.utils.structure")
    ns.serialization = importlib.import_module("pointcept.models.utils.serialization")
    ns.builder = importlib.import_module("pointcept.models.builder")
    ns.v3m1 = importlib.import_module("pointcept.models.point_transformer_v3.point_transformer_v3m1_base")
    ns.offset_head = importlib.import_module("pointcept.models.offset_keypoint_ptv3")
    ns.Point = ns.structure.Point
    # models/default.py (DefaultSegmentorV2) and the reference's own losses package (plain torch, imported in place)
    ns.losses = importlib.import_module("pointcept.models.losses")
    ns.default = importlib.import_module("pointcept.models.default")
    _loaded["ns"] = ns
    return ns
