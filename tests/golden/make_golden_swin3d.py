"""Build-container-only: list the state_dict of the REFERENCE's Swin3DUNet / OffsetKeypointSwin3D classes (imported
where they lie under /root/reference) for the two configs the reference ships, into
tests/golden/state_dict_swin3d_s3dis.txt and state_dict_offset_swin3d.txt (key, shape, dtype per line).

MinkowskiEngine and microsoft/Swin3D are not in the reference tree, so parameter-only stand-ins are installed for the
import to succeed.  Keys that come from a stand-in rather than from reference code - and therefore pin nothing beyond
MinkowskiEngine's documented parameter layout - are the stem's modules:
    stem_layer.conv_layers.0.kernel          (kernel_size^3, in, out)   MinkowskiConvolution, bias=False
    stem_layer.conv_layers.1.bn.*            MinkowskiBatchNorm wraps torch.nn.BatchNorm1d as `.bn`
    (third listing, stem_transformer=False: the same two kinds inside stem_layer.0 and MinkResBlock's conv1/2, norm1/2)
Every other key (cRSE tables, qkv / proj / mlp, LayerNorms, downsample / upsample linears, classifier, head) is produced
by the reference's own constructors.  No forward is run: the arithmetic of those libraries is not available.
"""
import importlib
import os
import sys
import types

import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_loader  # noqa: E402


def _install_swin_stubs():
    me = types.ModuleType("MinkowskiEngine")

    class MinkowskiConvolution(nn.Module):
        def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False,
                     kernel_generator=None, expand_coordinates=False, convolution_mode=None, dimension=-1):
            super().__init__()
            self.in_channels, self.out_channels = in_channels, out_channels
            kvol = kernel_size ** dimension
            shape = (kvol, in_channels, out_channels) if kvol > 1 else (in_channels, out_channels)
            self.kernel = nn.Parameter(torch.zeros(shape))
            self.bias = nn.Parameter(torch.zeros(1, out_channels)) if bias else None

    class MinkowskiBatchNorm(nn.Module):
        def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
            super().__init__()
            self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum, affine=affine,
                                     track_running_stats=track_running_stats)

    class _NoParams(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    me.MinkowskiConvolution = MinkowskiConvolution
    me.MinkowskiConvolutionTranspose = MinkowskiConvolution
    me.MinkowskiBatchNorm = MinkowskiBatchNorm
    for name in ("MinkowskiReLU", "MinkowskiMaxPooling", "MinkowskiAvgPooling", "MinkowskiPoolingTranspose"):
        setattr(me, name, type(name, (_NoParams,), {}))
    for name in ("SparseTensor", "TensorField", "SparseTensorQuantizationMode", "MinkowskiAlgorithm"):
        setattr(me, name, type(name, (), {}))
    sys.modules["MinkowskiEngine"] = me

    names = ["Swin3D", "Swin3D.sparse_dl", "Swin3D.sparse_dl.attn", "Swin3D.sparse_dl.attn.attn_coff",
             "Swin3D.sparse_dl.knn"]
    mods = {n: types.ModuleType(n) for n in names}
    for n, m in mods.items():
        m.__path__ = []
        sys.modules[n] = m
        if "." in n:
            setattr(mods[n.rsplit(".", 1)[0]], n.rsplit(".", 1)[1], m)
    for name in ("SelfAttnAIOFunction", "PosEmb", "TableDims", "IndexMode", "PrecisionMode"):
        setattr(mods["Swin3D.sparse_dl.attn.attn_coff"], name, type(name, (), {}))
    mods["Swin3D.sparse_dl.knn"].KNN = type("KNN", (), {})
    sys.modules["timm.layers"].trunc_normal_ = torch.nn.init.trunc_normal_


def _cfg(path):
    scope = {}
    exec(compile(open(os.path.join(ref_loader.REF, path)).read(), path, "exec"), scope)
    return scope["model"]


def main():
    ref_loader.load()
    _install_swin_stubs()
    ref_loader._bare_pkg("pointcept.models.swin3d", os.path.join(ref_loader.REF, "pointcept", "models", "swin3d"))
    importlib.import_module("pointcept.models.swin3d.swin3d_v1m1_base")
    importlib.import_module("pointcept.models.offset_keypoint_swin3d")
    from pointcept.models.builder import MODELS
    # third listing: the two constructor variants no shipped config uses (GridDownsample, MinkResBlock stem) on the
    # plumbing-size config of ptv3_hip/configs.py (TINY_SWIN3D_CFG)
    tiny = dict(type="Swin3D-v1m1", in_channels=9, num_classes=13, base_grid_size=0.02, depths=[2, 2, 2],
                channels=[16, 32, 32], num_heads=[2, 2, 2], window_sizes=[5, 7, 7], quant_size=4, drop_path_rate=0.3,
                up_k=3, num_layers=3, stem_transformer=False, down_stride=3, upsample="linear_attn", knn_down=False,
                cRSE="XYZ_RGB_NORM", fp16_mode=1)
    jobs = (("state_dict_swin3d_s3dis.txt", _cfg("configs/s3dis/semseg-swin3d-v1m1-0-small.py")["backbone"]),
            ("state_dict_offset_swin3d.txt", _cfg("configs/my_dataset/offset_keypoint_swin3d.py")),
            ("state_dict_swin3d_tiny_grid_resstem.txt", tiny))
    for fname, cfg in jobs:
        model = MODELS.build(dict(cfg))
        with open(os.path.join(HERE, fname), "w") as f:
            for k, v in model.state_dict().items():
                f.write(f"{k} {tuple(v.shape)} {v.dtype}\n")
        print(fname, len(model.state_dict()))


if __name__ == "__main__":
    main()
