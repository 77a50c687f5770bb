"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every declared symbol,
the registry / builder / state_dict surface matches the reference, and the product path refuses to run
without a GPU (no CPU fallback)."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ptv3_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ptv3_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from ptv3_hip.lib import lib, SIGNATURES, library_path
    assert os.path.exists(library_path()), "run __graft_entry__.build() first"
    declared = _declared_symbols()
    assert declared, "header parse failed"
    assert set(declared) == set(SIGNATURES), set(declared) ^ set(SIGNATURES)
    dll = lib.load()
    for name in declared:
        assert getattr(dll, name) is not None
    assert dll.ptv3_version() >= 100


def test_argument_validation_without_gpu():
    """Host-side checks run before any launch: bad arguments come back as error codes + message."""
    from ptv3_hip.lib import lib
    rc = lib.ptv3_window_attn_fwd(None, None, None, None, 10, 10, 30, 4, 5, 0.25, None, 0, None)
    assert rc != 0 and b"not divisible" in lib.ptv3_last_error()
    rc = lib.ptv3_gemm(None, None, None, 5, 6, 8, 1, None, None, None, None, None, 0, None, None, None, 0, None, 0,
                       None)
    assert rc != 0 and b"multiple of 4" in lib.ptv3_last_error()
    assert lib.ptv3_gemm_workspace_bytes(100000, 64, 192, 1, 1) == 0      # large M: single pass
    assert lib.ptv3_gemm_workspace_bytes(245, 512, 512, 27, 1) > 0        # deep-stage conv: split over K
    assert lib.ptv3_argsort_workspace_bytes(4, 100000) > 4 * 100000 * 24
    assert lib.ptv3_subm_table_slots(100000) == 262144


def test_no_cpu_fallback():
    from ptv3_hip import ops
    with pytest.raises(RuntimeError, match="GPU tensor"):
        ops.sfc_encode(torch.zeros(4, 3, dtype=torch.int64), torch.zeros(4, dtype=torch.int64), 3, ["z"])
    with pytest.raises(RuntimeError, match="GPU tensor"):
        ops.gemm(torch.zeros(4, 4), torch.zeros(4, 4))
    from ptv3_hip import autograd as A
    with pytest.raises(RuntimeError, match="GPU tensor"):
        A.linear(torch.zeros(4, 8, requires_grad=True), torch.zeros(8, 8), None)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        ops.gemm_tn(torch.zeros(4, 8), torch.zeros(4, 8))


def test_registry_semantics():
    from pointcept.utils.registry import Registry
    R = Registry("things")

    @R.register_module()
    class A:
        def __init__(self, x=1):
            self.x = x

    assert R.get("A") is A and "A" in R and len(R) == 1
    assert R.build(dict(type="A", x=3)).x == 3
    with pytest.raises(KeyError):
        R.register_module(module=A)
    R.register_module(name="A", force=True, module=A)
    R.register_module("alias", module=A)
    assert R.get("alias") is A
    with pytest.raises(KeyError, match="not in the things registry"):
        R.build(dict(type="nope"))
    with pytest.raises(TypeError, match="^A: "):
        R.build(dict(type="A", y=1))
    with pytest.raises(KeyError):
        R.build(dict(x=1))


def test_models_registered_and_state_dict_matches_reference(golden_dir):
    """Keys / shapes / order equal the reference class tree (captured from the reference in
    tests/golden/state_dict_fork_cfg.txt by instantiating its classes in the build container)."""
    from pointcept.models import MODELS, build_model
    from make_golden_cfg import FORK_CFG
    for name in ("PT-v3m1", "OffsetKeypointPTv3", "DefaultSegmentorV2"):
        assert MODELS.get(name) is not None
    cfg = dict(type="OffsetKeypointPTv3", num_keypoints=6, backbone_conf=dict(type="PT-v3m1", **FORK_CFG))
    keep = dict(cfg)
    model = build_model(cfg)
    assert cfg == keep  # build_model deep-copies (builder.py:15-17)
    mine = [f"{k} {tuple(v.shape)} {v.dtype}" for k, v in model.state_dict().items()]
    ref = open(os.path.join(golden_dir, "state_dict_fork_cfg.txt")).read().strip().split("\n")
    assert mine == ref
    assert sum(p.numel() for p in model.backbone.parameters()) == 46158272
    # training and eval both run on the HIP path only: CPU tensors are refused, never silently computed
    import ptv3_scenes as S
    data = S.make_batch([300], in_channels=4, extent=32, seed=0, with_target=6)
    for mode in (True, False):
        with pytest.raises(RuntimeError, match="GPU tensor|No HIP GPUs"):
            model.train(mode)(data)
    seg = build_model(dict(type="DefaultSegmentorV2", num_classes=19, backbone_out_channels=64,
                           backbone=dict(type="PT-v3m1", **FORK_CFG)))
    assert tuple(seg.seg_head.weight.shape) == (19, 64)


def test_point_dict_surface():
    from pointcept.models.utils.structure import Point
    p = Point(offset=torch.tensor([3, 5]), feat=torch.zeros(5, 2))
    assert p.batch.tolist() == [0, 0, 0, 1, 1] and "batch" in p.keys()
    q = Point(batch=torch.tensor([0, 0, 1]), feat=torch.zeros(3, 2))
    assert q.offset.tolist() == [2, 3]
    q.extra = 5
    assert q["extra"] == 5 and q.pop("extra") == 5
    with pytest.raises(AttributeError):
        q.missing


def test_sync_batchnorm_conversion_is_taken_back():
    """sync_bn=True (engines/train.py:256-257): torch replaces BatchNorm1d by SyncBatchNorm; the package adopts the
    converted modules back around the same tensors (optimizer / DDP / state_dict references stay valid)."""
    import torch
    from pointcept.models.utils.hip_layers import BatchNorm1d, adopt_sync_batchnorm
    seq = torch.nn.Sequential(torch.nn.Linear(4, 8), BatchNorm1d(8, eps=1e-3, momentum=0.01))
    keys = list(seq.state_dict().keys())
    seq = torch.nn.SyncBatchNorm.convert_sync_batchnorm(seq)
    assert isinstance(seq[1], torch.nn.SyncBatchNorm)
    params, bufs = list(seq.parameters()), list(seq.buffers())
    assert adopt_sync_batchnorm(seq) == 1 and adopt_sync_batchnorm(seq) == 0
    assert isinstance(seq[1], BatchNorm1d) and seq[1].sync_group is True
    assert seq[1].eps == 1e-3 and seq[1].momentum == 0.01
    assert all(a is b for a, b in zip(params, seq.parameters())) and all(a is b for a, b in zip(bufs, seq.buffers()))
    assert list(seq.state_dict().keys()) == keys


_REF_ROOT = "/root/reference"
_GRAFT_PROBE = r"""
import importlib.util, sys, os
ref, pkg = sys.argv[1], sys.argv[2]
sys.path.insert(0, pkg)
import pointcept.utils                      # this package's overlay (namespace for the two reference files)
for name in ("misc", "registry"):           # the REFERENCE's registry replaces the overlay's restatement
    spec = importlib.util.spec_from_file_location(f"pointcept.utils.{name}",
                                                  os.path.join(ref, "pointcept", "utils", f"{name}.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    setattr(pointcept.utils, name, mod)
from pointcept.models import build_model, MODELS
assert type(MODELS).__module__ == "pointcept.utils.registry" and MODELS.__class__.__name__ == "Registry"
assert sys.modules["pointcept.utils.registry"].__file__.startswith(ref)
scope = {}
exec(compile(open(os.path.join(ref, "configs", "my_dataset", "offset_keypoint_ptv3.py")).read(), "cfg", "exec"), scope)
model = build_model(scope["model"])
print("\n".join(f"{k} {tuple(v.shape)} {v.dtype}" for k, v in model.state_dict().items()))
"""


@pytest.mark.skipif(not os.path.isdir(_REF_ROOT), reason="build container only: needs the reference checkout")
def test_models_register_with_the_reference_registry_and_build_from_its_config():
    """INTEGRATION.md's graft, rehearsed: the reference's own Registry class (pointcept/utils/registry.py)
    takes this package's models, and the reference's training config (configs/my_dataset/
    offset_keypoint_ptv3.py:11-46) builds the offset model with exactly the reference's state_dict keys and
    shapes (tests/golden/state_dict_fork_cfg.txt, listed from the reference's own class)."""
    import subprocess
    import sys
    pkg = os.path.join(ROOT, "pointcept-keypointdetection_amd")
    r = subprocess.run([sys.executable, "-c", _GRAFT_PROBE, _REF_ROOT, pkg], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = [l.strip() for l in r.stdout.strip().splitlines() if l.strip()]
    want = [l.strip() for l in open(os.path.join(ROOT, "tests", "golden", "state_dict_fork_cfg.txt")) if l.strip()]
    assert got == want


@pytest.mark.parametrize("cfg_name,listing", [("SWIN3D_S3DIS_CFG", "state_dict_swin3d_s3dis.txt"),
                                              ("OFFSET_SWIN3D_CFG", "state_dict_offset_swin3d.txt"),
                                              ("TINY_SWIN3D_GRID_RESSTEM_CFG", "state_dict_swin3d_tiny_grid_resstem.txt")])
def test_swin3d_state_dict_matches_the_reference_classes(cfg_name, listing):
    """Keys, shapes, dtypes and order of "Swin3D-v1m1" / "OffsetKeypointSwin3D" against listings taken from the
    reference's own constructors (tests/golden/make_golden_swin3d.py; its header names the two stem modules whose keys
    come from a MinkowskiEngine stand-in).  The configs restate configs/s3dis/semseg-swin3d-v1m1-0-small.py:11-30 and
    configs/my_dataset/offset_keypoint_swin3d.py:11-40."""
    from ptv3_hip import configs
    from pointcept.models import build_model
    model = build_model(getattr(configs, cfg_name))
    got = [f"{k} {tuple(v.shape)} {v.dtype}" for k, v in model.state_dict().items()]
    want = [l.strip() for l in open(os.path.join(ROOT, "tests", "golden", listing)) if l.strip()]
    assert got == want


def test_swin3d_constructor_variants_build():
    from ptv3_hip import configs
    from pointcept.models import build_model
    build_model(dict(configs.TINY_SWIN3D_CFG, knn_down=False))            # GridDownsample: built since round 3
    build_model(dict(configs.TINY_SWIN3D_CFG, stem_transformer=False))    # MinkResBlock stem: built since round 3
    from pointcept.models.swin3d import WindowAttention
    # attn_drop: the reference builds nn.Dropout(attn_drop) (swin3d_layers.py:476) and never applies it - accepted, inert
    assert WindowAttention(32, 5, 4, 2, attn_drop=0.1).attn_drop.p == 0.1


def test_offset_models_report_the_reference_training_keys():
    """Every key the reference's offset wrappers put into their training result (InformationWriter logs all of them:
    offset_keypoint_ptv3.py:92-98, offset_keypoint_swin3d.py:92-124) is also produced by this package's classes -
    checked on the source text (the Swin3D class cannot be built here without MinkowskiEngine)."""
    import re
    ref = "/root/reference/pointcept/models"
    if not os.path.isdir(ref):
        pytest.skip("reference tree not present")
    pkg = os.path.join(ROOT, "pointcept-keypointdetection_amd", "pointcept", "models")
    for name in ("offset_keypoint_ptv3.py", "offset_keypoint_swin3d.py"):
        keys = lambda p: set(re.findall(r'\[f?"(train/[a-z0-9_{}]+)"\]', open(p).read()))  # noqa: E731
        want, got = keys(os.path.join(ref, name)), keys(os.path.join(pkg, name))
        assert want and want <= got, (name, want - got)
