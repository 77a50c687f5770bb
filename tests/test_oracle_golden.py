"""The oracle (CPU restatement) against golden vectors produced by the reference's own code.

Golden files: tests/golden/{sfc,padplan,attention,ptv3_tiny}.npz, written by
tests/golden/make_golden.py in the build container (reference imported in place)."""
import os

import numpy as np
import pytest
import torch

from oracle import sfc, ptv3 as O
from make_golden_cfg import TINY_CFG, ORDERS


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_sfc_codes_survey_appendix_a1():
    # captured from default.py:9-24 in the survey session
    gc = np.array([[684, 559, 629], [192, 835, 763], [707, 359, 9], [723, 277, 754], [804, 599, 70],
                   [472, 600, 396], [314, 705, 486], [551, 87, 174]])
    want = {
        "z": [948411859, 448565787, 580453047, 716501166, 873341402, 364408384, 359581802, 539665406],
        "z-trans": [944217573, 746361389, 341443383, 476901662, 840253932, 595095104, 599106652, 271696894],
        "hilbert": [679408746, 319150882, 1038347237, 855888336, 554424821, 404320082, 416813811, 1013732344],
        "hilbert-trans": [674082936, 927765046, 522841811, 390699204, 601067873, 978865122, 973232755, 515733064],
    }
    for o, w in want.items():
        assert sfc.encode(gc, np.zeros(8, dtype=np.int64), 10, o).tolist() == w


def test_sfc_golden(golden_dir):
    g = _load(golden_dir, "sfc.npz")
    for depth in (3, 5, 7, 10, 16):
        for B in (1, 2, 8):
            t = f"d{depth}_b{B}"
            code, order, inverse, _ = sfc.serialization(g[t + "_grid_coord"], g[t + "_batch"], ORDERS, depth)
            assert np.array_equal(code, g[t + "_code"])
            assert np.array_equal(order, g[t + "_order"])
            assert np.array_equal(inverse, g[t + "_inverse"])
    assert sfc.serialized_depth(g["auto_grid_coord"]) == int(g["auto_depth"])
    code, *_ = sfc.serialization(g["auto_grid_coord"], g["auto_batch"], ORDERS)
    assert np.array_equal(code, g["auto_code"])


def test_padplan_golden(golden_dir):
    g = _load(golden_dir, "padplan.npz")
    for i in range(int(g["n_cases"])):
        off = g[f"c{i}_offset"]
        K = sfc.patch_size_for(off, int(g[f"c{i}_pmax"]))
        assert K == int(g[f"c{i}_K"])
        pad, unpad, cu = sfc.pad_plan(off, K)
        assert np.array_equal(pad, g[f"c{i}_pad"])
        assert np.array_equal(unpad, g[f"c{i}_unpad"])
        assert np.array_equal(cu, g[f"c{i}_cu"])


def test_padplan_survey_appendix_a2():
    pad, unpad, cu = sfc.pad_plan([5, 12], 4)
    assert pad.tolist() == [0, 1, 2, 3, 4, 1, 2, 3, 5, 6, 7, 8, 9, 10, 11, 8]
    assert unpad.tolist() == [0, 1, 2, 3, 4, 8, 9, 10, 11, 12, 13, 14]
    assert cu.tolist() == [0, 4, 8, 12, 16]


def test_attention_golden(golden_dir):
    g = _load(golden_dir, "attention.npz")
    for i in range(int(g["n_cases"])):
        t = f"a{i}_"
        C, H, pmax, oi, rpe, K = [int(v) for v in g[t + "cfg"]]
        off = g[t + "offset"]
        gc = g[t + "grid_coord"]
        batch = np.repeat(np.arange(len(off)), np.diff(off, prepend=0))
        code, order, inverse, _ = sfc.serialization(gc, batch, ORDERS)
        assert sfc.patch_size_for(off, pmax) == K
        pad, unpad, _ = sfc.pad_plan(off, K)
        w = {k[len(t) + 2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(t + "w_")}
        out = O.window_attention(
            torch.from_numpy(g[t + "feat"]), w["qkv.weight"], w["qkv.bias"], w["proj.weight"], w["proj.bias"],
            torch.from_numpy(order[oi]), torch.from_numpy(inverse[oi]), torch.from_numpy(pad),
            torch.from_numpy(unpad), H, K,
            rpe_table=w.get("rpe.rpe_table") if rpe else None,
            grid_coord=torch.from_numpy(gc), patch_size_cfg=pmax)
        ref = torch.from_numpy(g[t + "out"])
        assert (out - ref).abs().max().item() < 2e-6, i
        # the fused core (what the HIP kernel computes) reproduces the pre-projection tensor
        core = O.window_attention_core(torch.from_numpy(g[t + "qkv"]), torch.from_numpy(order[oi]),
                                       torch.from_numpy(inverse[oi]), torch.from_numpy(pad),
                                       torch.from_numpy(unpad), H, K)
        if not rpe:
            proj = torch.nn.functional.linear(core, w["proj.weight"], w["proj.bias"])
            assert (proj - ref).abs().max().item() < 2e-6, i


def test_attention_k1024_golden(golden_dir):
    """the reference's SerializedAttention at (C, H, K) = (64, 4, 1024) and (512, 32, 1024)
    (tests/golden/make_golden_attention_k1024.py)."""
    g = _load(golden_dir, "attention_k1024.npz")
    for i in range(int(g["n_cases"])):
        t = f"a{i}_"
        C, H, pmax, oi, rpe, K = [int(v) for v in g[t + "cfg"]]
        assert K == 1024
        off, gc = g[t + "offset"], g[t + "grid_coord"]
        batch = np.repeat(np.arange(len(off)), np.diff(off, prepend=0))
        code, order, inverse, _ = sfc.serialization(gc, batch, ORDERS)
        pad, unpad, _ = sfc.pad_plan(off, K)
        w = {k[len(t) + 2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(t + "w_")}
        out = O.window_attention(
            torch.from_numpy(g[t + "feat"]), w["qkv.weight"], w["qkv.bias"], w["proj.weight"], w["proj.bias"],
            torch.from_numpy(order[oi]), torch.from_numpy(inverse[oi]), torch.from_numpy(pad),
            torch.from_numpy(unpad), H, K, grid_coord=torch.from_numpy(gc), patch_size_cfg=pmax)
        ref = torch.from_numpy(g[t + "out"])
        assert (out - ref).abs().max().item() < 5e-6 * max(1.0, ref.abs().max().item()), i


def test_evaluator_oracle_matches_reference_hook(golden_dir):
    """oracle/keypoints.py::evaluator_totals against what the REFERENCE's OffsetKeypointEvaluator hook reported on the
    same seeded batches (tests/golden/make_golden_evaluator.py ran engines/hooks/offset_keypoint_evaluator.py:19-123)."""
    from oracle import keypoints as KO
    g = _load(golden_dir, "evaluator.npz")
    for tag in ("r0", "r1", "r2"):
        tot = np.zeros(14)
        for i in range(int(g[tag + "_nbatch"])):
            b = {k: torch.from_numpy(g[f"{tag}_b{i}_{k}"]) for k in ("coord", "target", "offset", "pred")}
            scale = torch.from_numpy(g[f"{tag}_b{i}_scale"]) if f"{tag}_b{i}_scale" in g.files else None
            tot += np.array(KO.evaluator_totals(b["pred"], b["target"], b["coord"], b["offset"], scale, 6))
        assert abs(tot[0] / (tot[1] + 1e-6) - float(g[tag + "_mean_dist"])) < 1e-6
        assert np.array_equal(tot[8:].astype(int), g[tag + "_kp_counts"])
        assert np.allclose(tot[2:8] / (tot[8:] + 1e-6), g[tag + "_kp_mean_dist"], atol=1e-6)
        assert abs(float(g[tag + "_metric"]) + float(g[tag + "_mean_dist"])) < 1e-12


def test_full_model_golden(golden_dir):
    g = _load(golden_dir, "ptv3_tiny.npz")
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd_")}
    data = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("in_")}
    orc = O.OffsetKeypointOracle(TINY_CFG, sd)
    torch.manual_seed(int(g["shuffle_seed"]))
    with torch.no_grad():
        out = orc.forward(data)
    tr = orc.backbone.trace
    assert np.array_equal(tr["serialized_code"].numpy(), g["tap_serialized_code"])
    assert np.array_equal(tr["serialized_order"].numpy(), g["tap_serialized_order"])
    for k in ["embedding"] + [f"enc{s}" for s in range(5)] + [f"dec{s}" for s in range(4)]:
        assert tr[k].shape == g["tap_" + k].shape
        assert np.abs(tr[k].numpy() - g["tap_" + k]).max() < 2e-5, k
    assert np.abs(out["pred"].numpy() - g["pred"]).max() < 2e-5
    assert abs(out["loss"].item() - float(g["loss"])) < 1e-5


# ------------------------------------------------------------------------------------------------
# GridSample restatement vs the reference's own outputs (tests/golden/gridsample.npz)
# ------------------------------------------------------------------------------------------------
def _gs_cases(golden_dir):
    g = np.load(os.path.join(golden_dir, "gridsample.npz"))
    for ci in range(4):
        t = f"c{ci}_"
        yield ci, {k[len(t):]: g[k] for k in g.files if k.startswith(t)}


def test_gridsample_oracle_matches_reference(golden_dir):
    from oracle import gridsample as GS
    for ci, c in _gs_cases(golden_dir):
        coord, gs, hash_type = c["coord"], float(c["grid_size"]), str(c["hash"])
        p = GS.grid_sample_plan(coord, gs, hash_type)
        nvox = len(p["count"])
        # everything that does not depend on the order of equal keys is bit-exact
        assert np.array_equal(p["inverse"], c["train_inverse"]), ci
        assert np.array_equal(p["min_coord"], c["train_min_coord"]), ci
        assert nvox == len(c["train_grid_coord"]) == len(c["test_grid_coord"])
        assert int(c["test_nparts"]) == p["count"].max()
        assert np.array_equal(c["test_cover"], np.arange(len(coord)))
        # the reference's selected points: one member of voxel j at position j, grid_coord / displacement of it
        pid = c["train_point_id"]
        assert np.array_equal(p["inverse"][pid], np.arange(nvox)), ci
        assert np.array_equal(p["grid"][pid], c["train_grid_coord"]), ci
        assert np.array_equal((p["scaled"] - p["grid"] - 0.5)[pid], c["train_displacement"]), ci
        assert np.array_equal(coord[pid], c["train_coord"]), ci
        # voxels with one member: the pick itself is determined
        single = p["count"] == 1
        rand = np.zeros(nvox, dtype=np.int64)
        mine = GS.grid_sample_train(coord, gs, hash_type, rand)
        assert np.array_equal(mine["idx_unique"][single], pid[single]), ci
        assert np.array_equal(mine["grid_coord"], c["train_grid_coord"]), ci


def test_v3m2_oracle_matches_reference(golden_dir):
    """"PT-v3m2" restatement vs the reference's own eval run (tests/golden/ptv3m2_tiny.npz)."""
    from oracle import ptv3 as O
    from make_golden_cfg import TINY_M2_CFG
    g = np.load(os.path.join(golden_dir, "ptv3m2_tiny.npz"))
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd_")}
    data = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("in_")}
    orc = O.PTv3m2Oracle(TINY_M2_CFG, sd)
    torch.manual_seed(int(g["shuffle_seed"]))
    with torch.no_grad():
        P = orc.backbone(data)
    for name in ["embedding"] + [f"enc{s}" for s in range(5)] + [f"dec{s}" for s in (3, 2, 1, 0)]:
        ref = torch.from_numpy(g["tap_" + name])
        assert orc.trace[name].shape == ref.shape, name
        assert (orc.trace[name] - ref).abs().max().item() < 1e-5, name
    assert (P["feat"] - torch.from_numpy(g["feat"])).abs().max().item() < 1e-5


def test_training_oracle_matches_reference_autograd(golden_dir):
    """Training semantics of the oracle (batch-statistic BatchNorm, running-stat momenta, loss) pinned to the
    reference's own training-mode run + torch autograd (tests/golden/ptv3_tiny_train.npz)."""
    from oracle import ptv3 as O
    from make_golden_cfg import TINY_CFG
    g = np.load(os.path.join(golden_dir, "ptv3_tiny_train.npz"))
    cfg = dict(TINY_CFG, drop_path=0.0)
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd_")}
    data = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("in_")}
    orc = O.OffsetKeypointOracle(cfg, sd, training=True)
    torch.manual_seed(int(g["shuffle_seed"]))
    out = orc.forward(data)
    out["loss"].backward()
    assert abs(out["loss"].item() - float(g["loss"])) < 1e-6
    gmax = max(np.abs(g[k]).max() for k in g.files if k.startswith("grad_"))
    for name, p in orc.named_parameters():
        ref = torch.from_numpy(g["grad_" + name])
        err = (p.grad - ref).abs().max().item()
        assert err <= 1e-5 * max(ref.abs().max().item(), 1e-3 * gmax), (name, err)
    for name, b in orc.named_buffers():
        assert (b - torch.from_numpy(g["buf_" + name])).abs().max().item() < 1e-6, name


# ------------------------------------------------------------------------------------------------
# enable_flash=True (fixed patch, ragged windows) and DefaultSegmentorV2 vs the reference's own runs
# (tests/golden/flash_seg.npz, written by tests/golden/make_golden_flash_seg.py)
# ------------------------------------------------------------------------------------------------
def test_flash_padplan_golden(golden_dir):
    """get_padding_and_inverse with the patch FIXED (enable_flash=True): scenes shorter than K are one short window."""
    g = _load(golden_dir, "flash_seg.npz")
    ragged = 0
    for i in range(int(g["fp_cases"])):
        off, K = g[f"fp{i}_offset"], int(g[f"fp{i}_K"])
        pad, unpad, cu = sfc.pad_plan(off, K)
        assert np.array_equal(pad, g[f"fp{i}_pad"]), i
        assert np.array_equal(unpad, g[f"fp{i}_unpad"]), i
        assert np.array_equal(cu, g[f"fp{i}_cu"]), i
        ragged += int((np.diff(cu) < K).any())
    assert ragged >= 5   # the fixture really holds short windows


def test_flash_attention_golden(golden_dir):
    g = _load(golden_dir, "flash_seg.npz")
    for i in range(int(g["fa_cases"])):
        t = f"fa{i}_"
        C, H, K, oi = [int(v) for v in g[t + "cfg"]]
        off, gc = g[t + "offset"], g[t + "grid_coord"]
        batch = np.repeat(np.arange(len(off)), np.diff(off, prepend=0))
        _, order, inverse, _ = sfc.serialization(gc, batch, ORDERS)
        pad, unpad, cu = sfc.pad_plan(off, K)
        assert np.array_equal(cu, g[t + "cu"]), i
        w = {k[len(t) + 2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(t + "w_")}
        args = (torch.from_numpy(g[t + "feat"]), w["qkv.weight"], w["qkv.bias"], w["proj.weight"], w["proj.bias"],
                torch.from_numpy(order[oi]), torch.from_numpy(inverse[oi]), torch.from_numpy(pad),
                torch.from_numpy(unpad), torch.from_numpy(cu), H)
        ref = torch.from_numpy(g[t + "out"])
        # with the call site's own bf16 cast (:209) and the library's bf16 result: the reference run itself
        assert (O.window_attention_flash(*args, bf16_io=True) - ref).abs().max().item() < 2e-6, i
        # the exact-fp32 form (what the HIP fp32 mode computes) stays within bf16 rounding of it
        exact = O.window_attention_flash(*args, bf16_io=False)
        assert (exact - ref).abs().max().item() < 2e-2 * max(1.0, ref.abs().max().item()), i


def test_flash_model_golden(golden_dir):
    g = _load(golden_dir, "flash_seg.npz")
    sd = {k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("fm_sd_")}
    data = {k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("fm_in_")}
    cfg = dict(TINY_CFG, enable_flash=True)
    orc = O.OffsetKeypointOracle(cfg, sd)
    orc.backbone.flash_bf16_io = True
    torch.manual_seed(int(g["fm_shuffle_seed"]))
    with torch.no_grad():
        out = orc.forward(data)
    tr = orc.backbone.trace
    for k in [f"enc{s}" for s in range(5)] + [f"dec{s}" for s in range(4)]:
        assert tr[k].shape == g["fm_tap_" + k].shape, k
        assert np.abs(tr[k].numpy() - g["fm_tap_" + k]).max() < 5e-5, k
    assert np.abs(out["pred"].numpy() - g["fm_pred"]).max() < 5e-5
    assert abs(out["loss"].item() - float(g["fm_loss"])) < 1e-5


def test_segmentor_golden(golden_dir):
    """DefaultSegmentorV2 restatement vs the reference class + the reference's own CrossEntropyLoss."""
    g = _load(golden_dir, "flash_seg.npz")
    sd = {k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sg_sd_")}
    data = {k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sg_in_")}
    orc = O.SegmentorOracle(TINY_CFG, sd)
    torch.manual_seed(int(g["sg_shuffle_seed"]))
    with torch.no_grad():
        out = orc.forward(data)
    assert np.abs(out["seg_logits"].numpy() - g["sg_seg_logits"]).max() < 2e-5
    assert abs(out["loss"].item() - float(g["sg_loss"])) < 1e-5
