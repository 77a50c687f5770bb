"""GPU parity for the enable_flash=True form of the path (fixed patch, ragged windows: the
flash_attn_varlen_qkvpacked_func call site, point_transformer_v3m1_base.py:114-170, 207-215) and for
DefaultSegmentorV2 (models/default.py:41-95) - BASELINE configs[2] (PTv3 semseg on a ~120k-point LiDAR scan).

Goldens: tests/golden/flash_seg.npz (reference run, tests/golden/make_golden_flash_seg.py).  The reference casts
q/k/v to bf16 in front of flash-attn and receives a bf16 result (:209-214) whatever the model dtype; the HIP fp32
mode keeps fp32 there, so it is compared (1e-4) with the oracle's exact form - itself pinned to the reference run
through its bf16_io switch (tests/test_oracle_golden.py) - and with the reference run at a bf16-rounding bound."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from make_golden_cfg import ORDERS, TINY_CFG, SEMSEG_CFG  # noqa: E402

FP32_TOL = 1e-4
BF16_EPS = 2.0 ** -8


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X box"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "flash_seg.npz"))


def test_flash_pad_plan_golden(dev, g):
    """Fixed-patch pad plan incl. scenes shorter than the patch: pad / unpad / cu_seqlens bit-exact; the one-launch
    window plan of the executor writes the same cu_seqlens."""
    from ptv3_hip import ops
    for i in range(int(g["fp_cases"])):
        off, K = g[f"fp{i}_offset"], int(g[f"fp{i}_K"])
        offd = torch.from_numpy(off).to(dev)
        pad, unpad, cu = ops.pad_plan(offd, off.tolist(), K)
        assert np.array_equal(pad.cpu().numpy(), g[f"fp{i}_pad"]), i
        assert np.array_equal(unpad.cpu().numpy(), g[f"fp{i}_unpad"]), i
        assert np.array_equal(cu.cpu().numpy(), g[f"fp{i}_cu"]), i
        n = int(off[-1])
        ident = torch.arange(n, device=dev).view(1, n)
        wo, wi, cu2 = ops.window_plan(ident, ident, offd, off.tolist(), K, with_cu=True)
        assert np.array_equal(cu2.cpu().numpy(), g[f"fp{i}_cu"]), i
        assert np.array_equal(wo[0].cpu().numpy(), g[f"fp{i}_pad"]), i
        assert np.array_equal(wi[0].cpu().numpy(), g[f"fp{i}_unpad"]), i
        n_pad, nwin, ragged, _ = ops.plan_sizes(off.tolist(), K)
        assert n_pad == len(g[f"fp{i}_pad"]) and nwin == len(g[f"fp{i}_cu"]) - 1
        assert ragged == bool((np.diff(g[f"fp{i}_cu"]) < K).any())


def _flash_case(g, i):
    from oracle import sfc
    t = f"fa{i}_"
    C, H, K, oi = [int(v) for v in g[t + "cfg"]]
    off, gc = g[t + "offset"], g[t + "grid_coord"]
    batch = np.repeat(np.arange(len(off)), np.diff(off, prepend=0))
    _, order, inverse, _ = sfc.serialization(gc, batch, ORDERS)
    pad, unpad, cu = sfc.pad_plan(off, K)
    w = {k[len(t) + 2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(t + "w_")}
    return dict(C=C, H=H, K=K, off=off, order=order[oi], inverse=inverse[oi], pad=pad, unpad=unpad, cu=cu, w=w,
                qkv=g[t + "qkv"], out=g[t + "out"])


@pytest.mark.parametrize("case", [0, 1, 2, 3])
def test_flash_attention_vs_oracle_and_reference(dev, g, case):
    from ptv3_hip import ops
    from oracle import ptv3 as O
    c = _flash_case(g, case)
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    wo, wi = ops.window_maps(to(c["order"]), to(c["inverse"]), to(c["pad"]), to(c["unpad"]))
    cu = to(c["cu"])
    scale = (c["C"] // c["H"]) ** -0.5
    out = ops.window_attention_varlen(to(c["qkv"]), wo, wi, cu, c["H"], c["K"], scale)
    # exact form of the published varlen function on the same qkv
    o = torch.from_numpy(c["order"])[torch.from_numpy(c["pad"])]
    inv = torch.from_numpy(c["unpad"])[torch.from_numpy(c["inverse"])]
    qkv = torch.from_numpy(c["qkv"])
    core = O.varlen_attention(qkv[o].reshape(-1, 3, c["H"], c["C"] // c["H"]), c["cu"], c["H"], scale)
    core = core.reshape(-1, c["C"])[inv]
    assert (out.cpu() - core).abs().max().item() < FP32_TOL
    # the reference run (bf16 q/k/v and bf16 result around the library call): within bf16 rounding
    proj = ops.gemm(out, to(c["w"]["proj.weight"].numpy()), bias=to(c["w"]["proj.bias"].numpy())).cpu()
    ref = torch.from_numpy(c["out"])
    bound = 8 * BF16_EPS * max(1.0, ref.abs().max().item())
    assert (proj - ref).abs().max().item() < bound
    # bf16 mode of the kernel (the arithmetic the reference's flash path uses)
    out16 = ops.window_attention_varlen(to(c["qkv"]).bfloat16(), wo, wi, cu, c["H"], c["K"], scale).float().cpu()
    assert (out16 - core).abs().max().item() < 8 * BF16_EPS * max(1.0, core.abs().max().item())
    # uniform-window entry point refuses ragged plans instead of reading past a window
    if len(c["pad"]) % c["K"]:
        with pytest.raises(RuntimeError, match="not a multiple"):
            ops.window_attention(to(c["qkv"]), wo, wi, c["H"], c["K"], scale)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_flash_attention_backward_vs_autograd(dev, dtype):
    """ptv3_window_attn_varlen_bwd against torch autograd over the restated varlen function."""
    from ptv3_hip import ops
    from oracle import sfc, ptv3 as O
    gen = torch.Generator().manual_seed(3)
    C, H, K = 64, 4, 128
    off = np.array([300, 390, 518, 519, 800])     # scenes of 300, 90, 128, 1, 281 points
    n = int(off[-1])
    pad, unpad, cu = sfc.pad_plan(off, K)
    order = torch.cat([torch.randperm(b - a, generator=gen) + a for a, b in zip([0] + off[:-1].tolist(), off.tolist())])
    inverse = torch.empty_like(order)
    inverse[order] = torch.arange(n)
    qkv = torch.randn(n, 3 * C, generator=gen)
    dout = torch.randn(n, C, generator=gen)
    if dtype == torch.bfloat16:
        qkv, dout = qkv.bfloat16().float(), dout.bfloat16().float()
    scale = (C // H) ** -0.5
    q = qkv.clone().requires_grad_(True)
    o = order[torch.from_numpy(pad)]
    inv = torch.from_numpy(unpad)[inverse]
    ref = O.varlen_attention(q[o].reshape(-1, 3, H, C // H), cu, H, scale).reshape(-1, C)[inv]
    ref.backward(dout)
    wo, wi = ops.window_maps(order.to(dev), inverse.to(dev), torch.from_numpy(pad).to(dev),
                             torch.from_numpy(unpad).to(dev))
    cud = torch.from_numpy(cu).to(dev)
    qd = qkv.to(dev, dtype)
    out = ops.window_attention_varlen(qd, wo, wi, cud, H, K, scale)
    dq = ops.window_attention_bwd(qd, out, dout.to(dev, dtype), wo, wi, H, K, scale, cu_seqlens=cud).float().cpu()
    tol = 1e-4 if dtype == torch.float32 else 4 * BF16_EPS
    gscale = q.grad.abs().max().item()
    assert (out.float().cpu() - ref.detach()).abs().max().item() < tol * max(1.0, ref.abs().max().item())
    assert (dq - q.grad).abs().max().item() < tol * gscale, ((dq - q.grad).abs().max().item(), gscale)


def test_flash_attention_dropout_ragged_windows_vs_autograd_with_the_same_mask(dev):
    """dropout_p of the flash call (v3m1_base.py:211) on ragged windows: ptv3_window_attn_drop_fwd / _bwd with
    cu_seqlens against torch autograd over softmax * mask / (1 - p) per sequence, the mask rebuilt on the host from the
    kernels' hash (query slot = padded slot, key = position inside its sequence)."""
    from ptv3_hip import ops
    from oracle import sfc
    gen = torch.Generator().manual_seed(5)
    C, H, K, p, seed = 64, 4, 128, 0.3, 777
    off = np.array([300, 390, 518, 519, 800])     # scenes of 300, 90, 128, 1, 281 points
    n = int(off[-1])
    pad, unpad, cu = sfc.pad_plan(off, K)
    order = torch.cat([torch.randperm(b - a, generator=gen) + a for a, b in zip([0] + off[:-1].tolist(), off.tolist())])
    inverse = torch.empty_like(order)
    inverse[order] = torch.arange(n)
    qkv = torch.randn(n, 3 * C, generator=gen)
    dout = torch.randn(n, C, generator=gen)
    scale = (C // H) ** -0.5
    q = qkv.clone().requires_grad_(True)
    o = order[torch.from_numpy(pad)]
    inv = torch.from_numpy(unpad)[inverse]
    x = q[o].reshape(-1, 3, H, C // H)
    outs = torch.empty(x.shape[0], H, C // H)
    kept = total = 0
    for a, b in zip(cu[:-1].tolist(), cu[1:].tolist()):
        ln = b - a
        blk = x[a:b].permute(1, 2, 0, 3)                               # (3, H, ln, D)
        attn = torch.softmax((blk[0] * scale) @ blk[1].transpose(-2, -1), dim=-1)
        keep = ops.drop_keep_mask(np.arange(a, b)[None, :, None], np.arange(H)[:, None, None],
                                  np.arange(ln)[None, None, :], H, seed, p)
        kept += keep.sum(); total += keep.size
        attn = attn * torch.from_numpy(keep.astype(np.float32)) / (1.0 - float(np.float32(p)))
        outs[a:b] = (attn @ blk[2]).transpose(0, 1)
    assert abs(kept / total - (1 - p)) < 0.01
    ref = outs.reshape(-1, C)[inv]
    ref.backward(dout)
    wo, wi = ops.window_maps(order.to(dev), inverse.to(dev), torch.from_numpy(pad).to(dev),
                             torch.from_numpy(unpad).to(dev))
    cud = torch.from_numpy(cu).to(dev)
    qd = qkv.to(dev)
    out = ops.window_attention_drop(qd, wo, wi, H, K, scale, p, seed, cu_seqlens=cud)
    dq = ops.window_attention_drop_bwd(qd, out, dout.to(dev), wo, wi, H, K, scale, p, seed, cu_seqlens=cud).cpu()
    assert (out.cpu() - ref.detach()).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())
    assert (dq - q.grad).abs().max().item() < 1e-4 * q.grad.abs().max().item()


def _offset_model(cfg, hidden_dim=32):
    from pointcept.models import build_model
    return build_model(dict(type="OffsetKeypointPTv3", num_keypoints=6, hidden_dim=hidden_dim,
                            backbone_conf=dict(type="PT-v3m1", **cfg)))


def test_flash_model_vs_oracle_and_reference(dev, g):
    """OffsetKeypointPTv3 over PT-v3m1 with enable_flash=True on a batch holding scenes shorter than / equal to the
    patch: executor == module path (bitwise), == oracle (1e-4), reference run within bf16 rounding."""
    from oracle import ptv3 as O
    cfg = dict(TINY_CFG, enable_flash=True)
    sd = {k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("fm_sd_")}
    data = {k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("fm_in_")}
    seed = int(g["fm_shuffle_seed"])
    orc = O.OffsetKeypointOracle(cfg, sd)
    torch.manual_seed(seed)
    with torch.no_grad():
        ref = orc.forward(data)
    model = _offset_model(cfg)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    datad = {k: v.to(dev) for k, v in data.items()}
    outs = {}
    for use in (True, False):
        model.backbone.use_engine = use
        torch.manual_seed(seed)
        with torch.no_grad():
            outs[use] = model(datad)
    assert torch.equal(outs[True]["pred"], outs[False]["pred"])
    pred = outs[True]["pred"].cpu()
    assert (pred - ref["pred"]).abs().max().item() < FP32_TOL
    assert abs(outs[True]["loss"].item() - ref["loss"].item()) < FP32_TOL
    gold = torch.from_numpy(g["fm_pred"])
    assert (pred - gold).abs().max().item() < 16 * BF16_EPS * max(1.0, gold.abs().max().item())
    # bf16 compute: the reference's own arithmetic for the attention, everything else bf16 too
    model.backbone.use_engine = True
    model.backbone.compute_dtype = torch.bfloat16
    torch.manual_seed(seed)
    with torch.no_grad():
        p16 = model(datad)["pred"].cpu()
    assert (p16 - ref["pred"]).abs().max().item() < 32 * BF16_EPS * max(1.0, ref["pred"].abs().max().item())


def test_flash_train_step_vs_oracle_autograd(dev, g):
    """Training with enable_flash=True (ragged windows in the taped block Function): loss and every gradient
    against torch autograd over the oracle."""
    from oracle import ptv3 as O
    cfg = dict(TINY_CFG, enable_flash=True, drop_path=0.0)
    sd = {k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("fm_sd_")}
    data = {k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("fm_in_")}
    orc = O.OffsetKeypointOracle(cfg, sd, training=True)
    torch.manual_seed(17)
    ref = orc.forward(data)
    ref["loss"].backward()
    model = _offset_model(cfg)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).train()
    torch.manual_seed(17)
    out = model({k: v.to(dev) for k, v in data.items()})
    out["loss"].backward()
    assert abs(out["loss"].item() - ref["loss"].item()) < FP32_TOL
    grads = dict(orc.named_parameters())
    gmax = max(p.grad.abs().max().item() for p in grads.values())
    for name, p in model.named_parameters():
        r = grads[name].grad
        err = (p.grad.cpu() - r).abs().max().item()
        assert err <= 2e-3 * max(r.abs().max().item(), 1e-2 * gmax), (name, err)


# ------------------------------------------------------------------------------------------------
# DefaultSegmentorV2
# ------------------------------------------------------------------------------------------------
def _segmentor(cfg, num_classes, criteria=None):
    from pointcept.models import build_model
    return build_model(dict(type="DefaultSegmentorV2", num_classes=num_classes,
                            backbone_out_channels=cfg["dec_channels"][0], backbone=dict(type="PT-v3m1", **cfg),
                            criteria=criteria))


def test_segmentor_reference_golden(dev, g, golden_dir):
    """seg_logits and CrossEntropy loss of the reference's DefaultSegmentorV2 run; the three return modes."""
    crit = [dict(type="CrossEntropyLoss", loss_weight=1.0, ignore_index=-1)]
    model = _segmentor(TINY_CFG, 13, crit)
    mine = [f"{k} {tuple(v.shape)}" for k, v in model.state_dict().items()]
    want = open(os.path.join(golden_dir, "state_dict_segmentor_tiny.txt")).read().strip().split("\n")
    assert mine == want
    sd = {k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sg_sd_")}
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    data = {k[6:]: torch.from_numpy(g[k]).to(dev) for k in g.files if k.startswith("sg_in_")}
    seed = int(g["sg_shuffle_seed"])
    for use in (True, False):
        model.backbone.use_engine = use
        torch.manual_seed(seed)
        with torch.no_grad():
            out = model(data)
        assert sorted(out.keys()) == ["loss", "seg_logits"]
        assert np.abs(out["seg_logits"].cpu().numpy() - g["sg_seg_logits"]).max() < FP32_TOL, use
        assert abs(out["loss"].item() - float(g["sg_loss"])) < FP32_TOL
    torch.manual_seed(seed)
    with torch.no_grad():
        test_mode = model({k: v for k, v in data.items() if k != "segment"}, return_point=True)
    assert sorted(test_mode.keys()) == ["point", "seg_logits"]
    assert np.abs(test_mode["seg_logits"].cpu().numpy() - g["sg_seg_logits"]).max() < FP32_TOL
    model.train()
    torch.manual_seed(seed)
    tr = model(data)
    assert sorted(tr.keys()) == ["loss"] and tr["loss"].requires_grad
    tr["loss"].backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())


def test_semseg_lidar_120k_vs_oracle(dev):
    """BASELINE configs[2]: PTv3 semseg (upstream backbone: fork widths, enable_flash=True, 1024-point patches,
    Linear(64 -> 19) head) on one LiDAR-like scan of 120 000 voxels whose grid needs >= 11 bits per axis:
    serialization codes / orders bit-exact, seg_logits within 1e-4 (fp32) of the oracle; bf16 within the rounding
    bound below.  Deep levels hold fewer points than the patch -> short windows at every level below the first."""
    from oracle import ptv3 as O, sfc
    import ptv3_scenes as S
    torch.manual_seed(1234)
    model = _segmentor(SEMSEG_CFG, 19).eval()
    gen = torch.Generator().manual_seed(99)
    for n, b in model.named_buffers():
        if n.endswith("running_mean"):
            b.copy_(torch.randn(b.shape, generator=gen) * 0.1)
        if n.endswith("running_var"):
            b.copy_(torch.rand(b.shape, generator=gen) + 0.5)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    data = S.make_batch([120000], in_channels=4, extent=2048, seed=21, kind="lidar")
    depth = sfc.serialized_depth(data["grid_coord"].numpy())
    assert depth >= 11, depth
    orc = O.SegmentorOracle(SEMSEG_CFG, sd)
    torch.manual_seed(5)
    with torch.no_grad():
        ref = orc.forward(data)
    model = model.to(dev)
    datad = {k: v.to(dev) for k, v in data.items()}
    torch.manual_seed(5)
    with torch.no_grad():
        out = model(datad, return_point=True)
    pt = out["point"]
    assert int(pt.serialized_depth) == depth
    assert torch.equal(pt.serialized_code.cpu(), orc.backbone.trace["serialized_code"])
    assert torch.equal(pt.serialized_order.cpu(), orc.backbone.trace["serialized_order"])
    assert pt["_stage_points"] == [orc.backbone.trace[f"n{s}"] for s in range(5)]
    logits = out["seg_logits"].cpu()
    err = (logits - ref["seg_logits"]).abs().max().item()
    assert err < FP32_TOL, err
    assert torch.equal(logits.argmax(1), ref["seg_logits"].argmax(1)) or \
        (logits.argmax(1) != ref["seg_logits"].argmax(1)).float().mean().item() < 1e-4
    model.backbone.compute_dtype = torch.bfloat16
    torch.manual_seed(5)
    with torch.no_grad():
        l16 = model(datad)["seg_logits"].cpu()
    scale = ref["seg_logits"].abs().max().item()
    e16 = (l16 - ref["seg_logits"]).abs()
    print(f"semseg lidar 120k: fp32 max err {err:.2e}; bf16 max err {e16.max().item():.3e} mean {e16.mean().item():.3e} "
          f"(logit scale {scale:.3f})")
    assert e16.max().item() < 64 * BF16_EPS * max(1.0, scale)
