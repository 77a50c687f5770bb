"""GPU parity of a whole training step: OffsetKeypointPTv3.train() forward + backward on the HIP
path against torch autograd over the oracle (CPU fp32 restatement, training=True): loss, every parameter
gradient, BatchNorm running statistics, and the fused AdamW update.  drop_path = 0 in these configs (the
DropPath Bernoulli stream cannot be shared between CPU and device); DropPath itself is tested on its own."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

from make_golden_cfg import TINY_CFG, FORK_CFG  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X box"
    return torch.device("cuda:0")


def _build(cfg, hidden_dim=256):
    from pointcept.models import build_model
    return build_model(dict(type="OffsetKeypointPTv3", num_keypoints=6, hidden_dim=hidden_dim,
                            backbone_conf=dict(type="PT-v3m1", **cfg)))


def _perturb_stats(model, seed=99):
    gen = torch.Generator().manual_seed(seed)
    for n, b in model.named_buffers():
        if n.endswith("running_mean"):
            b.copy_(torch.randn(b.shape, generator=gen) * 0.1)
        if n.endswith("running_var"):
            b.copy_(torch.rand(b.shape, generator=gen) + 0.5)


def _rel(got, ref, floor=1e-6):
    return (got.detach().float().cpu() - ref).abs().max().item() / max(ref.abs().max().item(), floor)


@pytest.mark.parametrize("cfg_name,sizes,hidden", [("tiny", [900, 700], 32), ("fork", [5000, 3000], 256)])
def test_train_step_vs_oracle_autograd(dev, cfg_name, sizes, hidden):
    from oracle import ptv3 as O
    import ptv3_scenes as S
    cfg = dict(TINY_CFG if cfg_name == "tiny" else FORK_CFG, drop_path=0.0)
    torch.manual_seed(1234)
    model = _build(cfg, hidden_dim=hidden)
    _perturb_stats(model)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    data = S.make_batch(sizes, in_channels=4, extent=96, seed=21, with_target=6)
    # ---- reference: torch autograd over the oracle
    orc = O.OffsetKeypointOracle(cfg, sd, training=True)
    torch.manual_seed(5)
    ref = orc.forward(data)
    ref["loss"].backward()
    ref_grads = {k: v.grad for k, v in orc.named_parameters()}
    ref_bufs = dict(orc.named_buffers())
    # ---- HIP path
    model = model.to(dev).train()
    torch.manual_seed(5)
    out = model({k: v.to(dev) for k, v in data.items()})
    assert set(out) >= {"loss", "train/cls_loss", "train/reg_loss", "train/offset_l1_err"}
    assert all(torch.is_tensor(v) and v.dim() == 0 for v in out.values())   # 0-d tensors
    out["loss"].backward()
    assert abs(out["loss"].item() - ref["loss"].item()) < 1e-4
    missing = [n for n, p in model.named_parameters() if p.grad is None]
    assert not missing, f"parameters without gradient (DDP find_unused_parameters=False needs all): {missing[:5]}"
    # per-parameter max error relative to that gradient's scale; gradients that are analytically zero (a bias
    # feeding a batch-statistic BatchNorm) are measured against 1e-3 of the largest gradient instead
    gmax = max(g.abs().max().item() for g in ref_grads.values())
    worst = max(((n, _rel(p.grad, ref_grads[n], 1e-3 * gmax)) for n, p in model.named_parameters()),
                key=lambda t: t[1])
    assert worst[1] < 2e-3, worst
    for n, b in model.named_buffers():
        if n.endswith(("running_mean", "running_var")):
            assert _rel(b, ref_bufs[n]) < 1e-4, n
    # ---- optimizer step: fused AdamW on the HIP grads == torch AdamW on the same grads
    from ptv3_hip.optim import FusedAdamW
    twin = copy.deepcopy(model)
    for p, q in zip(model.parameters(), twin.parameters()):
        q.grad = p.grad.clone()
    groups = lambda m: [  # noqa: E731  (the fork config's "block" keyword group, offset_keypoint_ptv3.py config)
        dict(params=[p for n, p in m.named_parameters() if "block" in n], lr=2e-4),
        dict(params=[p for n, p in m.named_parameters() if "block" not in n])]
    FusedAdamW(groups(model), lr=2e-3, weight_decay=0.05).step()
    torch.optim.AdamW(groups(twin), lr=2e-3, weight_decay=0.05).step()
    for (n, p), q in zip(model.named_parameters(), twin.parameters()):
        assert (p - q).abs().max().item() < 1e-6, n


def test_train_step_bf16_close_to_fp32(dev):
    """bf16 activations (fp32 master weights, fp32 weight-gradient accumulation) against the fp32 HIP run."""
    import ptv3_scenes as S
    cfg = dict(TINY_CFG, drop_path=0.0)
    torch.manual_seed(1234)
    m32 = _build(cfg, hidden_dim=32).to(dev).train()
    m16 = copy.deepcopy(m32)
    m16.backbone.compute_dtype = torch.bfloat16
    data = {k: v.to(dev) for k, v in S.make_batch([1500, 1100], in_channels=4, extent=96, seed=3, with_target=6).items()}
    losses = []
    for m in (m32, m16):
        torch.manual_seed(5)
        out = m(data)
        out["loss"].backward()
        losses.append(out["loss"].item())
    assert abs(losses[0] - losses[1]) < 0.05 * abs(losses[0])
    num = sum(((p.grad - q.grad) ** 2).sum() for p, q in zip(m32.parameters(), m16.parameters())).sqrt().item()
    den = sum((p.grad ** 2).sum() for p in m32.parameters()).sqrt().item()
    assert num / den < 0.25, num / den      # bf16 activations: ~3 significant digits through 15 blocks
    dot = sum((p.grad * q.grad).sum() for p, q in zip(m32.parameters(), m16.parameters())).item()
    n16 = sum((q.grad ** 2).sum() for q in m16.parameters()).sqrt().item()
    assert dot / (den * n16) > 0.98, dot / (den * n16)
    assert all(q.grad.dtype == torch.float32 for q in m16.parameters())


def test_drop_path_statistics(dev):
    from pointcept.models.utils.hip_layers import DropPath
    dp = DropPath(0.3).train()
    x = torch.ones(200000, 8, device=dev, requires_grad=True)
    torch.manual_seed(0)
    y = dp(x)
    rows = y[:, 0]
    kept = (rows > 0).float().mean().item()
    assert abs(kept - 0.7) < 0.01
    assert torch.all((rows == 0) | ((rows - 1 / 0.7).abs() < 1e-6))           # scale_by_keep
    assert torch.equal((y > 0).all(1), (y > 0).any(1))                         # whole rows dropped (per point)
    y.sum().backward()
    assert torch.equal(x.grad, y.detach())
    assert torch.equal(dp.eval()(x), x)


def test_loss_decreases_over_steps(dev):
    """A few optimizer steps on one batch reduce the loss (end-to-end sanity of the training glue)."""
    import ptv3_scenes as S
    from ptv3_hip.optim import FusedAdamW
    torch.manual_seed(7)
    model = _build(dict(TINY_CFG, drop_path=0.1), hidden_dim=32).to(dev).train()
    data = {k: v.to(dev) for k, v in S.make_batch([1200, 800], in_channels=4, extent=96, seed=9, with_target=6).items()}
    opt = FusedAdamW(model.parameters(), lr=3e-3, weight_decay=0.01)
    losses = []
    for _ in range(12):
        opt.zero_grad(set_to_none=True)
        out = model(data)
        out["loss"].backward()
        opt.step()
        losses.append(out["loss"].item())
    assert losses[-1] < 0.8 * losses[0], losses


@pytest.mark.parametrize("flash", [False, True])
def test_training_with_attention_dropout(dev, flash):
    """attn_drop > 0 (nn.Dropout on the probabilities at enable_flash=False, flash_attn's dropout_p at True; 0.0 in the
    reference's configs): the step runs through ptv3_window_attn_drop_fwd / _bwd (uniform windows, and the ragged ones
    of a scene shorter than the flash patch), is reproducible under torch.manual_seed, differs between seeds, leaves
    eval untouched, and the loss still falls.  The kernels themselves are checked against torch autograd with the same
    mask in test_hip_backward.py."""
    import ptv3_scenes as S
    from ptv3_hip.optim import FusedAdamW
    cfg = dict(TINY_CFG, drop_path=0.0, attn_drop=0.2, enable_flash=flash)
    torch.manual_seed(7)
    model = _build(cfg, hidden_dim=32).to(dev)
    sizes = [1200, 40] if flash else [1200, 800]       # 40 points < the patch: one short window (varlen form)
    data = {k: v.to(dev) for k, v in S.make_batch(sizes, in_channels=4, extent=96, seed=9, with_target=6).items()}
    model.eval()
    with torch.no_grad():           # shuffle_orders draws from torch's generator in eval too: same seed, same orders
        torch.manual_seed(1)
        e0 = model(dict(data))
        torch.manual_seed(1)
        e1 = model(dict(data))
    key = [k for k, v in e0.items() if torch.is_tensor(v) and v.dim() > 0][0] if isinstance(e0, dict) else None
    assert key is None or torch.equal(e0[key], e1[key])          # no dropout in eval
    model.train()

    def grads(seed):
        model.zero_grad(set_to_none=True)
        torch.manual_seed(seed)
        out = model(dict(data))
        out["loss"].backward()
        return out["loss"].item(), torch.cat([p.grad.flatten() for p in model.parameters() if p.grad is not None])
    l0, g0 = grads(3)
    l1, g1 = grads(3)
    l2, g2 = grads(4)
    assert l0 == l1 and torch.equal(g0, g1)
    assert l0 != l2 and not torch.equal(g0, g2)
    assert torch.isfinite(g0).all() and torch.isfinite(g2).all()
    opt = FusedAdamW(model.parameters(), lr=3e-3, weight_decay=0.01)
    losses = []
    torch.manual_seed(11)
    for _ in range(12):
        opt.zero_grad(set_to_none=True)
        out = model(dict(data))
        out["loss"].backward()
        opt.step()
        losses.append(out["loss"].item())
    assert losses[-1] < 0.85 * losses[0], losses


def test_v3m2_train_step_vs_oracle_autograd(dev):
    """"PT-v3m2" (GridPooling, LayerScale, LayerNorm everywhere) forward + backward vs torch autograd on the oracle."""
    from oracle import ptv3 as O
    from pointcept.models import build_model
    from make_golden_cfg import TINY_M2_CFG
    import ptv3_scenes as S
    cfg = dict(TINY_M2_CFG, drop_path=0.0)
    torch.manual_seed(77)
    model = build_model(dict(type="PT-v3m2", **cfg))
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() > 1:
                p.mul_(8.0)
            elif not n.endswith("gamma"):
                p.add_(torch.randn(p.shape, generator=g) * 0.1)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    data = S.make_batch([1100, 800], in_channels=4, extent=64, seed=8)
    w = torch.randn(data["coord"].shape[0], cfg["dec_channels"][0], generator=g)
    orc = O.PTv3m2Oracle(cfg, sd, training=True)
    torch.manual_seed(5)
    (orc.backbone(data)["feat"] * w).sum().backward()
    ref = {k: v.grad for k, v in orc.sd.items() if v.requires_grad}
    model = model.to(dev).train()
    torch.manual_seed(5)
    (model({k: v.to(dev) for k, v in data.items()}).feat * w.to(dev)).sum().backward()
    gmax = max(v.abs().max().item() for v in ref.values())
    worst = max(((n, _rel(p.grad, ref[n], 1e-3 * gmax)) for n, p in model.named_parameters()), key=lambda t: t[1])
    assert worst[1] < 2e-3, worst


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_shadow_weights_match_cast_path(dev, dtype):
    """FusedAdamW(shadow_dtype=...) keeps compute-dtype and transposed / tap-mirrored copies of every weight inside its
    step kernel; the training Functions read them instead of casting per step.  Same numbers as the cast path."""
    import ptv3_scenes as S
    from ptv3_hip.optim import FusedAdamW, weight_shadow
    cfg = dict(TINY_CFG, drop_path=0.0)
    torch.manual_seed(1234)
    ma = _build(cfg, hidden_dim=32).to(dev).train()
    mb = copy.deepcopy(ma)
    for m in (ma, mb):
        m.backbone.compute_dtype = dtype
    data = {k: v.to(dev) for k, v in S.make_batch([1300, 900], in_channels=4, extent=96, seed=3, with_target=6).items()}
    oa = FusedAdamW(ma.parameters(), lr=3e-3, weight_decay=0.01)
    ob = FusedAdamW(mb.parameters(), lr=3e-3, weight_decay=0.01, shadow_dtype=dtype)
    for step in range(3):
        losses = []
        for m, o in ((ma, oa), (mb, ob)):
            o.zero_grad()
            torch.manual_seed(5)
            out = m(data)
            out["loss"].backward()
            o.step()
            losses.append(out["loss"].item())
        assert losses[0] == losses[1], (step, losses)
    for (n, p), q in zip(ma.named_parameters(), mb.parameters()):
        assert torch.equal(p, q), n
    w = mb.backbone.enc.enc1.block0.cpe[0].weight          # a sparse-conv weight (out, 3, 3, 3, in)
    sh = weight_shadow(w, dtype)
    assert sh is not None
    assert torch.equal(sh["nat"], w.detach().reshape(w.shape[0], -1).to(dtype))
    ref_t = w.detach().reshape(w.shape[0], 27, w.shape[-1]).flip(1).permute(2, 1, 0).reshape(w.shape[-1], -1).to(dtype)
    assert torch.equal(sh["t"], ref_t)
    lin = mb.head[0].weight
    assert torch.equal(weight_shadow(lin, dtype)["t"], lin.detach().t().to(dtype))
    with torch.no_grad():
        lin.mul_(2.0)                                       # a torch-side change invalidates the shadow
    assert weight_shadow(lin, dtype) is None


def test_train_step_vs_reference_golden(dev, golden_dir):
    """The reference's own training-mode forward + torch-autograd backward (tests/golden/ptv3_tiny_train.npz, made by
    importing the reference in place) reproduced by the HIP training path: loss, all gradients, running statistics."""
    import numpy as np
    import os
    g = np.load(os.path.join(golden_dir, "ptv3_tiny_train.npz"))
    cfg = dict(TINY_CFG, drop_path=0.0)
    model = _build(cfg, hidden_dim=32)
    model.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd_")}, strict=True)
    model = model.to(dev).train()
    data = {k[3:]: torch.from_numpy(g[k]).to(dev) for k in g.files if k.startswith("in_")}
    torch.manual_seed(int(g["shuffle_seed"]))
    out = model(data)
    out["loss"].backward()
    assert abs(out["loss"].item() - float(g["loss"])) < 1e-4
    gmax = max(np.abs(g[k]).max() for k in g.files if k.startswith("grad_"))
    worst = max(((n, _rel(p.grad, torch.from_numpy(g["grad_" + n]), 1e-3 * gmax)) for n, p in model.named_parameters()),
                key=lambda t: t[1])
    assert worst[1] < 2e-3, worst
    for n, b in model.named_buffers():
        if "running" in n:
            assert _rel(b, torch.from_numpy(g["buf_" + n])) < 1e-4, n


def test_rpe_attention_backward_vs_autograd(dev):
    """ptv3_window_attn_rpe_bwd: dqkv and the gradient of the relative-position table against torch autograd over the
    restated vanilla attention with RPE (point_transformer_v3m1_base.py:29-48, 196-204)."""
    from ptv3_hip import autograd as A, ops
    from oracle import sfc, ptv3 as O
    import ptv3_scenes as S
    C, H, K = 32, 2, 64
    data = S.make_batch([300, 200], in_channels=4, extent=24, seed=9)
    off = data["offset"].numpy()
    batch = torch.arange(2).repeat_interleave(torch.tensor([300, 200])).numpy()
    _, order, inverse, _ = sfc.serialization(data["grid_coord"].numpy(), batch, ["z", "hilbert"])
    Kp = sfc.patch_size_for(off, K)
    pad, unpad, _ = sfc.pad_plan(off, Kp)
    g = torch.Generator().manual_seed(4)
    n = 500
    qkv = torch.randn(n, 3 * C, generator=g)
    pos_bnd = int((4 * K) ** (1 / 3) * 2)
    table = torch.randn(3 * (2 * pos_bnd + 1), H, generator=g) * 0.5
    dout = torch.randn(n, C, generator=g)
    # reference: autograd through the oracle's statements
    q, t = qkv.clone().requires_grad_(True), table.clone().requires_grad_(True)
    o = torch.from_numpy(order[1])[torch.from_numpy(pad)]
    inv = torch.from_numpy(unpad)[torch.from_numpy(inverse[1])]
    x = q[o]
    qq, kk, vv = x.reshape(-1, Kp, 3, H, C // H).permute(2, 0, 3, 1, 4).unbind(dim=0)
    attn = (qq * (C // H) ** -0.5) @ kk.transpose(-2, -1) + O.rpe_bias(t, data["grid_coord"][o], Kp, K, H)
    ref = (torch.softmax(attn, dim=-1) @ vv).transpose(1, 2).reshape(-1, C)[inv]
    ref.backward(dout)
    wo, wi = ops.window_maps(torch.from_numpy(order[1]).to(dev), torch.from_numpy(inverse[1]).to(dev),
                             torch.from_numpy(pad).to(dev), torch.from_numpy(unpad).to(dev))
    qd, td = qkv.to(dev).requires_grad_(True), table.to(dev).requires_grad_(True)
    out = A.window_attention_rpe(qd, td, wo, wi, data["grid_coord"].int().to(dev).contiguous(), H, Kp,
                                 (C // H) ** -0.5, pos_bnd)
    out.backward(dout.to(dev))
    assert (out.detach().cpu() - ref.detach()).abs().max().item() < 1e-4
    assert _rel(qd.grad, q.grad) < 1e-4
    assert _rel(td.grad, t.grad) < 1e-4


def test_train_step_with_rpe_vs_oracle_autograd(dev):
    """enable_rpe=True in training (configs/s3dis/semseg-pt-v3m1-1-rpe.py style): every parameter gradient, the
    rpe tables included, against torch autograd over the oracle."""
    from oracle import ptv3 as O
    import ptv3_scenes as S
    cfg = dict(TINY_CFG, drop_path=0.0, enable_rpe=True)
    torch.manual_seed(77)
    model = _build(cfg, hidden_dim=32)
    _perturb_stats(model)
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("rpe_table"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.5)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    data = S.make_batch([800, 500], in_channels=4, extent=48, seed=23, with_target=6)
    orc = O.OffsetKeypointOracle(cfg, sd, training=True)
    torch.manual_seed(5)
    ref = orc.forward(data)
    ref["loss"].backward()
    ref_grads = {k: v.grad for k, v in orc.named_parameters()}
    model = model.to(dev).train()
    torch.manual_seed(5)
    out = model({k: v.to(dev) for k, v in data.items()})
    out["loss"].backward()
    assert abs(out["loss"].item() - ref["loss"].item()) < 1e-4
    tables = [n for n, _ in model.named_parameters() if n.endswith("rpe_table")]
    assert len(tables) == 10 and all(ref_grads[n].abs().max() > 0 for n in tables)
    gmax = max(v.abs().max().item() for v in ref_grads.values())
    worst = max(((n, _rel(p.grad, ref_grads[n], 1e-3 * gmax)) for n, p in model.named_parameters()),
                key=lambda t: t[1])
    assert worst[1] < 2e-3, worst


def test_eval_between_fused_adamw_steps_sees_new_weights(dev):
    """train -> eval -> train -> eval with FusedAdamW (the evaluator-hook flow of tools/train.py): the step kernel
    writes the parameters through raw pointers, so it must bump their torch version - every eval-side cache (cast /
    permuted / folded weights, the native executor's packed table) is keyed on it.  The second evaluation has to equal
    a freshly built model loaded from the same state_dict, on the executor and on the module path."""
    import ptv3_scenes as S
    from ptv3_hip.optim import FusedAdamW
    cfg = dict(TINY_CFG, drop_path=0.0)
    torch.manual_seed(31)
    model = _build(cfg, hidden_dim=32).to(dev)
    opt = FusedAdamW(model.parameters(), lr=5e-3, weight_decay=0.01)
    data = {k: v.to(dev) for k, v in S.make_batch([1200, 800], in_channels=4, extent=64, seed=8, with_target=6).items()}

    def evaluate(m, use_engine):
        m.eval()
        m.backbone.use_engine = use_engine
        torch.manual_seed(3)
        with torch.no_grad():
            return m(data)["pred"].clone()

    def train_steps(n):
        model.train()
        for _ in range(n):
            opt.zero_grad()
            torch.manual_seed(4)
            model(data)["loss"].backward()
            opt.step()

    first = {u: evaluate(model, u) for u in (True, False)}          # fills every eval cache
    versions = [p._version for p in model.parameters()]
    train_steps(3)
    assert all(p._version > v for p, v in zip(model.parameters(), versions))
    for use in (True, False):
        got = evaluate(model, use)
        fresh = _build(cfg, hidden_dim=32)
        fresh.load_state_dict({k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, strict=True)
        want = evaluate(fresh.to(dev), use)
        assert torch.equal(got, want), use
        assert not torch.equal(got, first[use])
    train_steps(2)                                                   # and once more round the loop
    fresh = _build(cfg, hidden_dim=32)
    fresh.load_state_dict({k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, strict=True)
    assert torch.equal(evaluate(model, True), evaluate(fresh.to(dev), True))


def test_fused_adamw_state_dict_round_trip_with_torch_adamw(dev):
    """Checkpoints are interchangeable with torch.optim.AdamW (engines/hooks/misc.py:269 resumes with
    optimizer.load_state_dict): per-parameter state["step"] survives save / load in both directions and the bias
    corrections continue from it."""
    from ptv3_hip.optim import FusedAdamW
    g = torch.Generator().manual_seed(0)
    shapes = [(64, 32), (32,), (16, 27, 8), (8,)]

    def params():
        gg = torch.Generator().manual_seed(1)
        return [torch.nn.Parameter(torch.randn(s, generator=gg).to(dev)) for s in shapes]

    grads = [[torch.randn(s, generator=g).to(dev) for s in shapes] for _ in range(6)]

    def run(opt, ps, steps):
        for k in steps:
            for p, gr in zip(ps, grads[k]):
                p.grad = gr.clone()
            opt.step()

    kw = dict(lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.1)
    # reference: torch AdamW for 6 steps
    pt = params(); ot = torch.optim.AdamW(pt, **kw); run(ot, pt, range(6))
    # fused for 3 steps -> state_dict -> torch AdamW continues
    pf = params(); of = FusedAdamW(pf, **kw); run(of, pf, range(3))
    sd = of.state_dict()
    assert all(float(st["step"]) == 3.0 for st in sd["state"].values())
    p2 = [torch.nn.Parameter(p.detach().clone()) for p in pf]; o2 = torch.optim.AdamW(p2, **kw)
    o2.load_state_dict(sd); run(o2, p2, range(3, 6))
    for a, b in zip(p2, pt):
        assert (a - b).abs().max().item() < 2e-6
    # torch AdamW for 3 steps -> state_dict -> fused continues (resume path)
    p3 = params(); o3 = torch.optim.AdamW(p3, **kw); run(o3, p3, range(3))
    p4 = [torch.nn.Parameter(p.detach().clone()) for p in p3]; o4 = FusedAdamW(p4, **kw)
    o4.load_state_dict(o3.state_dict()); run(o4, p4, range(3, 6))
    for a, b in zip(p4, pt):
        assert (a - b).abs().max().item() < 2e-6
    assert all(float(st["step"]) == 6.0 for st in o4.state_dict()["state"].values())
    # a parameter that joins late keeps its own count (per-entry lag inside the one launch)
    p5 = params(); o5 = FusedAdamW(p5, **kw)
    p6 = params(); o6 = torch.optim.AdamW(p6, **kw)
    for k in range(4):
        for i, (a, b) in enumerate(zip(p5, p6)):
            a.grad = b.grad = None
            if k >= 2 or i != 1:                 # parameter 1 gets no gradient in the first two steps
                a.grad, b.grad = grads[k][i].clone(), grads[k][i].clone()
        o5.step(); o6.step()
    for a, b in zip(p5, p6):
        assert (a - b).abs().max().item() < 2e-6
    st5 = o5.state_dict()["state"]   # keyed by parameter index
    assert [float(st5[i]["step"]) for i in range(4)] == [4.0, 2.0, 4.0, 4.0]


@pytest.mark.parametrize("dtype,drop_path,flash", [(torch.float32, 0.0, False), (torch.bfloat16, 0.0, True),
                                                   (torch.float32, 0.3, False)])
def test_native_block_calls_equal_the_per_op_composition(dev, dtype, drop_path, flash):
    """ptv3_block_train_fwd / _bwd (one native call per block and direction) against BlockFn, the same block composed
    from the per-op entry points: identical kernels in identical order, so loss and every gradient are bitwise equal
    without DropPath; with DropPath both paths form the factor from the SAME uniform draws (u < keep ? 1 / keep : 0),
    the native one inside a fused multiply-add, the composition as a tensor fed to torch.addcmul: fp32 results agree to
    rounding.  Covers uniform and ragged (enable_flash) windows and the first decoder block, whose conv reads the
    skip tensor."""
    import ptv3_scenes as S
    from ptv3_hip import autograd as A
    cfg = dict(TINY_CFG, drop_path=drop_path, enable_flash=flash)
    data = S.make_batch([1300, 40, 900], in_channels=4, extent=96, seed=4, with_target=6)
    results = []
    for native in (True, False):
        torch.manual_seed(77)
        model = _build(cfg, hidden_dim=32)
        _perturb_stats(model)
        model = model.to(dev).train()
        model.backbone.compute_dtype = dtype
        A._NATIVE_BLOCK = native
        try:
            torch.manual_seed(5)
            out = model({k: v.to(dev) for k, v in data.items()})
            out["loss"].backward()
        finally:
            A._NATIVE_BLOCK = True
        results.append((out["loss"].detach().clone(), {n: p.grad.clone() for n, p in model.named_parameters()}))
    (l1, g1), (l0, g0) = results
    if drop_path == 0.0:
        assert torch.equal(l1, l0)
        for n in g0:
            assert torch.equal(g1[n], g0[n]), n
    else:
        assert abs(l1.item() - l0.item()) < 1e-5 * max(1.0, abs(l0.item()))
        gmax = max(g.abs().max().item() for g in g0.values())
        for n in g0:
            assert (g1[n] - g0[n]).abs().max().item() <= 1e-3 * max(g0[n].abs().max().item(), 1e-2 * gmax), n
