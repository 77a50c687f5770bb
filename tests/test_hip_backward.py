"""GPU parity of the training path: every autograd.Function of ptv3_hip.autograd against torch
autograd of the oracle's fp32 statement of the same layer (CPU).  The reference has no hand-written backward:
its gradients ARE torch autograd over the forward statements, so the oracle differentiated by torch is the
reference gradient.  Tolerance: 1e-4 relative to the gradient's scale (fp32); bf16 checked loosely."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from make_golden_cfg import ORDERS  # noqa: E402

F = torch.nn.functional


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X box"
    return torch.device("cuda:0")


def _close(got, ref, tol=1e-4, what=""):
    scale = max(1.0, ref.abs().max().item())
    err = (got.detach().float().cpu() - ref).abs().max().item()
    assert err <= tol * scale, f"{what}: err {err:.3e} vs scale {scale:.3e}"


def _leaf(t, dev=None, dtype=None):
    t = t.clone()
    if dev is not None:
        t = t.to(dev)
    if dtype is not None:
        t = t.to(dtype)
    return t.requires_grad_(True)


@pytest.mark.parametrize("m,cin,cout", [(1, 8, 8), (1000, 32, 96), (5003, 64, 24), (700, 36, 64), (4100, 512, 2048)])
def test_linear_backward(dev, m, cin, cout):
    from ptv3_hip import autograd as A
    g = torch.Generator().manual_seed(m)
    x, w, b = torch.randn(m, cin, generator=g), torch.randn(cout, cin, generator=g) / cin ** 0.5, torch.randn(cout, generator=g)
    dy = torch.randn(m, cout, generator=g)
    xr, wr, br = _leaf(x), _leaf(w), _leaf(b)
    F.linear(xr, wr, br).backward(dy)
    xd, wd, bd = _leaf(x, dev), _leaf(w, dev), _leaf(b, dev)
    y = A.linear(xd, wd, bd)
    y.backward(dy.to(dev))
    _close(xd.grad, xr.grad, what="dx")
    _close(wd.grad, wr.grad, what="dw")
    _close(bd.grad, br.grad, what="db")
    # bf16 activations, fp32 master weights
    xb = _leaf(x, dev, torch.bfloat16)
    wd2, bd2 = _leaf(w, dev), _leaf(b, dev)
    A.linear(xb, wd2, bd2).backward(dy.to(dev).bfloat16())
    assert xb.grad.dtype == torch.bfloat16 and wd2.grad.dtype == torch.float32
    _close(wd2.grad, wr.grad, tol=3e-2, what="dw bf16")
    _close(xb.grad, xr.grad, tol=3e-2, what="dx bf16")


@pytest.mark.parametrize("n,cin,cout,k,seed", [(2500, 32, 32, 3, 0), (1800, 4, 32, 5, 1), (900, 64, 48, 3, 2)])
def test_subm_conv_backward(dev, n, cin, cout, k, seed):
    from ptv3_hip import autograd as A, ops
    from oracle import ptv3 as O
    import ptv3_scenes as S
    data = S.make_batch([n - n // 3, n // 3], in_channels=cin, extent=40, seed=seed)
    gc, off = data["grid_coord"], data["offset"]
    n = gc.shape[0]
    batch = torch.repeat_interleave(torch.arange(2), torch.diff(off, prepend=torch.zeros(1, dtype=torch.long)))
    indices = torch.cat([batch[:, None], gc], 1).int()
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, cin, generator=g)
    w = torch.randn(cout, k, k, k, cin, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g)
    dy = torch.randn(n, cout, generator=g)
    xr, wr, br = _leaf(x), _leaf(w), _leaf(b)
    O.subm_conv3d(xr, indices, wr, br).backward(dy)
    nbr, _ = ops.subm_neighbors(indices.to(dev), k)
    xd, wd, bd = _leaf(x, dev), _leaf(w, dev), _leaf(b, dev)
    y = A.subm_conv(xd, wd, bd, nbr, None)
    y.backward(dy.to(dev))
    _close(xd.grad, xr.grad, what="dx")
    _close(wd.grad, wr.grad, what="dw")
    _close(bd.grad, br.grad, what="db")


@pytest.mark.parametrize("m,c", [(1, 32), (1000, 32), (333, 64), (2049, 128), (517, 256), (300, 512), (77, 48)])
def test_layernorm_backward(dev, m, c):
    from ptv3_hip import autograd as A
    g = torch.Generator().manual_seed(c + m)
    x = torch.randn(m, c, generator=g) * 2 + 0.5
    gm, bt, dy = torch.randn(c, generator=g), torch.randn(c, generator=g), torch.randn(m, c, generator=g)
    xr, gr, br = _leaf(x), _leaf(gm), _leaf(bt)
    F.layer_norm(xr, (c,), gr, br, 1e-5).backward(dy)
    xd, gd, bd = _leaf(x, dev), _leaf(gm, dev), _leaf(bt, dev)
    A.layer_norm(xd, gd, bd, 1e-5).backward(dy.to(dev))
    _close(xd.grad, xr.grad, what="dx")
    _close(gd.grad, gr.grad, what="dgamma")
    _close(bd.grad, br.grad, what="dbeta")


@pytest.mark.parametrize("m,c,dtype", [(100003, 32, torch.float32), (4097, 64, torch.bfloat16), (1001, 256, torch.float32),
                                       (515, 48, torch.float32), (130, 512, torch.bfloat16)])
def test_layernorm_backward_with_residual_gradient(dev, m, c, dtype):
    """ptv3_layernorm_bwd(add=...): dx = add + d/dx LayerNorm, the residual connection's gradient folded into the
    store (Block backward); packed-row kernel (c a power of two in 32..256) and the one-wave-per-row kernel."""
    from ptv3_hip import ops
    g = torch.Generator().manual_seed(c + m)
    x = (torch.randn(m, c, generator=g) * 2 + 0.5).to(dtype)
    gm, dy, add = torch.randn(c, generator=g), torch.randn(m, c, generator=g).to(dtype), torch.randn(m, c, generator=g).to(dtype)
    xr = _leaf(x.float())
    gr = _leaf(gm)
    br = _leaf(torch.zeros(c))
    F.layer_norm(xr, (c,), gr, br, 1e-5).backward(dy.float())
    dx, dg, db = ops.layernorm_bwd(x.to(dev), dy.to(dev), gm.to(dev), 1e-5, add=add.to(dev))
    plain, dg0, db0 = ops.layernorm_bwd(x.to(dev), dy.to(dev), gm.to(dev), 1e-5)
    want = add.float() + xr.grad
    tol = 1e-4 if dtype == torch.float32 else 2.0 ** -7
    assert (dx.float().cpu() - want).abs().max() <= tol * max(1.0, want.abs().max())
    assert (plain.float().cpu() - xr.grad).abs().max() <= tol * max(1.0, xr.grad.abs().max())
    for got, ref in ((dg, gr.grad), (db, br.grad)):
        assert (got.cpu() - ref).abs().max() <= 1e-4 * max(1.0, ref.abs().max()) * (1 if dtype == torch.float32 else 1)
    assert torch.equal(dg, dg0) and torch.equal(db, db0)


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("act", ["none", "gelu", "relu"])
def test_batchnorm_act_backward(dev, training, act):
    from ptv3_hip import autograd as A, ops
    m, c = 3001, 64
    g = torch.Generator().manual_seed(7)
    x = torch.randn(m, c, generator=g) * 1.7 + 3.0
    dy = torch.randn(m, c, generator=g)
    bn = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01)
    with torch.no_grad():
        bn.weight.copy_(torch.randn(c, generator=g)); bn.bias.copy_(torch.randn(c, generator=g))
        bn.running_mean.copy_(torch.randn(c, generator=g)); bn.running_var.copy_(torch.rand(c, generator=g) + 0.5)
    bn.train(training)
    import copy
    bnd = copy.deepcopy(bn).to(dev)
    fn = {"none": (lambda t: t), "gelu": F.gelu, "relu": F.relu}[act]
    aid = {"none": ops.ACT_NONE, "gelu": ops.ACT_GELU, "relu": ops.ACT_RELU}[act]
    xr = _leaf(x)
    yr = fn(bn(xr))
    yr.backward(dy)
    xd = _leaf(x, dev)
    yd = A.batch_norm_act(xd, bnd, aid)
    yd.backward(dy.to(dev))
    _close(yd, yr.detach(), what="y")
    _close(xd.grad, xr.grad, what="dx")
    _close(bnd.weight.grad, bn.weight.grad, what="dgamma")
    _close(bnd.bias.grad, bn.bias.grad, what="dbeta")
    _close(bnd.running_mean, bn.running_mean, what="running_mean")
    _close(bnd.running_var, bn.running_var, what="running_var")


def _attn_setup(sizes, K, seed):
    from oracle import sfc
    import ptv3_scenes as S
    data = S.make_batch(sizes, in_channels=4, extent=64, seed=seed)
    gc, off = data["grid_coord"].numpy(), data["offset"].numpy()
    batch = np.repeat(np.arange(len(off)), np.diff(off, prepend=0))
    _, order, inverse, _ = sfc.serialization(gc, batch, ORDERS)
    pad, unpad, _ = sfc.pad_plan(off, K)
    return gc.shape[0], order[1], inverse[1], pad, unpad


@pytest.mark.parametrize("sizes,C,H,K", [([300, 200], 32, 2, 64), ([1500], 64, 4, 256), ([130, 77, 300], 64, 2, 50),
                                         ([2100, 1030], 32, 2, 1024), ([400], 128, 2, 128)])
def test_window_attention_backward(dev, sizes, C, H, K):
    from ptv3_hip import autograd as A, ops
    from oracle import ptv3 as O
    n, order, inverse, pad, unpad = _attn_setup(sizes, K, seed=C + K)
    g = torch.Generator().manual_seed(K)
    qkv = torch.randn(n, 3 * C, generator=g)
    dout = torch.randn(n, C, generator=g)
    qr = _leaf(qkv)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    ref = O.window_attention_core(qr, t(order), t(inverse), t(pad), t(unpad), H, K)
    ref.backward(dout)
    wo, wi = ops.window_maps(t(order).to(dev), t(inverse).to(dev), t(pad).to(dev), t(unpad).to(dev))
    qd = _leaf(qkv, dev)
    out = A.window_attention(qd, wo, wi, H, K, (C // H) ** -0.5)
    out.backward(dout.to(dev))
    _close(out, ref.detach(), what="out")
    _close(qd.grad, qr.grad, what="dqkv")
    qb = _leaf(qkv, dev, torch.bfloat16)
    A.window_attention(qb, wo, wi, H, K, (C // H) ** -0.5).backward(dout.to(dev).bfloat16())
    _close(qb.grad, qr.grad, tol=5e-2, what="dqkv bf16")


@pytest.mark.parametrize("sizes,C,H,K", [([300, 200], 32, 2, 64), ([2100], 64, 4, 1024), ([130, 77, 300], 64, 2, 50)])
def test_window_attention_train_forward_leaves_the_log_sum_exp(dev, sizes, C, H, K):
    """ptv3_window_attn_train_fwd: the same output as the eval entry point plus, per (padded slot, head), the log2-domain
    log-sum-exp of the scaled scores - checked against torch.logsumexp on the gathered q, k; the backward that takes it
    (ptv3_window_attn_train_bwd) against the one that recomputes it."""
    from ptv3_hip import ops
    n, order, inverse, pad, unpad = _attn_setup(sizes, K, seed=C + K)
    g = torch.Generator().manual_seed(K + 1)
    qkv = torch.randn(n, 3 * C, generator=g)
    dout = torch.randn(n, C, generator=g)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    wo, wi = ops.window_maps(t(order).to(dev), t(inverse).to(dev), t(pad).to(dev), t(unpad).to(dev))
    scale = (C // H) ** -0.5
    x = qkv[t(order)[t(pad)]]
    q, k, _ = x.reshape(-1, K, 3, H, C // H).permute(2, 0, 3, 1, 4).unbind(dim=0)          # (W, H, K, D)
    want = torch.logsumexp((q * scale) @ k.transpose(-2, -1), dim=-1) / np.log(2.0)            # (W, H, K)
    want = want.permute(0, 2, 1).reshape(-1, H)                                                 # (n_pad, H)
    for dtype, tol in ((torch.float32, 1e-5), (torch.bfloat16, 2e-2)):
        qd = qkv.to(dev, dtype)
        out, lse = ops.window_attention_train(qd, wo, wi, H, K, scale)
        assert torch.equal(out, ops.window_attention(qd, wo, wi, H, K, scale))
        ref = want if dtype == torch.float32 else \
            (torch.logsumexp((q.bfloat16().float() * scale) @ k.bfloat16().float().transpose(-2, -1), dim=-1)
             / np.log(2.0)).permute(0, 2, 1).reshape(-1, H)
        assert (lse.cpu() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())
        d1 = ops.window_attention_train_bwd(qd, out, dout.to(dev, dtype), lse, wo, wi, H, K, scale)
        d0 = ops.window_attention_bwd(qd, out, dout.to(dev, dtype), wo, wi, H, K, scale)
        gs = d0.float().abs().max().item()
        assert (d1.float() - d0.float()).abs().max().item() <= (1e-5 if dtype == torch.float32 else 2e-2) * gs


@pytest.mark.parametrize("sizes,C,H,K,p", [([300, 200], 32, 2, 64, 0.1), ([1500], 64, 4, 256, 0.25), ([400], 128, 2, 128, 0.5),
                                           ([2100], 32, 2, 1024, 0.1)])
def test_window_attention_dropout_forward_and_backward(dev, sizes, C, H, K, p):
    """attn_drop > 0 in training (reference: nn.Dropout on the probabilities, v3m1_base.py:203 / flash dropout_p :211).
    The kernels draw the keep mask from a hash of (padded query slot, head, key slot, seed); the same mask rebuilt on
    the host (ops.drop_keep_mask) makes torch autograd over softmax * mask / (1 - p) the exact reference for the
    forward output and for dqkv.  The reference's own RNG stream is not reproduced (equal in distribution only)."""
    from ptv3_hip import autograd as A, ops
    n, order, inverse, pad, unpad = _attn_setup(sizes, K, seed=C + K)
    g = torch.Generator().manual_seed(K)
    qkv = torch.randn(n, 3 * C, generator=g)
    dout = torch.randn(n, C, generator=g)
    seed = 12345 + K
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    n_pad = len(pad)
    W = n_pad // K
    slot = (np.arange(W)[:, None, None, None] * K + np.arange(K)[None, None, :, None])
    keep = ops.drop_keep_mask(slot, np.arange(H)[None, :, None, None], np.arange(K)[None, None, None, :], H, seed, p)
    assert keep.shape == (W, H, K, K)
    assert abs(keep.mean() - (1.0 - p)) < 0.01                      # the hash drops a fraction p of the pairs
    mask = torch.from_numpy(keep.astype(np.float32)) / (1.0 - float(np.float32(p)))
    qr = _leaf(qkv)
    o, inv = t(order)[t(pad)], t(unpad)[t(inverse)]
    q, k, v = qr[o].reshape(-1, K, 3, H, C // H).permute(2, 0, 3, 1, 4).unbind(dim=0)
    attn = torch.softmax((q * (C // H) ** -0.5) @ k.transpose(-2, -1), dim=-1) * mask
    ref = (attn @ v).transpose(1, 2).reshape(-1, C)[inv]
    ref.backward(dout)
    wo, wi = ops.window_maps(t(order).to(dev), t(inverse).to(dev), t(pad).to(dev), t(unpad).to(dev))
    qd = _leaf(qkv, dev)
    out = A.window_attention_drop(qd, wo, wi, H, K, (C // H) ** -0.5, p, seed)
    out.backward(dout.to(dev))
    _close(out, ref.detach(), what="out")
    _close(qd.grad, qr.grad, what="dqkv")
    # another seed is another mask; the same seed the same result
    out2 = ops.window_attention_drop(qd.detach(), wo, wi, H, K, (C // H) ** -0.5, p, seed + 1)
    out3 = ops.window_attention_drop(qd.detach(), wo, wi, H, K, (C // H) ** -0.5, p, seed)
    assert not torch.equal(out2, out.detach()) and torch.equal(out3, out.detach())
    qb = _leaf(qkv, dev, torch.bfloat16)
    A.window_attention_drop(qb, wo, wi, H, K, (C // H) ** -0.5, p, seed).backward(dout.to(dev).bfloat16())
    _close(qb.grad, qr.grad, tol=5e-2, what="dqkv bf16")


def test_segment_max_and_cluster_gather_backward(dev):
    from ptv3_hip import autograd as A
    g = torch.Generator().manual_seed(3)
    n, c = 5000, 48
    seg_len = torch.randint(1, 9, (1200,), generator=g)
    seg_len = seg_len[: int((seg_len.cumsum(0) <= n).sum())]
    n = int(seg_len.sum())
    n_out = seg_len.numel()
    seg_start = torch.cat([torch.zeros(1, dtype=torch.long), seg_len.cumsum(0)]).int()
    order0 = torch.randperm(n, generator=g)
    feat = torch.randn(n, c, generator=g)
    dy = torch.randn(n_out, c, generator=g)
    fr = _leaf(feat)
    ref = torch.segment_reduce(fr[order0], "max", lengths=seg_len, axis=0)
    ref.backward(dy)
    fd = _leaf(feat, dev)
    out = A.segment_max(fd, order0.to(dev), seg_start.to(dev), n_out)
    out.backward(dy.to(dev))
    _close(out, ref.detach(), tol=0.0, what="max")
    _close(fd.grad, fr.grad, tol=0.0, what="dfeat")
    # gather by cluster id / segment-sum backward
    cluster = torch.empty(n, dtype=torch.long)
    cluster[order0] = torch.repeat_interleave(torch.arange(n_out), seg_len)
    pf = torch.randn(n_out, c, generator=g)
    dyg = torch.randn(n, c, generator=g)
    pr = _leaf(pf)
    pr[cluster].backward(dyg)
    pd = _leaf(pf, dev)
    A.cluster_gather(pd, cluster.to(dev), order0.to(dev), seg_start.to(dev)).backward(dyg.to(dev))
    _close(pd.grad, pr.grad, what="dparent")


def test_adamw_matches_torch(dev):
    from ptv3_hip.optim import FusedAdamW
    g = torch.Generator().manual_seed(0)
    shapes = [(64, 32), (32,), (3, 5, 7), (100000,), (1,)]
    ref_p = [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in shapes]
    dev_p = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in ref_p]
    groups = lambda ps: [dict(params=ps[:2], lr=2e-3, weight_decay=0.05), dict(params=ps[2:], lr=5e-4)]  # noqa: E731
    ref = torch.optim.AdamW(groups(ref_p), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    opt = FusedAdamW(groups(dev_p), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    for step in range(4):
        for pr, pd in zip(ref_p, dev_p):
            gr = torch.randn(pr.shape, generator=g)
            pr.grad = gr.clone()
            pd.grad = gr.to(dev)
        if step == 2:  # scheduler-style lr change between steps
            for o in (ref, opt):
                o.param_groups[0]["lr"] = 1e-3
        ref.step()
        opt.step()
    for pr, pd in zip(ref_p, dev_p):
        _close(pd, pr.detach(), tol=2e-6, what="param")
    total = torch.sqrt(sum((p.grad.float() ** 2).sum() for p in ref_p))
    assert abs(opt.grad_norm().item() - total.item()) <= 1e-4 * total.item()
