"""GPU parity: every HIP entry point (through the C ABI) against the oracle / the golden vectors.

Bit-exact for integer outputs (codes, order, inverse, pad plan, cluster ids, neighbour tables, knn idx);
fp32 feature outputs within 1e-4 of the oracle (tolerance of BASELINE.json north_star); bf16 mode is
checked against the fp32 oracle with a bf16-sized tolerance."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from make_golden_cfg import ORDERS, TINY_CFG, FORK_CFG  # noqa: E402

FP32_TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X box"
    return torch.device("cuda:0")


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


# ------------------------------------------------------------------------------------------------
# serialization
# ------------------------------------------------------------------------------------------------
def test_sfc_encode_golden(dev, golden_dir):
    from ptv3_hip import ops
    g = _g(golden_dir, "sfc.npz")
    for depth in (3, 5, 7, 10, 16):
        for B in (1, 2, 8):
            t = f"d{depth}_b{B}"
            for dt in (torch.int64, torch.int32):
                gc = torch.from_numpy(g[t + "_grid_coord"]).to(dev).to(dt)
                batch = torch.from_numpy(g[t + "_batch"]).to(dev)
                code = ops.sfc_encode(gc, batch, depth, ORDERS)
                assert np.array_equal(code.cpu().numpy(), g[t + "_code"]), (t, dt)
                end_bit = depth * 3 + max(B - 1, 0).bit_length()
                order, inverse = ops.argsort_codes(code, max(1, end_bit))
                assert np.array_equal(order.cpu().numpy(), g[t + "_order"]), t
                assert np.array_equal(inverse.cpu().numpy(), g[t + "_inverse"]), t


def test_sfc_encode_large_vs_oracle(dev):
    from ptv3_hip import ops
    from oracle import sfc
    import ptv3_scenes as S
    data = S.make_batch([60000, 41000], in_channels=4, extent=512, seed=5)
    gc, off = data["grid_coord"], data["offset"]
    batch = torch.repeat_interleave(torch.arange(2), torch.diff(off, prepend=torch.zeros(1, dtype=torch.long)))
    depth = sfc.serialized_depth(gc.numpy())
    code = ops.sfc_encode(gc.to(dev), batch.to(dev), depth, ORDERS)
    ref_code, ref_order, ref_inv, _ = sfc.serialization(gc.numpy(), batch.numpy(), ORDERS, depth)
    assert np.array_equal(code.cpu().numpy(), ref_code)
    order, inverse = ops.argsort_codes(code, depth * 3 + 1)
    assert np.array_equal(order.cpu().numpy(), ref_order)
    assert np.array_equal(inverse.cpu().numpy(), ref_inv)


@pytest.mark.parametrize("n,k,bits", [(1, 1, 8), (63, 2, 9), (2048, 1, 16), (2049, 3, 24), (100000, 4, 31),
                                        (1 << 20, 2, 49), (300007, 1, 63)])
def test_argsort_properties(dev, n, k, bits):
    """size-independent properties at and beyond BASELINE sizes: sortedness, stability, inverse."""
    from ptv3_hip import ops
    g = torch.Generator().manual_seed(n + bits)
    hi = torch.randint(0, 1 << min(bits, 31), (k, n), generator=g, dtype=torch.int64)
    lo = torch.randint(0, 1 << 31, (k, n), generator=g, dtype=torch.int64)
    code = ((hi << 31) | lo) & ((1 << bits) - 1)
    if n > 10:
        code[:, n // 2] = code[:, n // 3]  # force ties: stable order must keep the lower index first
    code = code.to(dev)
    order, inverse = ops.argsort_codes(code, bits)
    ref = torch.argsort(code.cpu(), dim=1, stable=True)
    assert torch.equal(order.cpu(), ref)
    ar = torch.arange(n).repeat(k, 1)
    assert torch.equal(torch.gather(inverse.cpu(), 1, ref), ar)


def test_pad_plan_golden(dev, golden_dir):
    from ptv3_hip import ops
    g = _g(golden_dir, "padplan.npz")
    for i in range(int(g["n_cases"])):
        off = g[f"c{i}_offset"]
        K = int(g[f"c{i}_K"])
        pad, unpad, cu = ops.pad_plan(torch.from_numpy(off).to(dev), off.tolist(), K)
        assert np.array_equal(pad.cpu().numpy(), g[f"c{i}_pad"]), i
        assert np.array_equal(unpad.cpu().numpy(), g[f"c{i}_unpad"]), i
        assert np.array_equal(cu.cpu().numpy(), g[f"c{i}_cu"]), i


# ------------------------------------------------------------------------------------------------
# window attention
# ------------------------------------------------------------------------------------------------
def _attn_case(g, i, dev):
    from oracle import sfc
    t = f"a{i}_"
    C, H, pmax, oi, rpe, K = [int(v) for v in g[t + "cfg"]]
    off, gc = g[t + "offset"], g[t + "grid_coord"]
    batch = np.repeat(np.arange(len(off)), np.diff(off, prepend=0))
    code, order, inverse, _ = sfc.serialization(gc, batch, ORDERS)
    pad, unpad, _ = sfc.pad_plan(off, K)
    w = {k[len(t) + 2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(t + "w_")}
    return dict(C=C, H=H, K=K, oi=oi, rpe=rpe, pmax=pmax, off=off, gc=gc, order=order[oi], inverse=inverse[oi],
                pad=pad, unpad=unpad, w=w, qkv=g[t + "qkv"], feat=g[t + "feat"], out=g[t + "out"])


@pytest.mark.parametrize("case", [0, 1, 2, 3, 4])
def test_window_attention_golden_fp32(dev, golden_dir, case):
    from ptv3_hip import ops
    from oracle import ptv3 as O
    c = _attn_case(_g(golden_dir, "attention.npz"), case, dev)
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    wo, wi = ops.window_maps(to(c["order"]), to(c["inverse"]), to(c["pad"]), to(c["unpad"]))
    assert np.array_equal(wo.cpu().numpy(), c["order"][c["pad"]])
    assert np.array_equal(wi.cpu().numpy(), c["unpad"][c["inverse"]])
    # the one-launch plan used by the native executor gives the same maps
    wo2, wi2 = ops.window_plan(to(c["order"])[None], to(c["inverse"])[None], to(c["off"]), c["off"].tolist(), c["K"])
    assert torch.equal(wo2[0], wo) and torch.equal(wi2[0], wi)
    out = ops.window_attention(to(c["qkv"]), wo, wi, c["H"], c["K"], (c["C"] // c["H"]) ** -0.5)
    core = O.window_attention_core(torch.from_numpy(c["qkv"]), torch.from_numpy(c["order"]),
                                   torch.from_numpy(c["inverse"]), torch.from_numpy(c["pad"]),
                                   torch.from_numpy(c["unpad"]), c["H"], c["K"])
    assert (out.cpu() - core).abs().max().item() < FP32_TOL
    # through the projection GEMM: the reference module's output
    proj = ops.gemm(out, to(c["w"]["proj.weight"].numpy()), bias=to(c["w"]["proj.bias"].numpy()))
    assert (proj.cpu() - torch.from_numpy(c["out"])).abs().max().item() < FP32_TOL


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_window_attention_launch_configs_bitwise(dev, dtype):
    """Every (waves per workgroup, query tiles per wave) launch configuration of the resident-window kernel gives the
    bits of the configuration the cost model picks: a query's key walk does not depend on how queries are dealt to
    waves.  Shapes: ragged scenes (borrowed rows), a partial last key tile, head_dim 16 and 32."""
    import os
    from ptv3_hip import ops
    g = torch.Generator().manual_seed(11)
    for sizes, c, h, patch in (([1500, 1100], 64, 4, 1024), ([900], 32, 2, 200), ([2600], 64, 2, 512)):
        n = sum(sizes)
        qkv = torch.randn(n, 3 * c, generator=g).to(dev).to(dtype)
        off = torch.tensor(sizes).cumsum(0)
        order = torch.cat([torch.randperm(m, generator=g) + (int(off[i]) - m) for i, m in enumerate(sizes)])
        inverse = torch.empty_like(order)
        inverse[order] = torch.arange(n)
        wo, wi = ops.window_plan(order[None].to(dev), inverse[None].to(dev), off.to(dev), off.tolist(), patch)
        wo, wi = wo[0].contiguous(), wi[0].contiguous()
        try:
            ref = ops.window_attention(qkv, wo, wi, h, patch, (c // h) ** -0.5)
            for waves in (8, 4):
                for qt in (4, 2, 1):
                    os.environ["PTV3_ATTN_WAVES"], os.environ["PTV3_ATTN_QT"] = str(waves), str(qt)
                    out = ops.window_attention(qkv, wo, wi, h, patch, (c // h) ** -0.5)
                    assert torch.equal(out, ref), (sizes, c, h, patch, waves, qt)
        finally:
            os.environ.pop("PTV3_ATTN_WAVES", None)
            os.environ.pop("PTV3_ATTN_QT", None)


@pytest.mark.parametrize("case", [0, 1, 2, 3, 4])
def test_window_attention_bf16(dev, golden_dir, case):
    """bf16 kernel against the REFERENCE module's own output (attention.npz `out`, through the projection) and against
    the fp32 restatement of the core: 8 bf16 steps of the largest value (q, k, v, P and the result are rounded to
    8 significant bits; the softmax statistics stay fp32)."""
    from ptv3_hip import ops
    from oracle import ptv3 as O
    c = _attn_case(_g(golden_dir, "attention.npz"), case, dev)
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    wo, wi = ops.window_maps(to(c["order"]), to(c["inverse"]), to(c["pad"]), to(c["unpad"]))
    out = ops.window_attention(to(c["qkv"]).bfloat16(), wo, wi, c["H"], c["K"], (c["C"] // c["H"]) ** -0.5)
    assert out.dtype == torch.bfloat16
    core = O.window_attention_core(torch.from_numpy(c["qkv"]), torch.from_numpy(c["order"]),
                                   torch.from_numpy(c["inverse"]), torch.from_numpy(c["pad"]),
                                   torch.from_numpy(c["unpad"]), c["H"], c["K"])
    assert (out.float().cpu() - core).abs().max().item() < 8 * 2.0 ** -8 * max(1.0, core.abs().max().item())
    proj = ops.gemm(out, to(c["w"]["proj.weight"].numpy()).bfloat16(), bias=to(c["w"]["proj.bias"].numpy()))
    gold = torch.from_numpy(c["out"])
    assert (proj.float().cpu() - gold).abs().max().item() < 8 * 2.0 ** -8 * max(1.0, gold.abs().max().item())


@pytest.mark.parametrize("case", [0, 1])
def test_window_attention_k1024_golden(dev, golden_dir, case):
    """(C, H, K) = (64, 4, 1024) and (512, 32, 1024): the resident-window kernel at the full patch against the reference
    module's own output, fp32 1e-4 through the projection and bf16 within 8 steps."""
    from ptv3_hip import ops
    from oracle import sfc
    g = _g(golden_dir, "attention_k1024.npz")
    t = f"a{case}_"
    C, H, pmax, oi, rpe, K = [int(v) for v in g[t + "cfg"]]
    off, gc = g[t + "offset"], g[t + "grid_coord"]
    batch = np.repeat(np.arange(len(off)), np.diff(off, prepend=0))
    code, order, inverse, _ = sfc.serialization(gc, batch, ORDERS)
    pad, unpad, _ = sfc.pad_plan(off, K)
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    wo, wi = ops.window_maps(to(order[oi]), to(inverse[oi]), to(pad), to(unpad))
    w = {k[len(t) + 2:]: to(g[k]) for k in g.files if k.startswith(t + "w_")}
    gold = torch.from_numpy(g[t + "out"])
    for dtype, tol in ((torch.float32, FP32_TOL), (torch.bfloat16, 8 * 2.0 ** -8)):
        qkv = ops.gemm(to(g[t + "feat"]).to(dtype), w["qkv.weight"].to(dtype), bias=w["qkv.bias"])
        out = ops.window_attention(qkv, wo, wi, H, K, (C // H) ** -0.5)
        proj = ops.gemm(out, w["proj.weight"].to(dtype), bias=w["proj.bias"])
        assert (proj.float().cpu() - gold).abs().max().item() < tol * max(1.0, gold.abs().max().item())


def test_window_attention_rpe(dev, golden_dir):
    from ptv3_hip import ops
    from oracle import ptv3 as O
    c = _attn_case(_g(golden_dir, "attention.npz"), 5, dev)
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    wo, wi = ops.window_maps(to(c["order"]), to(c["inverse"]), to(c["pad"]), to(c["unpad"]))
    o = torch.from_numpy(c["order"])[torch.from_numpy(c["pad"])]
    bias = O.rpe_bias(c["w"]["rpe.rpe_table"], torch.from_numpy(c["gc"])[o], c["K"], c["pmax"], c["H"]).contiguous()
    out = ops.window_attention(to(c["qkv"]), wo, wi, c["H"], c["K"], (c["C"] // c["H"]) ** -0.5,
                               rpe_bias=bias.to(dev))
    proj = ops.gemm(out, to(c["w"]["proj.weight"].numpy()), bias=to(c["w"]["proj.bias"].numpy()))
    assert (proj.cpu() - torch.from_numpy(c["out"])).abs().max().item() < FP32_TOL
    # the same bias looked up from the table inside the kernel (no (windows, H, K, K) tensor)
    pos_bnd = int((4 * c["pmax"]) ** (1 / 3) * 2)
    out2 = ops.window_attention_rpe(to(c["qkv"]), wo, wi, c["H"], c["K"], (c["C"] // c["H"]) ** -0.5,
                                    to(c["gc"]).int().contiguous(), c["w"]["rpe.rpe_table"].float().contiguous().to(dev),
                                    pos_bnd)
    assert out2 is not None
    assert (out2.cpu() - out.cpu()).abs().max().item() < 1e-5
    proj2 = ops.gemm(out2, to(c["w"]["proj.weight"].numpy()), bias=to(c["w"]["proj.bias"].numpy()))
    assert (proj2.cpu() - torch.from_numpy(c["out"])).abs().max().item() < FP32_TOL
    out16 = ops.window_attention_rpe(to(c["qkv"]).bfloat16(), wo, wi, c["H"], c["K"], (c["C"] // c["H"]) ** -0.5,
                                     to(c["gc"]).int().contiguous(),
                                     c["w"]["rpe.rpe_table"].float().contiguous().to(dev), pos_bnd)
    assert (out16.float().cpu() - out.cpu()).abs().max().item() < 3e-2


# ------------------------------------------------------------------------------------------------
# implicit GEMM: linear and sparse conv
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,cin,cout", [(1, 4, 4), (67, 32, 96), (1000, 64, 19), (513, 128, 512), (4099, 512, 24),
                                         (300, 36, 64), (9001, 128, 320), (8300, 256, 72)])
def test_linear_fp32(dev, m, cin, cout):
    from ptv3_hip import ops
    g = torch.Generator().manual_seed(m + cin)
    x = torch.randn(m, cin, generator=g)
    w = torch.randn(cout, cin, generator=g) / cin ** 0.5
    b = torch.randn(cout, generator=g)
    ref = torch.nn.functional.linear(x, w, b)
    out = ops.gemm(x.to(dev), w.to(dev), bias=b.to(dev))
    assert (out.cpu() - ref).abs().max().item() < FP32_TOL
    # full epilogue: folded BN, GELU, indexed residual, dual output
    s, t = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    res = torch.randn(17, cout, generator=g)
    ridx = torch.randint(0, 17, (m,), generator=g, dtype=torch.int32)
    pre = torch.nn.functional.gelu(ref * s + t)
    o1, o2 = ops.gemm(x.to(dev), w.to(dev), bias=b.to(dev), bn_scale=s.to(dev), bn_shift=t.to(dev),
                      act=ops.ACT_GELU, res=res.to(dev), res_index=ridx.to(dev), dual=True)
    assert (o1.cpu() - pre).abs().max().item() < FP32_TOL
    assert (o2.cpu() - (pre + res[ridx.long()])).abs().max().item() < FP32_TOL


@pytest.mark.parametrize("m", [777, 9001])
def test_linear_bf16(dev, m):
    from ptv3_hip import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(m, 128, generator=g)
    w = torch.randn(96, 128, generator=g) / 128 ** 0.5
    b = torch.randn(96, generator=g)
    ref = torch.nn.functional.linear(x.bfloat16().float(), w.bfloat16().float(), b)
    out = ops.gemm(x.to(dev).bfloat16(), w.to(dev).bfloat16(), bias=b.to(dev))
    assert (out.float().cpu() - ref).abs().max().item() < 5e-2


@pytest.mark.parametrize("n,k,extent", [(20000, 5, 96), (20000, 3, 64), (5000, 7, 40), (1, 3, 8), (300, 5, 7)])
def test_subm_neighbors_symmetric_fill_equals_full_probing(dev, n, k, extent):
    """ptv3_subm_neighbors probes half the taps and fills j = nbr[i][d] and i = nbr[j][kvol-1-d] from one hit (the
    relation of a submanifold conv is symmetric); PTV3_NBR_SYMMETRIC=0 probes every tap.  Whole tables, bitwise - two
    scenes in the batch, sites at the coordinate origin (taps at negative coordinates) and a dense little cube."""
    import os
    from ptv3_hip import ops
    import ptv3_scenes as S
    if n > 1:
        data = S.make_batch([n - n // 3, n // 3], in_channels=4, extent=extent, seed=n + k)
        off = data["offset"]
        batch = torch.repeat_interleave(torch.arange(2), torch.diff(off, prepend=torch.zeros(1, dtype=torch.long)))
        idx = torch.cat([batch[:, None].int(), data["grid_coord"].int()], 1).contiguous()
    else:
        idx = torch.zeros(1, 4, dtype=torch.int32)
    idx = idx.to(dev)
    half, _ = ops.subm_neighbors(idx, k)
    os.environ["PTV3_NBR_SYMMETRIC"] = "0"
    try:
        full, _ = ops.subm_neighbors(idx, k)
    finally:
        os.environ.pop("PTV3_NBR_SYMMETRIC", None)
    assert torch.equal(half, full)
    assert (half[:, k ** 3 // 2].cpu() == torch.arange(idx.shape[0])).all()


@pytest.mark.parametrize("n,cin,cout,k", [(3000, 4, 32, 5), (2500, 32, 32, 3), (1200, 64, 64, 3), (700, 8, 16, 3)])
def test_subm_conv_vs_oracle(dev, n, cin, cout, k):
    from ptv3_hip import ops
    from oracle import ptv3 as O
    import ptv3_scenes as S
    data = S.make_batch([n - n // 3, n // 3], in_channels=cin, extent=48, seed=n)
    off = data["offset"]
    batch = torch.repeat_interleave(torch.arange(2), torch.diff(off, prepend=torch.zeros(1, dtype=torch.long)))
    idx = torch.cat([batch[:, None].int(), data["grid_coord"].int()], 1).contiguous()
    g = torch.Generator().manual_seed(1)
    w = torch.randn(cout, k, k, k, cin, generator=g) / (cin * k ** 3) ** 0.5
    b = torch.randn(cout, generator=g)
    ref = O.subm_conv3d(data["feat"], idx, w, b)
    nbr, table = ops.subm_neighbors(idx.to(dev), k)
    # neighbour table is integer work: bit-exact against a dictionary lookup
    sites = {tuple(r): i for i, r in enumerate(idx.tolist())}
    nb = nbr.cpu().numpy()
    rng = np.random.default_rng(0)
    for i in rng.integers(0, n, 200):
        bb, x, y, z = idx[i].tolist()
        for d in range(k ** 3):
            a, b_, c = d // (k * k) - k // 2, (d // k) % k - k // 2, d % k - k // 2
            assert nb[i, d] == sites.get((bb, x + a, y + b_, z + c), -1)
    order = torch.randperm(n, generator=g).int()
    for ro in (None, order.to(dev)):
        out = ops.gemm(data["feat"].to(dev), w.reshape(cout, -1).contiguous().to(dev), bias=b.to(dev), nbr=nbr,
                       kvol=k ** 3, row_order=ro)
        assert (out.cpu() - ref).abs().max().item() < FP32_TOL


@pytest.mark.parametrize("c,m", [(32, 1000), (64, 777), (64, 15), (32, 16), (128, 333), (256, 100), (256, 16),
                                 (512, 37)])
def test_fused_block_halves_vs_torch(dev, c, m):
    """ptv3_block_head / ptv3_block_tail (register-chained GEMMs) against plain torch fp32 of the same chain."""
    from ptv3_hip import ops
    F = torch.nn.functional
    g = torch.Generator().manual_seed(c)
    rnd = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    x, shortcut = rnd(m, c), rnd(m, c)
    g0, b0, g1, b1, g2, b2 = (rnd(c) for _ in range(6))
    wqkv, bqkv = rnd(3 * c, c) / c ** 0.5, rnd(3 * c)
    wproj, bproj = rnd(c, c) / c ** 0.5, rnd(c)
    w1, bias1 = rnd(4 * c, c) / c ** 0.5, rnd(4 * c)
    w2, bias2 = rnd(c, 4 * c) / (4 * c) ** 0.5, rnd(c)
    f1_ref = F.layer_norm(x, (c,), g0, b0, 1e-5) + shortcut
    qkv_ref = F.linear(F.layer_norm(f1_ref, (c,), g1, b1, 1e-5), wqkv, bqkv)
    attn = rnd(m, c)
    f2 = F.linear(attn, wproj, bproj) + f1_ref
    out_ref = f2 + F.linear(F.gelu(F.linear(F.layer_norm(f2, (c,), g2, b2, 1e-5), w1, bias1)), w2, bias2)
    d = lambda t: t.to(dev).contiguous()  # noqa: E731
    for dtype, tol in ((torch.float32, FP32_TOL), (torch.bfloat16, 0.15)):
        cv = lambda t: d(t).to(dtype).contiguous()  # noqa: E731
        mode = ops.block_fusable(c, 4 * c, dtype)          # m=0: which kernel variant exists for (c, dtype)
        if mode == 0:                       # fp32 at c=512: the tail's LDS footprint exceeds a CU
            assert (c, dtype) == (512, torch.float32)
            continue
        perm = (lambda w: ops.chain_permute(cv(w), dtype)) if mode == 1 else cv  # noqa: E731
        f1, qkv = ops.block_head(cv(x), None, 0, None, cv(shortcut), d(g0), d(b0), d(g1), d(b1), perm(wqkv), d(bqkv), 1e-5)
        assert (f1.float().cpu() - f1_ref).abs().max().item() < tol
        assert (qkv.float().cpu() - qkv_ref).abs().max().item() < tol
        out = ops.block_tail(cv(attn), cv(f1_ref), cv(wproj), d(bproj), d(g2), d(b2), perm(w1), d(bias1), perm(w2),
                             d(bias2), 1e-5)
        assert (out.float().cpu() - out_ref).abs().max().item() < tol
        # split-K slabs as the head's input: slabs sum + bias == x
        if dtype == torch.float32:
            slabs = torch.stack([x * 0.25, x * 0.5, x * 0.25 - 1.0]).contiguous()
            f1s, _ = ops.block_head(None, d(slabs), 3, d(torch.ones(c)), cv(shortcut), d(g0), d(b0), d(g1), d(b1),
                                    perm(wqkv), d(bqkv), 1e-5)
            assert (f1s.cpu() - f1_ref).abs().max().item() < tol


@pytest.mark.parametrize("c,m", [(128, 24576 + 77), (256, 24576 + 1), (256, 24576 + 127)])
def test_wide_block_halves_vs_torch(dev, c, m):
    """The weight-streaming variant of ptv3_block_head / ptv3_block_tail (csrc/block_wide.hip: c in {128, 256} from
    24 576 rows on; natural weight layout; ragged last workgroup) against plain torch fp32 of the same chain, and - in
    bf16 - against the launches it replaces (LayerNorm + tiled GEMMs) on the same bf16 inputs."""
    from ptv3_hip import ops
    F = torch.nn.functional
    g = torch.Generator().manual_seed(c + m)
    rnd = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    x, shortcut, attn = rnd(m, c), rnd(m, c), rnd(m, c)
    g0, b0, g1, b1, g2, b2 = (rnd(c) for _ in range(6))
    wqkv, bqkv = rnd(3 * c, c) / c ** 0.5, rnd(3 * c)
    wproj, bproj = rnd(c, c) / c ** 0.5, rnd(c)
    w1, bias1 = rnd(4 * c, c) / c ** 0.5, rnd(4 * c)
    w2, bias2 = rnd(c, 4 * c) / (4 * c) ** 0.5, rnd(c)
    f1_ref = F.layer_norm(x, (c,), g0, b0, 1e-5) + shortcut
    qkv_ref = F.linear(F.layer_norm(f1_ref, (c,), g1, b1, 1e-5), wqkv, bqkv)
    f2 = F.linear(attn, wproj, bproj) + f1_ref
    out_ref = f2 + F.linear(F.gelu(F.linear(F.layer_norm(f2, (c,), g2, b2, 1e-5), w1, bias1)), w2, bias2)
    d = lambda t: t.to(dev).contiguous()  # noqa: E731
    for dtype in (torch.float32, torch.bfloat16):
        cv = lambda t: d(t).to(dtype).contiguous()  # noqa: E731
        assert ops.block_fusable(c, 4 * c, dtype, m) == 3
        f1, qkv = ops.block_head(cv(x), None, 0, None, cv(shortcut), d(g0), d(b0), d(g1), d(b1), cv(wqkv), d(bqkv), 1e-5)
        out = ops.block_tail(cv(attn), cv(f1_ref), cv(wproj), d(bproj), d(g2), d(b2), cv(w1), d(bias1), cv(w2), d(bias2), 1e-5)
        for got, ref in ((f1, f1_ref), (qkv, qkv_ref), (out, out_ref)):
            # bf16: inputs, weights and three intermediate activations are rounded to 8 significant bits
            tol = FP32_TOL if dtype == torch.float32 else 8 * 2.0 ** -8 * max(1.0, ref.abs().max().item())
            assert (got.float().cpu() - ref).abs().max().item() < tol
        if dtype == torch.bfloat16:
            f1u, t3 = ops.layernorm(cv(x), d(g0), d(b0), 1e-5, res=cv(shortcut), gamma2=d(g1), beta2=d(b1))
            qkvu = ops.gemm(t3, cv(wqkv), bias=d(bqkv))
            f2u = ops.gemm(cv(attn), cv(wproj), bias=d(bproj), res=cv(f1_ref))
            t6 = ops.gemm(ops.layernorm(f2u, d(g2), d(b2), 1e-5), cv(w1), bias=d(bias1), act=ops.ACT_GELU)
            outu = ops.gemm(t6, cv(w2), bias=d(bias2), res=f2u)
            for got, ref in ((f1, f1u), (qkv, qkvu), (out, outu)):   # a few bf16 steps of the value (summation order, GELU form)
                assert (got.float() - ref.float()).abs().max().item() <= 4 * 2.0 ** -8 * max(1.0, ref.float().abs().max().item())


@pytest.mark.parametrize("c,m", [(128, 300), (256, 1388), (512, 245), (512, 20000 + 77), (256, 9000 + 3), (128, 1)])
def test_rows_linear_vs_torch(dev, c, m):
    """ptv3_rows_linear (csrc/block_wide.hip): the three prologues - none (proj + residual), LayerNorm (norm2 -> fc1 ->
    GELU), LayerNorm + shortcut -> stored f1 -> LayerNorm (cpe norm + shortcut -> norm1 -> qkv) - against torch fp32 and,
    in bf16, against the launches they replace.  Small m runs column groups (several workgroups per row tile), the
    large cases end in a ragged row tile; c = 512 is served in bf16 only (fp32 keeps LayerNorm + tiled GEMM)."""
    from ptv3_hip import ops
    F = torch.nn.functional
    g = torch.Generator().manual_seed(c + m)
    rnd = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    x, shortcut, res = rnd(m, c) * 1.5 + 0.3, rnd(m, c), rnd(m, c)
    g0, b0, g1, b1 = rnd(c), rnd(c), rnd(c), rnd(c)
    wqkv, bqkv = rnd(3 * c, c) / c ** 0.5, rnd(3 * c)
    wproj, bproj = rnd(c, c) / c ** 0.5, rnd(c)
    w1, bias1 = rnd(4 * c, c) / c ** 0.5, rnd(4 * c)
    f1_ref = F.layer_norm(x, (c,), g0, b0, 1e-5) + shortcut
    qkv_ref = F.linear(F.layer_norm(f1_ref, (c,), g1, b1, 1e-5), wqkv, bqkv)
    proj_ref = F.linear(x, wproj, bproj) + res
    fc1_ref = F.gelu(F.linear(F.layer_norm(x, (c,), g1, b1, 1e-5), w1, bias1))
    d = lambda t: t.to(dev).contiguous()  # noqa: E731
    for dtype in (torch.float32, torch.bfloat16):
        if not ops.rows_linear_capable(c, 3 * c, dtype, m):
            assert (c, dtype) == (512, torch.float32)
            continue
        cv = lambda t: d(t).to(dtype).contiguous()  # noqa: E731
        f1, qkv = ops.rows_linear(cv(x), cv(wqkv), d(bqkv), ln=(d(g1), d(b1)), ln0=(d(g0), d(b0)), shortcut=cv(shortcut))
        proj = ops.rows_linear(cv(x), cv(wproj), d(bproj), res=cv(res))
        fc1 = ops.rows_linear(cv(x), cv(w1), d(bias1), act=ops.ACT_GELU, ln=(d(g1), d(b1)))
        for got, ref in ((f1, f1_ref), (qkv, qkv_ref), (proj, proj_ref), (fc1, fc1_ref)):
            tol = FP32_TOL if dtype == torch.float32 else 8 * 2.0 ** -8
            assert (got.float().cpu() - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item())
        if dtype == torch.bfloat16:
            f1u, t3 = ops.layernorm(cv(x), d(g0), d(b0), 1e-5, res=cv(shortcut), gamma2=d(g1), beta2=d(b1))
            pairs = ((f1, f1u), (qkv, ops.gemm(t3, cv(wqkv), bias=d(bqkv))),
                     (proj, ops.gemm(cv(x), cv(wproj), bias=d(bproj), res=cv(res))),
                     (fc1, ops.gemm(ops.layernorm(cv(x), d(g1), d(b1), 1e-5), cv(w1), bias=d(bias1), act=ops.ACT_GELU)))
            for got, ref in pairs:   # a few bf16 steps of the value (summation order; the polynomial GELU of the bf16 path)
                assert (got.float() - ref.float()).abs().max().item() <= 4 * 2.0 ** -8 * max(1.0, ref.float().abs().max().item())


# ------------------------------------------------------------------------------------------------
# norms
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,c", [(1, 4), (1000, 32), (333, 48), (257, 64), (100, 256), (65, 512), (9, 2048)])
def test_layernorm(dev, m, c):
    from ptv3_hip import ops
    g = torch.Generator().manual_seed(c)
    x = torch.randn(m, c, generator=g) * 3 + 1
    g1, b1, g2, b2 = (torch.randn(c, generator=g) for _ in range(4))
    res = torch.randn(m, c, generator=g)
    F = torch.nn.functional
    y_ref = F.layer_norm(x, (c,), g1, b1, 1e-5) + res
    y2_ref = F.layer_norm(y_ref, (c,), g2, b2, 1e-5)
    y, y2 = ops.layernorm(x.to(dev), g1.to(dev), b1.to(dev), 1e-5, res=res.to(dev), gamma2=g2.to(dev), beta2=b2.to(dev))
    assert (y.cpu() - y_ref).abs().max().item() < FP32_TOL
    assert (y2.cpu() - y2_ref).abs().max().item() < FP32_TOL
    y = ops.layernorm(x.to(dev), g1.to(dev), b1.to(dev), 1e-5)
    assert (y.cpu() - F.layer_norm(x, (c,), g1, b1, 1e-5)).abs().max().item() < FP32_TOL


def test_affine_act_and_cast(dev):
    from ptv3_hip import ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1001, 32, generator=g)
    s, t = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g)
    y = ops.affine_act(x.to(dev), s.to(dev), t.to(dev), ops.ACT_GELU)
    assert (y.cpu() - torch.nn.functional.gelu(x * s + t)).abs().max().item() < 1e-5
    assert torch.equal(ops.cast(x.to(dev), torch.bfloat16).cpu(), x.bfloat16())


# ------------------------------------------------------------------------------------------------
# pooling
# ------------------------------------------------------------------------------------------------
def test_pooling_vs_oracle(dev):
    from ptv3_hip import ops
    from oracle import sfc
    import ptv3_scenes as S
    data = S.make_batch([5000, 3000], in_channels=16, extent=64, seed=11)
    off = data["offset"]
    batch = torch.repeat_interleave(torch.arange(2), torch.diff(off, prepend=torch.zeros(1, dtype=torch.long)))
    code, order, inverse, depth = sfc.serialization(data["grid_coord"].numpy(), batch.numpy(), ORDERS)
    code_t = torch.from_numpy(code)
    cluster, seg_start, n_out = ops.pool_segments(code_t[0].contiguous().to(dev),
                                                  torch.from_numpy(order[0]).contiguous().to(dev), 3)
    uniq, ref_cluster = torch.unique(code_t[0] >> 3, sorted=True, return_inverse=True)
    assert n_out == len(uniq)
    assert torch.equal(cluster.cpu(), ref_cluster)
    feat_out, coord_out, grid_out, batch_out, code_out = ops.pool_reduce(
        data["feat"].to(dev), data["coord"].to(dev), data["grid_coord"].to(dev), batch.to(dev), code_t.to(dev),
        torch.from_numpy(order[0]).contiguous().to(dev), seg_start, n_out, 1)
    ref_feat = torch.full((n_out, 16), -float("inf")).scatter_reduce(
        0, ref_cluster[:, None].expand(-1, 16), data["feat"], "amax")
    assert torch.equal(feat_out.cpu(), ref_feat)
    cnt = torch.bincount(ref_cluster, minlength=n_out).float()
    ref_coord = torch.zeros(n_out, 3).index_add_(0, ref_cluster, data["coord"]) / cnt[:, None]
    assert (coord_out.cpu() - ref_coord).abs().max().item() < 1e-5
    # parent codes for every order == encode(grid >> 1, depth - 1)
    ref_code = np.stack([sfc.encode(grid_out.cpu().numpy(), batch_out.cpu().numpy(), depth - 1, o) for o in ORDERS])
    assert np.array_equal(code_out.cpu().numpy(), ref_code)
    assert np.array_equal(code_out[0].cpu().numpy(), uniq.numpy())


# ------------------------------------------------------------------------------------------------
# pointops
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nsample", [1, 3, 16, 70])
def test_knn_grouping_interpolation_vs_c_oracle(dev, nsample):
    import pointops
    from oracle import pointops as OP
    rng = np.random.default_rng(nsample)
    xyz = rng.normal(size=(3000, 3)).astype(np.float32)
    new_xyz = rng.normal(size=(1700, 3)).astype(np.float32)
    offset = np.array([1000, 1040, 3000], dtype=np.int32)
    new_offset = np.array([600, 900, 1700], dtype=np.int32)
    ref_idx, ref_d2 = OP.knn_query(nsample, xyz, offset, new_xyz, new_offset)
    idx, dist = pointops.knn_query(nsample, torch.from_numpy(xyz).to(dev), torch.from_numpy(offset).to(dev),
                                   torch.from_numpy(new_xyz).to(dev), torch.from_numpy(new_offset).to(dev))
    assert np.array_equal(idx.cpu().numpy(), ref_idx)
    assert np.array_equal(dist.cpu().numpy(), np.sqrt(ref_d2))
    feat = rng.normal(size=(3000, 20)).astype(np.float32)
    if nsample <= 16 and (ref_idx >= 0).all():
        grouped = pointops.grouping2(torch.from_numpy(feat).to(dev), idx)
        assert np.array_equal(grouped.cpu().numpy(), OP.grouping_forward(feat, ref_idx))
    if nsample == 3:
        out = pointops.interpolation(torch.from_numpy(xyz).to(dev), torch.from_numpy(new_xyz).to(dev),
                                     torch.from_numpy(feat).to(dev), torch.from_numpy(offset).to(dev),
                                     torch.from_numpy(new_offset).to(dev), k=3)
        d = np.sqrt(ref_d2)
        w = (1.0 / (d + 1e-8))
        w = (w / w.sum(1, keepdims=True)).astype(np.float32)
        ref = OP.interpolation_forward(feat, ref_idx, w)
        assert np.abs(out.cpu().numpy() - ref).max() < 1e-5


def _assert_knn_equal(idx, d, ref_idx, ref_d2):
    """Indices AND distances bit-exact, ties included: the kernel keeps the reference's heap (reheap / heap_sort,
    knn_query_cuda_kernel.cu:15-42), so the order inside groups of equal distances is the reference's too."""
    assert np.array_equal(d, np.sqrt(ref_d2))
    assert np.array_equal(idx, ref_idx)


@pytest.mark.parametrize("m,nsample", [(5001, 16), (17003, 16), (5001, 100)])
def test_knn_multi_query_waves_vs_c_oracle(dev, m, nsample):
    """Large query sets run 2 / 4 queries per wave (shared candidate loads); query groups that straddle a scene
    boundary take the one-by-one path.  Indices and distances stay bit-exact against the C oracle."""
    import pointops
    from oracle import pointops as OP
    rng = np.random.default_rng(m)
    xyz = rng.normal(size=(2500, 3)).astype(np.float32)
    new_xyz = rng.normal(size=(m, 3)).astype(np.float32)
    offset = np.array([700, 1503, 2500], dtype=np.int32)
    new_offset = np.array([m // 3 + 1, 2 * m // 3 + 3, m], dtype=np.int32)   # not multiples of 2 or 4
    ref_idx, ref_d2 = OP.knn_query(nsample, xyz, offset, new_xyz, new_offset)
    idx, dist = pointops.knn_query(nsample, torch.from_numpy(xyz).to(dev), torch.from_numpy(offset).to(dev),
                                   torch.from_numpy(new_xyz).to(dev), torch.from_numpy(new_offset).to(dev))
    _assert_knn_equal(idx.cpu().numpy(), dist.cpu().numpy(), ref_idx, ref_d2)


# ------------------------------------------------------------------------------------------------
# whole model
# ------------------------------------------------------------------------------------------------
def _build(cfg, hidden_dim=256):
    from pointcept.models import build_model
    return build_model(dict(type="OffsetKeypointPTv3", num_keypoints=6, hidden_dim=hidden_dim,
                            backbone_conf=dict(type="PT-v3m1", **cfg)))


def test_tiny_model_golden(dev, golden_dir):
    """The reference's own output (tests/golden/ptv3_tiny.npz) reproduced by the HIP model."""
    g = _g(golden_dir, "ptv3_tiny.npz")
    model = _build(TINY_CFG, hidden_dim=32)
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd_")}
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    data = {k[3:]: torch.from_numpy(g[k]).to(dev) for k in g.files if k.startswith("in_")}
    taps = {}

    def mk(name):
        def hook(m, i, o):
            taps[name] = o.feat.float().cpu().numpy()
            if name == "embedding":
                taps["serialized_code"] = o.serialized_code.cpu().numpy()
                taps["serialized_order"] = o.serialized_order.cpu().numpy()
            if name.startswith("enc"):
                taps[name + "_code"] = o.serialized_code.cpu().numpy()
                taps[name + "_order"] = o.serialized_order.cpu().numpy()
                taps[name + "_grid_coord"] = o.grid_coord.cpu().numpy()
                if "pooling_inverse" in o.keys():
                    taps[name + "_pooling_inverse"] = o.pooling_inverse.cpu().numpy()
        return hook

    bb = model.backbone
    bb.embedding.register_forward_hook(mk("embedding"))
    for s in range(5):
        getattr(bb.enc, f"enc{s}").register_forward_hook(mk(f"enc{s}"))
    for s in range(4):
        getattr(bb.dec, f"dec{s}").register_forward_hook(mk(f"dec{s}"))
    torch.manual_seed(int(g["shuffle_seed"]))
    with torch.no_grad():
        out = model(data)
    # integer state: bit exact
    assert np.array_equal(taps["serialized_code"], g["tap_serialized_code"])
    assert np.array_equal(taps["serialized_order"], g["tap_serialized_order"])
    for s in range(5):
        assert np.array_equal(taps[f"enc{s}_code"], g[f"tap_enc{s}_code"]), s
        assert np.array_equal(taps[f"enc{s}_order"], g[f"tap_enc{s}_order"]), s
        assert np.array_equal(taps[f"enc{s}_grid_coord"], g[f"tap_enc{s}_grid_coord"]), s
        if s > 0:
            assert np.array_equal(taps[f"enc{s}_pooling_inverse"], g[f"tap_enc{s}_pooling_inverse"]), s
    for k in ["embedding"] + [f"enc{s}" for s in range(5)] + [f"dec{s}" for s in (3, 2, 1, 0)]:
        err = np.abs(taps[k] - g["tap_" + k]).max()
        assert err < FP32_TOL, (k, err)
    assert np.abs(out["pred"].cpu().numpy() - g["pred"]).max() < FP32_TOL
    assert abs(out["loss"].item() - float(g["loss"])) < FP32_TOL


def test_native_executor_matches_module_path_and_golden(dev, golden_dir):
    """ptv3_forward (one C-ABI call) against the module-by-module path: bitwise equal in fp32 (same kernels,
    same order), and against the reference's own output."""
    g = _g(golden_dir, "ptv3_tiny.npz")
    model = _build(TINY_CFG, hidden_dim=32)
    model.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd_")}, strict=True)
    model = model.to(dev).eval()
    data = {k[3:]: torch.from_numpy(g[k]).to(dev) for k in g.files if k.startswith("in_")}
    outs = {}
    for use in (True, False):
        model.backbone.use_engine = use
        torch.manual_seed(int(g["shuffle_seed"]))
        with torch.no_grad():
            outs[use] = model(data)
    assert torch.equal(outs[True]["pred"], outs[False]["pred"])
    assert np.abs(outs[True]["pred"].cpu().numpy() - g["pred"]).max() < FP32_TOL
    assert abs(outs[True]["loss"].item() - float(g["loss"])) < FP32_TOL
    # backbone-only call (DefaultSegmentorV2 route): Point with features in input order + serialization keys
    model.backbone.use_engine = True
    torch.manual_seed(int(g["shuffle_seed"]))
    with torch.no_grad():
        pt = model.backbone(data)
    assert np.abs(pt.feat.cpu().numpy() - g["tap_dec0"]).max() < FP32_TOL
    assert np.array_equal(pt.serialized_code.cpu().numpy(), g["tap_serialized_code"])
    assert np.array_equal(pt.serialized_order.cpu().numpy(), g["tap_serialized_order"])


def test_pipelined_calls_match_serial(dev):
    """inputs_resident=True lets the geometry of call i+1 run under the feature tail of call i (two geometry
    arenas, events between the two streams).  Back-to-back calls on different scenes, no synchronisation in
    between, must give exactly the serial results.  Scene sizes go up and down on purpose: the arena parts must
    sit at fixed offsets (ptv3_forward_io.arena_n) - laid out per call they would slide under the kernels of the
    call still in flight (this was an intermittent failure of the first version)."""
    import ptv3_scenes as S
    torch.manual_seed(3)
    model = _build(TINY_CFG, hidden_dim=32).to(dev).eval()
    scenes = [{k: v.to(dev) for k, v in S.make_batch(sz, in_channels=4, extent=64, seed=40 + i).items()}
              for i, sz in enumerate([[4000, 2500], [1500], [5000, 1000, 800], [2500, 2500], [9000, 300], [3000]])]
    torch.cuda.synchronize()
    serial = []
    for sc in scenes:
        torch.manual_seed(9)
        with torch.no_grad():
            serial.append(model(sc)["pred"].clone())
        torch.cuda.synchronize()
    model.backbone.inputs_resident = True
    outs = []
    for rep in range(3):
        for sc in scenes:
            torch.manual_seed(9)
            with torch.no_grad():
                outs.append(model(sc)["pred"])
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        assert torch.equal(o, serial[i % len(scenes)]), i
    # throughput mode: whole feature pipelines of consecutive calls overlap on executor-owned streams
    # (ptv3_forward_io.overlap_calls; outputs in a ring of three, consumed here by the model's own clone)
    model.backbone.overlap_calls = True
    outs = []
    for rep in range(4):
        for sc in scenes:
            torch.manual_seed(9)
            with torch.no_grad():
                outs.append(model(sc)["pred"])
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        assert torch.equal(o, serial[i % len(scenes)]), i
    # fp32 features in, bf16 compute: the pad + cast moves into the executor in this mode
    model.backbone.overlap_calls = False
    model.backbone.compute_dtype = torch.bfloat16
    torch.manual_seed(9)
    with torch.no_grad():
        ref16 = model(scenes[0])["pred"].clone()
    model.backbone.overlap_calls = True
    for _ in range(3):
        torch.manual_seed(9)
        with torch.no_grad():
            got16 = model(scenes[0])["pred"]
        assert torch.equal(got16, ref16)


@pytest.mark.parametrize("sizes,kind,extent", [([12000, 9000], "surface", 128), ([15000], "lidar", 1024)])
def test_fork_config_vs_oracle(dev, sizes, kind, extent):
    """configs/my_dataset/offset_keypoint_ptv3.py shape (46M parameters) - HIP model against the oracle
    on the same seeded weights and scene; keypoint-offset L2 and mask logits within 1e-4."""
    from oracle import ptv3 as O
    import ptv3_scenes as S
    torch.manual_seed(1234)
    model = _build(FORK_CFG).eval()
    gen = torch.Generator().manual_seed(99)
    for n, b in model.named_buffers():
        if n.endswith("running_mean"):
            b.copy_(torch.randn(b.shape, generator=gen) * 0.1)
        if n.endswith("running_var"):
            b.copy_(torch.rand(b.shape, generator=gen) + 0.5)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    data = S.make_batch(sizes, in_channels=4, extent=extent, seed=21, kind=kind, with_target=6)
    orc = O.OffsetKeypointOracle(FORK_CFG, sd)
    torch.manual_seed(5)
    with torch.no_grad():
        ref = orc.forward(data)
    model = model.to(dev)
    torch.manual_seed(5)
    with torch.no_grad():
        out = model({k: v.to(dev) for k, v in data.items()})
    pred = out["pred"].cpu()
    l2 = (pred[..., :3] - ref["pred"][..., :3]).norm(dim=-1).max().item()
    prob = (pred[..., 3] - ref["pred"][..., 3]).abs().max().item()
    assert l2 < FP32_TOL, l2
    assert prob < FP32_TOL, prob
    assert abs(out["loss"].item() - ref["loss"].item()) < FP32_TOL
    # bf16 mode stays close to the fp32 result (autocast-style run of config[1])
    model.backbone.compute_dtype = torch.bfloat16
    torch.manual_seed(5)
    with torch.no_grad():
        out16 = model({k: v.to(dev) for k, v in data.items()})
    # 30 blocks of bf16 activations: max-abs error within 64 bf16 steps of the largest logit (the bound of
    # tests/test_hip_flash_seg.py), and a mean error of a few steps
    err = (out16["pred"].float().cpu() - ref["pred"]).abs()
    scale = max(1.0, ref["logits"].abs().max().item())     # pred = (offset, sigmoid(mask logit)) of the head's logits
    assert err.max().item() < 64 * 2.0 ** -8 * scale, (err.max().item(), scale)
    assert err.mean().item() < 8 * 2.0 ** -8 * scale, (err.mean().item(), scale)


# ------------------------------------------------------------------------------------------------
# keypoint aggregation (evaluator hook / infer_offset post-step)
# ------------------------------------------------------------------------------------------------
def _kp_case(seed, sizes, K=6, empty_kp=True):
    g = torch.Generator().manual_seed(seed)
    n = sum(sizes)
    coord = torch.randn(n, 3, generator=g)
    pred = torch.randn(n, K, 4, generator=g) * 0.2
    pred[..., 3] = torch.rand(n, K, generator=g)
    target = torch.randn(n, K, 4, generator=g) * 0.2
    target[..., 3] = (torch.rand(n, K, generator=g) > 0.9).float()
    if empty_kp:
        target[: sizes[0], 2, 3] = 0      # keypoint 2 has no valid point in scene 0
        pred[: sizes[0], 1, 3] = 0.1      # keypoint 1 never passes the weighted threshold in scene 0
    offset = torch.tensor(sizes).cumsum(0)
    scale = torch.rand(len(sizes), generator=g) + 0.5
    centroid = torch.randn(len(sizes), 3, generator=g)
    return coord, pred, target, offset, scale, centroid


@pytest.mark.parametrize("sizes", [[700, 1300, 50], [5000], [1, 2, 300]])
def test_keypoint_aggregate_vs_reference_loops(dev, sizes):
    from ptv3_hip import ops
    from oracle import keypoints as KO
    coord, pred, target, offset, scale, centroid = _kp_case(len(sizes), sizes)
    d = lambda t: t.to(dev)  # noqa: E731
    for method, mode in (("argmax", ops.KP_ARGMAX), ("weighted", ops.KP_WEIGHTED)):
        ref_p, ref_t = KO.infer_keypoints(pred, target, coord, offset, scale, centroid, 6, method, 0.5)
        kp, _ = ops.keypoint_aggregate(d(coord), d(pred), d(offset), mode, d(scale), d(centroid), 0.5)
        if method == "argmax":
            assert torch.equal(kp.cpu(), ref_p)                      # same fp32 statements: bit-exact
        else:
            assert (kp.cpu() - ref_p).abs().max().item() < 1e-5
        gt, idx = ops.keypoint_aggregate(d(coord), d(target), d(offset), ops.KP_GT_FIRST, d(scale), d(centroid))
        assert torch.equal(torch.isnan(gt.cpu()), torch.isnan(ref_t))
        assert torch.equal(torch.nan_to_num(gt.cpu()), torch.nan_to_num(ref_t))
        assert torch.equal(idx.cpu() < 0, torch.isnan(ref_t[..., 0]))
    # evaluator totals (hook): one device vector vs the reference's python accumulation
    from pointcept.engines.hooks.offset_keypoint_evaluator import evaluate_batch
    ref = torch.tensor(KO.evaluator_totals(pred, target, coord, offset, scale, 6))
    got = evaluate_batch(d(pred), d(target), d(coord), d(offset), d(scale)).cpu()
    assert (got - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())
    assert torch.equal(got[1], ref[1]) and torch.equal(got[8:], ref[8:])   # sample / per-keypoint counts exact


def test_offset_keypoint_evaluator_hook(dev):
    """The hook's own flow (eval loop, totals, logging, SaveBest metric) with a stub trainer."""
    from pointcept.engines.hooks.builder import HOOKS
    from oracle import keypoints as KO
    import types
    coord, pred, target, offset, scale, _ = _kp_case(5, [400, 600])
    hook = HOOKS.build(dict(type="OffsetKeypointEvaluator", num_keypoints=6))
    logs, scalars = [], {}
    trainer = types.SimpleNamespace(
        val_loader=[dict(coord=coord, target=target, offset=offset, scale=scale, _pred=pred)],
        model=None, logger=types.SimpleNamespace(info=logs.append),
        writer=types.SimpleNamespace(add_scalar=lambda k, v, e: scalars.__setitem__(k, v)), epoch=0, comm_info={})

    class M:
        def eval(self):
            return self

        def __call__(self, d):
            return {"pred": d["_pred"]}
    trainer.model = M()
    hook.trainer = trainer
    hook.after_epoch()
    ref = KO.evaluator_totals(pred, target, coord, offset, scale, 6)
    mean = ref[0] / (ref[1] + 1e-6)
    assert abs(trainer.comm_info["current_metric_value"] + mean) < 1e-4
    assert trainer.comm_info["current_metric_name"] == "mean_dist"
    assert abs(scalars["val/MeanDist"] - mean) < 1e-4 and "val/KP_5_MeanDist" in scalars
    assert any("Keypoint 3 Mean Distance" in s for s in logs)


@pytest.mark.parametrize("tag", ["r0", "r1", "r2"])
def test_offset_keypoint_evaluator_vs_reference_hook(dev, golden_dir, tag):
    """This package's hook (device totals, two launches per batch) on the batches the REFERENCE's hook was run on in the
    build container: MeanDist, the per-keypoint means / sample counts and the SaveBest metric it reported."""
    from pointcept.engines.hooks.builder import HOOKS
    import types
    import re
    g = _g(golden_dir, "evaluator.npz")
    loader = []
    for i in range(int(g[tag + "_nbatch"])):
        b = {k: torch.from_numpy(g[f"{tag}_b{i}_{k}"]) for k in ("coord", "target", "offset")}
        b["_pred"] = torch.from_numpy(g[f"{tag}_b{i}_pred"])
        if f"{tag}_b{i}_scale" in g.files:
            b["scale"] = torch.from_numpy(g[f"{tag}_b{i}_scale"])
        loader.append(b)
    hook = HOOKS.build(dict(type="OffsetKeypointEvaluator", num_keypoints=6))
    logs, scalars = [], {}

    class M:
        def eval(self):
            return self

        def __call__(self, d):
            return {"pred": d["_pred"]}
    hook.trainer = types.SimpleNamespace(val_loader=loader, model=M(), logger=types.SimpleNamespace(info=logs.append),
                                         writer=types.SimpleNamespace(add_scalar=lambda k, v, e: scalars.__setitem__(k, v)),
                                         epoch=0, comm_info={})
    hook.after_epoch()
    assert abs(scalars["val/MeanDist"] - float(g[tag + "_mean_dist"])) < 1e-5
    got = np.array([scalars[f"val/KP_{k}_MeanDist"] for k in range(6)])
    assert np.allclose(got, g[tag + "_kp_mean_dist"], atol=1e-5)
    counts = [int(re.search(r"Valid Samples Evaluated: (\d+)", s).group(1)) for s in logs if "Valid Samples" in s]
    assert counts == g[tag + "_kp_counts"].tolist()
    assert abs(hook.trainer.comm_info["current_metric_value"] - float(g[tag + "_metric"])) < 1e-5


# ------------------------------------------------------------------------------------------------
# GridSample on the device (the step before the model)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ci", [0, 1, 2, 3])
def test_gridsample_device_vs_oracle_and_reference(dev, golden_dir, ci):
    from ptv3_hip import ops
    from oracle import gridsample as GS
    from pointcept.datasets import TRANSFORMS
    g = _g(golden_dir, "gridsample.npz")
    t = f"c{ci}_"
    coord, gs, hash_type = g[t + "coord"], float(g[t + "grid_size"]), str(g[t + "hash"])
    p = GS.grid_sample_plan(coord, gs, hash_type)
    # kernel level: voxel coordinates, keys, stable key order, unique - all integer, bit-exact
    grid, mm, key = ops.grid_hash(torch.from_numpy(coord).to(dev), gs, ops.HASH_FNV if hash_type == "fnv" else ops.HASH_RAVEL)
    assert np.array_equal(grid.cpu().numpy(), p["grid"])
    assert np.array_equal(key.cpu().numpy().view(np.uint64), p["key"])
    idx_sort, inverse, seg_start, nvox = ops.voxel_unique(key)
    assert np.array_equal(idx_sort.cpu().numpy(), p["idx_sort"])
    assert np.array_equal(inverse.cpu().numpy(), p["inverse"]) and nvox == len(p["count"])
    assert np.array_equal(np.diff(seg_start.cpu().numpy()), p["count"])
    # transform level, train mode: same numpy RNG draw -> same pick as the stable-sort restatement
    tr = TRANSFORMS.build(dict(type="GridSample", grid_size=gs, hash_type=hash_type, mode="train", return_inverse=True,
                               return_grid_coord=True, return_min_coord=True, return_displacement=True))
    feat = np.arange(len(coord), dtype=np.float32)[:, None] * np.ones((1, 2), np.float32)
    np.random.seed(100 + ci)
    out = tr(dict(coord=coord.copy(), color=feat.copy(), index_valid_keys=["coord", "color"]))
    np.random.seed(100 + ci)
    rand = np.random.randint(0, p["count"].max(), p["count"].size)
    ref = GS.grid_sample_train(coord, gs, hash_type, rand)
    assert np.array_equal(out["color"][:, 0].long().cpu().numpy(), ref["idx_unique"])
    assert np.array_equal(out["coord"].cpu().numpy(), coord[ref["idx_unique"]])
    assert np.array_equal(out["inverse"].cpu().numpy(), ref["inverse"])
    assert np.array_equal(out["displacement"].cpu().numpy(), ref["displacement"])
    # against the reference's own run (order-of-equal-keys independent outputs)
    assert np.array_equal(out["grid_coord"].cpu().numpy(), g[t + "train_grid_coord"])
    assert np.array_equal(out["inverse"].cpu().numpy(), g[t + "train_inverse"])
    assert np.array_equal(out["min_coord"].cpu().numpy(), g[t + "train_min_coord"])
    # test mode: count.max() parts, every point covered, part 0 = first member of every voxel
    te = TRANSFORMS.build(dict(type="GridSample", grid_size=gs, hash_type=hash_type, mode="test", return_grid_coord=True))
    parts = te(dict(coord=coord.copy(), color=feat.copy(), index_valid_keys=["coord", "color"]))
    assert len(parts) == int(g[t + "test_nparts"])
    cover = torch.unique(torch.cat([q["index"] for q in parts])).cpu().numpy()
    assert np.array_equal(cover, g[t + "test_cover"])
    assert np.array_equal(parts[0]["grid_coord"].cpu().numpy(), g[t + "test_grid_coord"])


def test_gridsample_collect_collate_feed_the_model(dev):
    """GridSample -> Collect -> point_collate_fn on the device produces the dict the model consumes (A0)."""
    from pointcept.datasets import TRANSFORMS, point_collate_fn
    rng = np.random.default_rng(3)
    samples = []
    for n in (4000, 2500):
        coord = rng.normal(size=(n, 3)).astype(np.float32)
        coord /= np.abs(coord).max()
        d = dict(coord=coord, normal=rng.normal(size=(n, 3)).astype(np.float32),
                 curvature=rng.random((n, 1)).astype(np.float32), index_valid_keys=["coord", "normal", "curvature"])
        np.random.seed(n)
        d = TRANSFORMS.build(dict(type="GridSample", grid_size=0.02, mode="train", return_grid_coord=True))(d)
        d = TRANSFORMS.build(dict(type="Collect", keys=("coord", "grid_coord"), feat_keys=("coord", "curvature")))(d)
        samples.append(d)
    batch = point_collate_fn(samples)
    assert batch["feat"].shape[1] == 4 and batch["feat"].is_cuda and batch["grid_coord"].dtype == torch.int64
    assert batch["offset"].tolist() == [samples[0]["coord"].shape[0], samples[0]["coord"].shape[0] + samples[1]["coord"].shape[0]]
    # voxels are unique per scene after sampling: the model's serialization contract
    for a, b in zip([0] + batch["offset"].tolist()[:-1], batch["offset"].tolist()):
        gc = batch["grid_coord"][a:b]
        assert torch.unique(gc, dim=0).shape[0] == gc.shape[0]
    model = _build(TINY_CFG, hidden_dim=32).to(dev).eval()
    with torch.no_grad():
        out = model({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()})
    assert out["pred"].shape == (batch["coord"].shape[0], 6, 4) and torch.isfinite(out["pred"]).all()


# ------------------------------------------------------------------------------------------------
# "PT-v3m2" (GridPooling / LayerScale / LayerNorm stem) - SURVEY 8 f2
# ------------------------------------------------------------------------------------------------
def test_v3m2_golden_and_oracle(dev, golden_dir):
    from pointcept.models import build_model
    from oracle import ptv3 as O
    from make_golden_cfg import TINY_M2_CFG
    import ptv3_scenes as S
    g = _g(golden_dir, "ptv3m2_tiny.npz")
    model = build_model(dict(type="PT-v3m2", **TINY_M2_CFG))
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd_")}
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    data = {k[3:]: torch.from_numpy(g[k]).to(dev) for k in g.files if k.startswith("in_")}
    taps = {}

    def mk(name):
        def hook(m, i, o):
            taps[name] = o
        return hook
    for s in range(5):
        getattr(model.enc, f"enc{s}").register_forward_hook(mk(f"enc{s}"))
    torch.manual_seed(int(g["shuffle_seed"]))
    with torch.no_grad():
        point = model(data)
    for s in range(1, 5):   # pooled geometry: integer outputs bit-exact, mean coordinate to fp32 rounding
        o = taps[f"enc{s}"]
        assert np.array_equal(o.grid_coord.cpu().numpy(), g[f"tap_enc{s}_grid_coord"]), s
        assert np.array_equal(o.batch.cpu().numpy(), g[f"tap_enc{s}_batch"]), s
        assert np.array_equal(o.serialized_order.cpu().numpy(), g[f"tap_enc{s}_order"]), s
        assert np.array_equal(o.pooling_inverse.cpu().numpy(), g[f"tap_enc{s}_pooling_inverse"]), s
        assert (o.coord.cpu() - torch.from_numpy(g[f"tap_enc{s}_coord"])).abs().max().item() < 1e-5
    assert (point.feat.cpu() - torch.from_numpy(g["feat"])).abs().max().item() < FP32_TOL
    # a second scene, against the oracle (itself pinned to the reference by test_v3m2_oracle_matches_reference)
    data2 = S.make_batch([2600, 1900, 700], in_channels=4, extent=80, seed=31)
    orc = O.PTv3m2Oracle(TINY_M2_CFG, sd)
    torch.manual_seed(3)
    with torch.no_grad():
        ref = orc.backbone(data2)["feat"]
    torch.manual_seed(3)
    with torch.no_grad():
        out = model({k: v.to(dev) for k, v in data2.items()}).feat
    assert (out.cpu() - ref).abs().max().item() < FP32_TOL
    # segmentor wrapper of the semseg configs (default.py:41-95): backbone + Linear head
    seg = build_model(dict(type="DefaultSegmentorV2", num_classes=13, backbone_out_channels=16,
                           backbone=dict(type="PT-v3m2", **TINY_M2_CFG))).to(dev).eval()
    with torch.no_grad():
        logits = seg({k: v.to(dev) for k, v in data2.items()})["seg_logits"]
    assert logits.shape == (data2["coord"].shape[0], 13) and torch.isfinite(logits).all()


@pytest.mark.parametrize("m,cin,hidden,cout,act", [(1000, 64, 256, 24, "relu"), (777, 32, 128, 64, "gelu"),
                                                   (15, 64, 64, 16, "none"), (4099, 32, 256, 4, "relu")])
def test_mlp2_vs_torch(dev, m, cin, hidden, cout, act):
    """ptv3_mlp2 (the dense keypoint head with its hidden layer in registers) against plain torch fp32."""
    from ptv3_hip import ops
    F = torch.nn.functional
    g = torch.Generator().manual_seed(m)
    x = torch.randn(m, cin, generator=g)
    w1, b1 = torch.randn(hidden, cin, generator=g) / cin ** 0.5, torch.randn(hidden, generator=g)
    s1, t1 = torch.rand(hidden, generator=g) + 0.5, torch.randn(hidden, generator=g)
    w2, b2 = torch.randn(cout, hidden, generator=g) / hidden ** 0.5, torch.randn(cout, generator=g)
    fn = {"relu": F.relu, "gelu": F.gelu, "none": lambda t: t}[act]
    aid = {"relu": ops.ACT_RELU, "gelu": ops.ACT_GELU, "none": ops.ACT_NONE}[act]
    ref = F.linear(fn(F.linear(x, w1, b1) * s1 + t1), w2, b2)
    d = lambda t: t.to(dev).contiguous()  # noqa: E731
    for dtype, tol in ((torch.float32, FP32_TOL), (torch.bfloat16, 0.1)):
        assert ops.mlp2_fusable(cin, hidden, cout, dtype)
        out = ops.mlp2(d(x).to(dtype), d(w1).to(dtype), d(b1), d(s1), d(t1), aid, ops.mlp2_weight2(d(w2), dtype), d(b2), cout)
        assert out.dtype == torch.float32 and out.shape == (m, cout)
        assert (out.cpu() - ref).abs().max().item() < tol
    out16 = ops.mlp2(d(x).bfloat16(), d(w1).bfloat16(), None, None, None, aid, ops.mlp2_weight2(d(w2), torch.bfloat16),
                     None, cout, out_f32=False)
    assert out16.dtype == torch.bfloat16
    assert (out16.float().cpu() - F.linear(fn(F.linear(x, w1)), w2)).abs().max().item() < 0.1


def test_layernorm_slabs(dev):
    from ptv3_hip import ops
    g = torch.Generator().manual_seed(4)
    m, c, splits = 300, 256, 5
    slab = torch.randn(splits, m, c, generator=g)
    bias, g1, b1, g2, b2 = (torch.randn(c, generator=g) for _ in range(5))
    res = torch.randn(m, c, generator=g)
    x = slab.sum(0) + bias
    F = torch.nn.functional
    y_ref = F.layer_norm(x, (c,), g1, b1, 1e-5) + res
    y2_ref = F.layer_norm(y_ref, (c,), g2, b2, 1e-5)
    d = lambda t: t.to(dev).contiguous()  # noqa: E731
    y, y2 = ops.layernorm_slabs(d(slab), splits, m, c, d(bias), torch.float32, d(g1), d(b1), 1e-5, res=d(res),
                                gamma2=d(g2), beta2=d(b2))
    assert (y.cpu() - y_ref).abs().max().item() < FP32_TOL and (y2.cpu() - y2_ref).abs().max().item() < FP32_TOL


def test_model_with_rpe_vs_oracle(dev):
    """enable_rpe=True (configs/s3dis/semseg-pt-v3m1-1-rpe.py style): every block's attention adds the relative
    position bias; the HIP model evaluates it inside the attention kernel, the oracle as the reference does."""
    from pointcept.models import build_model
    from oracle import ptv3 as O
    import ptv3_scenes as S
    cfg = dict(TINY_CFG, enable_rpe=True)
    torch.manual_seed(11)
    model = build_model(dict(type="PT-v3m1", **cfg)).eval()
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("rpe_table"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.5)   # trunc_normal(0.02) would hide the bias
    for n, b in model.named_buffers():
        if n.endswith("running_var"):
            b.copy_(torch.rand(b.shape, generator=g) + 0.5)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    data = S.make_batch([1800, 1200], in_channels=4, extent=48, seed=13)
    orc = O.PTv3Oracle(cfg, sd)
    torch.manual_seed(2)
    with torch.no_grad():
        ref = orc.backbone(data)["feat"]
    model = model.to(dev)
    torch.manual_seed(2)
    with torch.no_grad():
        out = model({k: v.to(dev) for k, v in data.items()}).feat
    assert (out.cpu() - ref).abs().max().item() < FP32_TOL


@pytest.mark.parametrize("sizes", [[40, 17, 3], [2], [70, 1], [129, 64, 65]])
def test_tiny_and_ragged_scenes_vs_oracle(dev, sizes):
    """Degenerate batches: scenes far below the patch size (K shrinks to the smallest scene, down to 1-2 points),
    levels that pool to a single point, windows with more borrowed than own rows - module path and native executor
    against the oracle."""
    from oracle import ptv3 as O
    import ptv3_scenes as S
    torch.manual_seed(21)
    model = _build(TINY_CFG, hidden_dim=32).eval()
    gen = torch.Generator().manual_seed(5)
    for n, b in model.named_buffers():
        if n.endswith("running_var"):
            b.copy_(torch.rand(b.shape, generator=gen) + 0.5)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    data = S.make_batch(sizes, in_channels=4, extent=12, seed=sum(sizes), with_target=6)
    orc = O.OffsetKeypointOracle(TINY_CFG, sd)
    torch.manual_seed(4)
    with torch.no_grad():
        ref = orc.forward(data)
    model = model.to(dev)
    for use_engine in (True, False):
        model.backbone.use_engine = use_engine
        torch.manual_seed(4)
        with torch.no_grad():
            out = model({k: v.to(dev) for k, v in data.items()})
        assert (out["pred"].cpu() - ref["pred"]).abs().max().item() < FP32_TOL, use_engine
