"""The large-M tile of ptv3_gemm (gemm_big_kernel: 128 points x 64 | 128 channels, double-buffered LDS, LDS-resident
neighbour table) against the 64-point tile it replaces at chip-filling sizes and against torch fp32.
PTV3_GEMM_BIG = 0 / 2 switches the policy off / forces the large tile (read per call)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X box"
    return torch.device("cuda:0")


def _both(fn):
    out = {}
    for mode in ("0", "2"):
        os.environ["PTV3_GEMM_BIG"] = mode
        os.environ["PTV3_GEMM_SPLITK"] = "0"     # one pass over K in both: the same summation order
        os.environ["PTV3_CONV_TILE"] = "0"
        try:
            out[mode] = fn()
        finally:
            for k in ("PTV3_GEMM_BIG", "PTV3_GEMM_SPLITK", "PTV3_CONV_TILE"):
                os.environ.pop(k, None)
    return out["0"], out["2"]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("m,cin,cout", [(5000, 64, 64), (777, 256, 768), (3000, 512, 2048), (4099, 2048, 512),
                                         (129, 64, 192), (1, 128, 128), (260, 72, 100)])
def test_dense_linear_big_tile(dev, dtype, m, cin, cout):
    from ptv3_hip import ops
    g = torch.Generator().manual_seed(m + cin)
    x = torch.randn(m, cin, generator=g).to(dev, dtype)
    w = (torch.randn(cout, cin, generator=g) / cin ** 0.5).to(dev, dtype)
    bias = torch.randn(cout, generator=g).to(dev)
    res = torch.randn(m, cout, generator=g).to(dev, dtype)
    small, big = _both(lambda: ops.gemm(x, w, bias=bias, act=ops.ACT_GELU, res=res))
    assert torch.equal(small, big)      # same accumulation order over K: bitwise the same result
    ref = torch.nn.functional.gelu(x.float() @ w.float().t() + bias) + res.float()
    tol = 1e-4 if dtype == torch.float32 else 2.0 ** -7
    assert (big.float() - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n,cin,cout", [(3000, 64, 64), (2500, 128, 128), (1200, 32, 64), (700, 16, 64), (900, 48, 72)])
def test_sparse_conv_big_tile(dev, dtype, n, cin, cout):
    """Gathered rows through the LDS-resident neighbour table, row_order on, K steps that span one or two taps
    (cin 64+ / 32 / 16) and a channel count that is not a power of two; bn / dual-output / indexed-residual epilogue."""
    from ptv3_hip import ops
    import ptv3_scenes as S
    from oracle import ptv3 as O
    if dtype == torch.bfloat16 and cin % 8:
        pytest.skip("bf16 needs cin % 8 == 0")
    sc = S.make_scene(n, cin, 40, seed=n)
    gc = torch.from_numpy(sc["grid_coord"])
    idx = torch.cat([torch.zeros(n, 1, dtype=torch.int32), gc.int()], 1).contiguous()
    g = torch.Generator().manual_seed(n)
    feat = torch.randn(n, cin, generator=g)
    w = torch.randn(cout, 3, 3, 3, cin, generator=g) / (27 * cin) ** 0.5
    bias = torch.randn(cout, generator=g)
    nbr, _ = ops.subm_neighbors(idx.to(dev), 3)
    order = torch.randperm(n, generator=g).int().to(dev)
    scale, shift = torch.rand(cout, generator=g).to(dev) + 0.5, torch.randn(cout, generator=g).to(dev)
    ridx = torch.randint(0, 50, (n,), generator=g).int().to(dev)
    res = torch.randn(50, cout, generator=g).to(dev, dtype)
    xd, wd = feat.to(dev, dtype), w.reshape(cout, -1).to(dev, dtype).contiguous()

    def run():
        return ops.gemm(xd, wd, bias=bias.to(dev), nbr=nbr, kvol=27, row_order=order, bn_scale=scale, bn_shift=shift,
                        act=ops.ACT_RELU, res=res, res_index=ridx, dual=True)
    (s0, s1), (b0, b1) = _both(run)
    assert torch.equal(s0, b0) and torch.equal(s1, b1)
    ref = O.subm_conv3d(xd.float().cpu(), idx, wd.float().cpu().reshape(w.shape), bias)
    ref = torch.relu(ref * scale.cpu() + shift.cpu())
    tol = 1e-4 if dtype == torch.float32 else 2.0 ** -6
    assert (b0.float().cpu() - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item())
    ref2 = ref + res.float().cpu()[ridx.cpu().long()]
    assert (b1.float().cpu() - ref2).abs().max().item() < tol * max(1.0, ref2.abs().max().item())


def _tiles(fn):
    """the same call through the 64-point tile and through the 256-point LDS-DMA tile (conv_tile_kernel)"""
    out = {}
    for name, env in (("small", {"PTV3_GEMM_BIG": "0", "PTV3_CONV_TILE": "0", "PTV3_GEMM_SPLITK": "0"}), ("tile", {"PTV3_CONV_TILE": "2"})):
        os.environ.update(env)
        try:
            out[name] = fn()
        finally:
            for k in env:
                os.environ.pop(k, None)
    return out["small"], out["tile"]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n,cin,cout", [(3000, 128, 128), (700, 256, 256), (1537, 64, 384), (300, 128, 136), (257, 512, 512)])
def test_sparse_conv_256_point_tile(dev, dtype, n, cin, cout):
    """conv_tile_kernel: operands by LDS-DMA through buffer descriptors - missing neighbours, rows past m (n is not a
    multiple of 256) and channel rows past cout (384 and 136 leave a partial channel tile) arrive in LDS as zeros.
    Bitwise against the 64-point tile (same K order), every epilogue option on; then against the conv restatement."""
    from ptv3_hip import ops
    import ptv3_scenes as S
    from oracle import ptv3 as O
    sc = S.make_scene(n, 4, 24, seed=n)
    idx = torch.cat([torch.zeros(n, 1, dtype=torch.int32), torch.from_numpy(sc["grid_coord"]).int()], 1).contiguous()
    g = torch.Generator().manual_seed(n)
    feat = torch.randn(n, cin, generator=g)
    w = torch.randn(cout, 3, 3, 3, cin, generator=g) / (27 * cin) ** 0.5
    bias = torch.randn(cout, generator=g)
    nbr, _ = ops.subm_neighbors(idx.to(dev), 3)
    assert 0.05 < (nbr >= 0).float().mean().item() < 0.9          # the scene has missing neighbours
    order = torch.randperm(n, generator=g).int().to(dev)
    scale, shift = torch.rand(cout, generator=g).to(dev) + 0.5, torch.randn(cout, generator=g).to(dev)
    ridx = torch.randint(0, 50, (n,), generator=g).int().to(dev)
    res = torch.randn(50, cout, generator=g).to(dev, dtype)
    xd, wd = feat.to(dev, dtype), w.reshape(cout, -1).to(dev, dtype).contiguous()

    def run():
        return ops.gemm(xd, wd, bias=bias.to(dev), nbr=nbr, kvol=27, row_order=order, bn_scale=scale, bn_shift=shift,
                        act=ops.ACT_RELU, res=res, res_index=ridx, dual=True)
    (s0, s1), (t0, t1) = _tiles(run)
    assert torch.equal(s0, t0) and torch.equal(s1, t1)
    plain_s, plain_t = _tiles(lambda: ops.gemm(xd, wd, nbr=nbr, kvol=27))
    assert torch.equal(plain_s, plain_t)
    ref = O.subm_conv3d(xd.float().cpu(), idx, wd.float().cpu().reshape(w.shape), bias)
    ref = torch.relu(ref * scale.cpu() + shift.cpu())
    tol = 1e-4 if dtype == torch.float32 else 2.0 ** -6
    assert (t0.float().cpu() - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n,c", [(5600, 128), (1400, 256), (300, 512), (257, 128)])
def test_sparse_conv_split_k_through_the_256_point_tile(dev, dtype, n, c):
    """Deep levels of a 100k-point scene: the conv is split over K into fp32 slabs.  conv_tile_kernel (grid.y = split)
    must leave bitwise the slabs of gemm_kernel's splits (PTV3_CONV_TILE_SPLIT=0): same step boundaries, same order;
    then the reduced + epilogue'd output (out != NULL) the same way, and against the conv restatement."""
    from ptv3_hip import ops
    import ptv3_scenes as S
    from oracle import ptv3 as O
    sc = S.make_scene(n, 4, 24 if n < 2000 else 48, seed=n)
    idx = torch.cat([torch.zeros(n, 1, dtype=torch.int32), torch.from_numpy(sc["grid_coord"]).int()], 1).contiguous()
    g = torch.Generator().manual_seed(n + c)
    feat = torch.randn(n, c, generator=g)
    w = torch.randn(c, 3, 3, 3, c, generator=g) / (27 * c) ** 0.5
    bias = torch.randn(c, generator=g)
    nbr, _ = ops.subm_neighbors(idx.to(dev), 3)
    order = torch.randperm(n, generator=g).int().to(dev)
    xd, wd = feat.to(dev, dtype), w.reshape(c, -1).to(dev, dtype).contiguous()

    def both(fn):
        out = []
        for v in ("0", "1"):
            os.environ["PTV3_CONV_TILE_SPLIT"] = v
            try:
                out.append(fn())
            finally:
                os.environ.pop("PTV3_CONV_TILE_SPLIT", None)
        return out
    (s0, k0), (s1, k1) = both(lambda: ops.conv_slabs(xd, wd, nbr, 27, order))
    assert k0 == k1 > 1
    assert torch.equal(s0, s1)
    o0, o1 = both(lambda: ops.gemm(xd, wd, bias=bias.to(dev), nbr=nbr, kvol=27, row_order=order, act=ops.ACT_RELU))
    assert torch.equal(o0, o1)
    ref = torch.relu(O.subm_conv3d(xd.float().cpu(), idx, wd.float().cpu().reshape(w.shape), bias))
    tol = 1e-4 if dtype == torch.float32 else 2.0 ** -6
    assert (o1.float().cpu() - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item())
