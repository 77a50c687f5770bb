"""pointops on the GPU vs the plain-C restatement of libs/pointops (oracle/pointops_oracle.c; the reference's .cu
files need CUDA + ATen headers and cannot be built here, and no reference-held vector exists for these ops:
PARITY UNPINNED against the reference itself, exact against the restatement).

knn_query: indices and distances bit-exact INCLUDING ties - voxel-centre coordinates (the real call sites,
engines/test.py:939-945, hooks/evaluator.py:569-575) make equal distances the common case, and the reference's
order inside a tie group is whatever its heap (reheap / heap_sort, knn_query_cuda_kernel.cu:15-42) leaves.
grouping / interpolation backward (atomicAdd in the reference and here: fp32 sums up to the order of additions)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X box"
    return torch.device("cuda:0")


def _grid_scene(rng, n, extent, scale=0.05):
    g = rng.integers(0, extent, size=(4 * n, 3))
    g = np.unique(g, axis=0)
    g = g[rng.permutation(len(g))[:n]]
    return ((g + 0.5) * scale).astype(np.float32)


@pytest.mark.parametrize("nsample,m", [(1, 3000), (3, 3000), (8, 3000), (16, 20000), (33, 5000), (100, 2500)])
def test_knn_gridded_ties_bit_exact(dev, nsample, m):
    """Candidates and queries on voxel centres of a small grid: most rows hold several exactly equal distances and
    the k-th distance is usually shared by candidates inside and outside the result."""
    import pointops
    from oracle import pointops as OP
    rng = np.random.default_rng(100 + nsample)
    sizes = [1500, 40, 2100]
    xyz = np.concatenate([_grid_scene(rng, s, 14) for s in sizes])
    offset = np.cumsum(sizes).astype(np.int32)
    qsz = [m // 2, 7, m - m // 2 - 7]
    new_xyz = np.concatenate([_grid_scene(rng, s, 40 if s > 4000 else 16) for s in qsz])
    new_offset = np.cumsum(qsz).astype(np.int32)
    ref_idx, ref_d2 = OP.knn_query(nsample, xyz, offset, new_xyz, new_offset)
    ties = (np.diff(ref_d2, axis=1) == 0).any(axis=1).mean() if nsample > 1 else 1.0
    assert ties > 0.5, ties   # the fixture really is tie-dominated
    idx, dist = pointops.knn_query(nsample, torch.from_numpy(xyz).to(dev), torch.from_numpy(offset).to(dev),
                                   torch.from_numpy(new_xyz).to(dev), torch.from_numpy(new_offset).to(dev))
    assert np.array_equal(dist.cpu().numpy(), np.sqrt(ref_d2))
    assert np.array_equal(idx.cpu().numpy(), ref_idx)
    # self-query (new_xyz omitted): every point finds itself first, at distance 0
    sidx, sdist = pointops.knn_query(min(nsample, 8), torch.from_numpy(xyz).to(dev), torch.from_numpy(offset).to(dev))
    r2, d2 = OP.knn_query(min(nsample, 8), xyz, offset)
    assert np.array_equal(sidx.cpu().numpy(), r2) and np.array_equal(sdist.cpu().numpy(), np.sqrt(d2))


def test_knn_descending_stream_and_short_scene(dev):
    """Adversarial order (every candidate beats the current k-th: one heap operation per candidate) and a scene
    with fewer candidates than nsample (-1 / 1e10 padding after the heap sort)."""
    import pointops
    from oracle import pointops as OP
    n = 700
    xyz = np.stack([np.linspace(50.0, 1.0, n), np.zeros(n), np.zeros(n)], 1).astype(np.float32)
    xyz = np.concatenate([xyz, np.array([[0, 1, 0], [0, 2, 0], [0, 2, 0]], dtype=np.float32)])
    offset = np.array([n, n + 3], dtype=np.int32)
    new_xyz = np.zeros((6, 3), dtype=np.float32)
    new_offset = np.array([3, 6], dtype=np.int32)
    for ns in (5, 16, 70):
        ref_idx, ref_d2 = OP.knn_query(ns, xyz, offset, new_xyz, new_offset)
        idx, dist = pointops.knn_query(ns, torch.from_numpy(xyz).to(dev), torch.from_numpy(offset).to(dev),
                                       torch.from_numpy(new_xyz).to(dev), torch.from_numpy(new_offset).to(dev))
        assert np.array_equal(idx.cpu().numpy(), ref_idx), ns
        assert np.array_equal(dist.cpu().numpy(), np.sqrt(ref_d2)), ns
        assert (ref_idx[3:, 3:] == -1).all()


def test_grouping_backward_vs_c_oracle(dev):
    import pointops
    from oracle import pointops as OP
    rng = np.random.default_rng(5)
    n, m, ns, c = 900, 1300, 9, 20
    feat = rng.normal(size=(n, c)).astype(np.float32)
    idx = rng.integers(0, n, size=(m, ns)).astype(np.int32)
    idx[:, 0] = 7                                  # one row collects m contributions per channel
    gout = rng.normal(size=(m, ns, c)).astype(np.float32)
    f = torch.from_numpy(feat).to(dev).requires_grad_(True)
    out = pointops.grouping2(f, torch.from_numpy(idx).to(dev))
    assert np.array_equal(out.detach().cpu().numpy(), OP.grouping_forward(feat, idx))
    out.backward(torch.from_numpy(gout).to(dev))
    ref = OP.grouping_backward(gout, idx, n)
    err = np.abs(f.grad.cpu().numpy() - ref)
    assert (err <= 1e-5 * np.maximum(1.0, np.abs(ref)) * np.sqrt(m)).all(), err.max()
    # -1 entries: a zero row forward (the wrapper's appended row, functions/grouping.py:41-63), no gradient back
    idx2 = idx.copy()
    hole = rng.random(idx2.shape) < 0.2
    idx2[hole] = -1
    f2 = torch.from_numpy(feat).to(dev).requires_grad_(True)
    xyz = torch.from_numpy(rng.normal(size=(n, 3)).astype(np.float32)).to(dev)
    qxyz = torch.from_numpy(rng.normal(size=(m, 3)).astype(np.float32)).to(dev)
    g = pointops.grouping(torch.from_numpy(idx2).to(dev), f2, xyz, qxyz, with_xyz=True)
    assert tuple(g.shape) == (m, ns, 3 + c)
    gn = g.detach().cpu().numpy()
    assert (gn[hole] == 0).all()
    keep = ~hole
    assert np.array_equal(gn[..., 3:][keep], feat[idx2[keep]])
    rel = xyz.cpu().numpy()[idx2[keep]] - np.broadcast_to(qxyz.cpu().numpy()[:, None, :], (m, ns, 3))[keep]
    assert np.array_equal(gn[..., :3][keep], rel.astype(np.float32))
    g[..., 3:].backward(torch.from_numpy(gout).to(dev))
    safe = np.where(hole, 0, idx2)
    ref2 = OP.grouping_backward(np.where(hole[..., None], 0.0, gout).astype(np.float32), safe, n)
    err2 = np.abs(f2.grad.cpu().numpy() - ref2)
    assert (err2 <= 1e-5 * np.maximum(1.0, np.abs(ref2)) * np.sqrt(m)).all(), err2.max()


def test_interpolation_forward_backward_vs_c_oracle(dev):
    import pointops
    from oracle import pointops as OP
    rng = np.random.default_rng(6)
    sizes, qsizes, c, k = [400, 650], [900, 1100], 24, 3
    xyz = np.concatenate([_grid_scene(rng, s, 12) for s in sizes])
    new_xyz = rng.normal(size=(sum(qsizes), 3)).astype(np.float32) * 0.2 + 0.3
    offset, new_offset = np.cumsum(sizes).astype(np.int32), np.cumsum(qsizes).astype(np.int32)
    feat = rng.normal(size=(sum(sizes), c)).astype(np.float32)
    f = torch.from_numpy(feat).to(dev).requires_grad_(True)
    out = pointops.interpolation(torch.from_numpy(xyz).to(dev), torch.from_numpy(new_xyz).to(dev), f,
                                 torch.from_numpy(offset).to(dev), torch.from_numpy(new_offset).to(dev), k=k)
    ref_idx, ref_d2 = OP.knn_query(k, xyz, offset, new_xyz, new_offset)
    w = 1.0 / (np.sqrt(ref_d2) + np.float32(1e-8))
    w = (w / w.sum(1, keepdims=True)).astype(np.float32)
    ref = OP.interpolation_forward(feat, ref_idx, w)
    assert np.abs(out.detach().cpu().numpy() - ref).max() < 1e-5
    gout = rng.normal(size=ref.shape).astype(np.float32)
    out.backward(torch.from_numpy(gout).to(dev))
    gref = OP.interpolation_backward(gout, ref_idx, w, feat.shape[0])
    err = np.abs(f.grad.cpu().numpy() - gref)
    assert (err <= 2e-5 * np.maximum(1.0, np.abs(gref)) * 8).all(), err.max()
    # the two spellings are the same op here
    out2 = pointops.interpolation2(torch.from_numpy(xyz).to(dev), torch.from_numpy(new_xyz).to(dev), f.detach(),
                                   torch.from_numpy(offset).to(dev), torch.from_numpy(new_offset).to(dev), k)
    assert torch.equal(out2, out.detach())


def _sorted_pairs(idx, d2):
    """rows as (distance, index) pairs in ascending pair order - the order knn_query_cells writes."""
    out_i, out_d = np.empty_like(idx), np.empty_like(d2)
    for r in range(idx.shape[0]):
        o = np.lexsort((idx[r], d2[r]))
        out_i[r], out_d[r] = idx[r][o], d2[r][o]
    return out_i, out_d


@pytest.mark.parametrize("nsample,kind", [(1, "grid"), (3, "grid"), (16, "grid"), (16, "float"), (3, "float"),
                                          (70, "float")])
def test_knn_query_cells_against_the_reference_scan(dev, nsample, kind):
    """ptv3_knn_query_cells against the restated reference scan.  Every row holds the same DISTANCES bit for bit; the
    neighbours strictly closer than the row's k-th distance are the same set; a neighbour AT the k-th distance may be a
    different member of that tie group (which of several equidistant candidates survives in the reference depends on
    where its heap happened to hold them when a closer candidate evicted the root) and the order inside any tie group
    is ascending index instead of the heap's by-product.  nsample = 1 has no such freedom: bit-identical.
    Gridded coordinates make ties (also at the k-th distance) the common case; three scenes, one of them smaller than
    nsample (padding with -1 / 1e10), queries far outside the candidates (fallback scan)."""
    from ptv3_hip import ops
    from oracle import pointops as OP
    rng = np.random.default_rng(7 + nsample)
    sizes = [4000, 12, 2500]
    qsizes = [1500, 9, 800]
    if kind == "grid":
        xyz = np.concatenate([_grid_scene(rng, s, 22) for s in sizes])
        new_xyz = np.concatenate([_grid_scene(rng, s, 26) for s in qsizes])
        cell = 0.05
    else:
        xyz = np.concatenate([rng.normal(size=(s, 3)) * [3, 3, 0.2] for s in sizes]).astype(np.float32)
        new_xyz = np.concatenate([rng.normal(size=(s, 3)) * [3, 3, 0.2] for s in qsizes]).astype(np.float32)
        new_xyz[:5] += 40.0                      # outliers: nothing within the shells
        cell = None
    offset = np.cumsum(sizes).astype(np.int32)
    new_offset = np.cumsum(qsizes).astype(np.int32)
    want_i, want_d = OP.knn_query(nsample, xyz, offset, new_xyz, new_offset)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    got_i, got_d = ops.knn_query_cells(nsample, t(xyz), t(offset), t(new_xyz), t(new_offset), cell)
    got_i, got_d = got_i.cpu().numpy(), got_d.cpu().numpy()
    if nsample == 1:
        assert np.array_equal(got_i, want_i) and np.array_equal(got_d, want_d)
    si, sd = _sorted_pairs(got_i, got_d)
    assert np.array_equal(si, got_i) and np.array_equal(sd, got_d)          # rows ascend in (distance, index)
    wi, wd = _sorted_pairs(want_i, want_d)
    assert np.array_equal(got_d, wd)                                        # the same distances, bit for bit
    start = np.concatenate([[0], offset[:-1]])
    scene = np.repeat(np.arange(3), qsizes)
    for r in range(len(new_xyz)):
        real = got_i[r] >= 0
        assert real.sum() == min(nsample, sizes[scene[r]]) and (got_d[r][~real] == 1e10).all()
        ids = got_i[r][real]
        assert len(set(ids.tolist())) == len(ids)
        assert (ids >= start[scene[r]]).all() and (ids < offset[scene[r]]).all()
        d = xyz[ids] - new_xyz[r]
        assert np.array_equal((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2], got_d[r][real])
        inside = got_d[r] < got_d[r][real][-1]
        assert set(got_i[r][inside].tolist()) == set(wi[r][wd[r] < got_d[r][real][-1]].tolist())


@pytest.mark.parametrize("shift,cell", [(250.0, 0.05), (2500.0, 0.05), (40000.0, 1.0)])
def test_knn_query_cells_far_from_the_origin(dev, shift, cell):
    """Coordinates far from the origin (|x / cell| = 5e3 .. 5e4: an outdoor scene in metres at a 5 cm cell, Swin3D voxel
    units past 1000): the cell of a point is taken relative to the bounding box's corner and the shell test keeps a
    margin for its rounding, so the nearest neighbour is still the scan's, bit for bit (nsample = 1 has no tie freedom)
    and for nsample = 8 the distances are."""
    from ptv3_hip import ops
    from oracle import pointops as OP
    rng = np.random.default_rng(int(shift))
    n, m = 6000, 3000
    span = 600 * cell                                     # a few hundred cells across
    xyz = (rng.uniform(0, span, size=(n, 3)) + shift).astype(np.float32)
    new_xyz = (rng.uniform(0, span, size=(m, 3)) + shift).astype(np.float32)
    offset, new_offset = np.array([n], np.int32), np.array([m], np.int32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    for nsample in (1, 8):
        want_i, want_d = OP.knn_query(nsample, xyz, offset, new_xyz, new_offset)
        got_i, got_d = ops.knn_query_cells(nsample, t(xyz), t(offset), t(new_xyz), t(new_offset), cell)
        got_i, got_d = got_i.cpu().numpy(), got_d.cpu().numpy()
        wi, wd = _sorted_pairs(want_i, want_d)
        assert np.array_equal(got_d, wd)
        if nsample == 1:
            assert np.array_equal(got_i, want_i)


def test_knn_rejects_offsets_that_do_not_end_at_the_row_counts(dev):
    """The kernels trust the scene ends; a last entry beyond the arrays would read past them."""
    import pointops
    from ptv3_hip import ops
    xyz = torch.rand(100, 3, device=dev)
    good = torch.tensor([100], dtype=torch.int32, device=dev)
    bad = torch.tensor([1000], dtype=torch.int32, device=dev)
    for fn in (lambda: pointops.knn_query(2, xyz, good, xyz, bad), lambda: ops.knn_query(2, xyz, bad, xyz, good),
               lambda: ops.knn_query_cells(2, xyz, good, xyz, bad)):
        with pytest.raises(ValueError, match="offsets end at"):
            fn()
    idx, _ = pointops.knn_query(1, xyz, good)
    assert torch.equal(idx[:, 0].cpu(), torch.arange(100, dtype=torch.int32))
