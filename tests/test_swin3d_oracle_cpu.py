"""CPU checks of oracle/swin3d.py (row A19, PARITY UNPINNED: the reference's arithmetic for this path lives in
MinkowskiEngine and microsoft/Swin3D, neither in its tree, and it holds no test or fixture for it).  What can be
checked without them: the restated index bookkeeping against a brute-force window partition, and the attention
formula's limits (zero tables = plain softmax attention inside each window; permutation equivariance)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pointcept-keypointdetection_amd"))

from oracle import swin3d as O  # noqa: E402


def _voxels(n, extent, seed, batch=1, stride=1, lo=0):
    rng = np.random.default_rng(seed)
    c = np.unique(rng.integers(lo, lo + extent, size=(n * 2, 3)), axis=0)
    c = c[rng.permutation(len(c))[:n]]
    b = rng.integers(0, batch, size=(len(c), 1))
    return np.concatenate([b, c * stride], axis=1).astype(np.int32)


def test_window_mapping_against_brute_force():
    for ws, shift, stride, lo in ((5, 0, 1, 0), (5, 2, 1, 0), (7, 3, 2, -40), (3, 1, 4, -7)):
        c = _voxels(600, 30, ws + shift, batch=2, stride=stride, lo=lo)
        w_w_id, w_w_xyz, nempty, sort_idx, inv = O.window_mapping(c, stride, ws, shift)
        assert sorted(sort_idx.tolist()) == list(range(len(c))) and np.array_equal(sort_idx[inv], np.arange(len(c)))
        assert nempty.sum() == len(c) and nempty.min() >= 1 and nempty.max() <= ws ** 3
        # every window's members share floor((x/stride + shift)/ws) and the batch; local xyz matches the coordinates
        v = c[:, 1:].astype(np.int64) // stride + shift
        start = 0
        seen = set()
        for m in nempty:
            rows = sort_idx[start:start + m]
            win = np.floor_divide(v[rows], ws)
            assert (win == win[0]).all() and (c[rows, 0] == c[rows[0], 0]).all()
            key = (int(c[rows[0], 0]), *win[0].tolist())
            assert key not in seen
            seen.add(key)
            assert np.array_equal(w_w_xyz[start:start + m], v[rows] - win * ws)
            assert (np.diff(w_w_id[start:start + m]) > 0).all()          # sorted by position inside the window
            start += m
        assert np.array_equal(w_w_id, (w_w_xyz[:, 0] * ws + w_w_xyz[:, 1]) * ws + w_w_xyz[:, 2])


def test_sparse_self_attention_enumerates_each_window_block():
    sizes = np.array([3, 1, 4])
    x, y, m2w, w_sizes, w2n, w2m = O.sparse_self_attention(sizes)
    assert w2n.tolist() == [0, 3, 4] and w2m.tolist() == [0, 9, 10] and len(x) == 26
    want = [(a, b) for s0, m in zip(w2n, sizes) for a in range(s0, s0 + m) for b in range(s0, s0 + m)]
    assert list(zip(x.tolist(), y.tolist())) == want
    assert m2w.tolist() == [0] * 9 + [1] + [2] * 16


def _attn_case(seed, n=300, heads=3, hd=8, ws=5, quant=4, crse="XYZ_RGB_NORM", table_std=0.02):
    rng = np.random.default_rng(seed)
    c = _voxels(n, 14, seed)
    n = len(c)
    w_w_id, w_w_xyz, nempty, sort_idx, _ = O.window_mapping(c, 1, ws, 0)
    nsig = {"XYZ": 0, "XYZ_RGB": 3, "XYZ_RGB_NORM": 6}[crse]
    nc = O.n_coords(w_w_xyz, rng.random((n, 3), dtype=np.float32), rng.uniform(-1, 1, (n, nsig)).astype(np.float32),
                    sort_idx)
    rows = O.table_lengths(ws, quant, crse)
    offs = [r * heads * hd for r in rows for _ in range(3)]
    tabs = [rng.normal(0, table_std, sum(offs)).astype(np.float32) for _ in range(3)]
    q, k, v = (rng.normal(size=(n, heads, hd)).astype(np.float32) for _ in range(3))
    _, _, _, w_sizes, w2n, _ = O.sparse_self_attention(nempty)
    return q, k, v, tabs, offs, w_sizes, w2n, sort_idx, O.n_crse(nc, quant, crse)


def test_crse_attention_with_zero_tables_is_window_softmax_attention():
    q, k, v, tabs, offs, w_sizes, w2n, n2n, cr = _attn_case(0)
    zero = [np.zeros_like(t) for t in tabs]
    out = O.crse_attention(q, k, v, *zero, offs, w_sizes, w2n, n2n, cr)
    for w in range(len(w_sizes)):
        rows = n2n[w2n[w]:w2n[w] + w_sizes[w]]
        lg = np.einsum("ihd,jhd->hij", q[rows].astype(np.float64), k[rows].astype(np.float64))
        p = np.exp(lg - lg.max(-1, keepdims=True))
        p /= p.sum(-1, keepdims=True)
        np.testing.assert_allclose(out[rows], np.einsum("hij,jhd->ihd", p, v[rows]), rtol=1e-5, atol=1e-6)


def test_crse_attention_tables_matter_and_rows_are_independent_of_voxel_order():
    q, k, v, tabs, offs, w_sizes, w2n, n2n, cr = _attn_case(1, table_std=0.5)
    out = O.crse_attention(q, k, v, *tabs, offs, w_sizes, w2n, n2n, cr)
    zero = O.crse_attention(q, k, v, *[np.zeros_like(t) for t in tabs], offs, w_sizes, w2n, n2n, cr)
    assert np.abs(out - zero).max() > 0.05
    perm = np.random.default_rng(2).permutation(len(q))          # relabel the voxels: same sorted tokens
    inv = np.empty_like(perm)
    inv[perm] = np.arange(len(perm))
    out_p = O.crse_attention(q[perm], k[perm], v[perm], *tabs, offs, w_sizes, w2n, inv[n2n], cr)
    np.testing.assert_allclose(out_p, out[perm], rtol=0, atol=0)


def test_index_range_stays_inside_the_tables():
    """xyz differences are strictly inside (-L, L); colour / normal differences reach +-L exactly, which is why the
    restatement clamps (a +2 difference would index row 2L)."""
    q, k, v, tabs, offs, w_sizes, w2n, n2n, cr = _attn_case(3, crse="XYZ_RGB")
    rows = O.table_lengths(5, 4, "XYZ_RGB")
    assert rows == [40, 32]
    m = int(w_sizes[0])
    d = cr[:m, None, :3] - cr[None, :m, :3]
    assert np.floor(d + 20).min() >= 0 and np.floor(d + 20).max() <= 39
    assert np.floor(np.float32(8.0) - np.float32(-8.0) + np.float32(16)) == 32     # white vs black: one past the table


def test_whole_model_oracle_is_equivariant_to_the_order_of_the_input_points():
    """Swin3DOracle.forward (voxel averaging, stem, attention stages, KNN down / up-sampling, classifier, slice back):
    relabelling the input points permutes the output rows and changes nothing else - voxels are numbered by sorted
    coordinate, never by input order.  (A sanity check of the restatement itself; the HIP model is compared with it in
    tests/test_hip_swin3d.py.)"""
    import torch
    from ptv3_hip import configs
    from pointcept.models import build_model
    cfg = dict(configs.TINY_SWIN3D_CFG, depths=[1, 1, 1])
    torch.manual_seed(0)
    model = build_model(cfg)
    sd = {k: v.numpy() for k, v in model.state_dict().items()}
    rng = np.random.default_rng(5)
    g = _voxels(700, 16, 9)[:, 1:].astype(np.int64)
    n = len(g)
    data = {"coord": ((g + rng.random(g.shape)) * 0.02).astype(np.float32), "grid_coord": g,
            "feat": rng.normal(size=(n, 9)).astype(np.float32),
            "coord_feat": rng.uniform(-1, 1, (n, 6)).astype(np.float32), "offset": np.array([n], np.int64)}
    oracle = O.Swin3DOracle(sd, cfg)
    base = oracle.forward(data)
    perm = rng.permutation(n)
    shuffled = {k: (v[perm] if k != "offset" else v) for k, v in data.items()}
    again = oracle.forward(shuffled)
    assert base.shape == (n, 13) and np.isfinite(base).all()
    np.testing.assert_allclose(again, base[perm], rtol=1e-5, atol=1e-5)
