"""Seeded synthetic scenes shaped like the reference's collated input dict.

Mimics what OffsetKeypointDataset + GridSample(train) + ShufflePoint + Collect + point_collate_fn
hand to the model (reference: pointcept/datasets/offset_keypoint_dataset.py:116-197,
pointcept/datasets/transform.py:848-895, pointcept/datasets/utils.py:16-69): per scene
``coord (N,3) f32``, ``grid_coord (N,3) int64`` (non-negative, unique rows), ``feat (N,C) f32``;
batched with cumulative ``offset (B,) int64``.  There is no dataset in this repo (none ships
with the reference either), so tests / bench / smoke all draw from here.
"""
import numpy as np
import torch


def _surface_points(rng, n_raw, extent):
    """Union of 3 random ellipsoid shells + a ground patch, coordinates in voxel units."""
    pts = []
    per = n_raw // 4
    for _ in range(3):
        c = rng.uniform(0.3, 0.7, 3) * extent
        r = rng.uniform(0.12, 0.28, 3) * extent
        u = rng.normal(size=(per, 3))
        u /= np.linalg.norm(u, axis=1, keepdims=True)
        pts.append(c + u * r)
    g = np.stack([rng.uniform(0.05, 0.95, n_raw - 3 * per) * extent,
                  rng.uniform(0.05, 0.95, n_raw - 3 * per) * extent,
                  np.full(n_raw - 3 * per, 0.2 * extent)], axis=1)
    pts.append(g)
    p = np.concatenate(pts, 0)
    p += rng.normal(scale=0.25, size=p.shape)
    return p


def _lidar_points(rng, n_raw, extent):
    """64-ring scan hitting a ground plane and a few walls; very anisotropic density."""
    ring = rng.integers(0, 64, n_raw)
    az = rng.uniform(0, 2 * np.pi, n_raw)
    el = np.deg2rad(-24.0 + ring * (26.0 / 63.0))
    h = 1.8
    rng_ground = np.where(el < -0.01, h / np.maximum(np.tan(-el), 1e-3), 1e9)
    r = np.minimum(rng_ground, rng.uniform(5.0, 50.0, n_raw))
    x = r * np.cos(el) * np.cos(az)
    y = r * np.cos(el) * np.sin(az)
    z = h + r * np.sin(el)
    p = np.stack([x, y, z], 1) / 0.05
    p -= p.min(0)
    s = min(1.0, (extent - 1) / p.max())
    return p * s


def _dense_scene(n_points, in_channels, seed, kind):
    """GridSample-like scene: the surface is sampled densely (every surface voxel hit), the extent is
    chosen so that about ``n_points`` voxels are occupied, then the small excess is dropped at random.
    Pools ~4x per stage for kind="surface" (like real voxelised scans), ~1.4x for kind="lidar"."""
    rng = np.random.default_rng(seed)
    lo, hi = 16.0, 8192.0
    for _ in range(24):
        extent = (lo * hi) ** 0.5
        sub = np.random.default_rng(seed + 1)
        fn = _surface_points if kind == "surface" else _lidar_points
        p = fn(sub, int(n_points * (10 if kind == "surface" else 3)), extent)
        g = np.floor(p).astype(np.int64)
        g -= g.min(0)
        _, first = np.unique(g, axis=0, return_index=True)
        if len(first) < n_points:
            lo = extent
        elif len(first) > 1.04 * n_points:
            hi = extent
        else:
            break
    if len(first) < n_points:
        raise RuntimeError("dense scene search failed")
    sel = first[rng.permutation(len(first))[:n_points]]
    grid = g[sel]
    coord = (p[sel] - p[sel].mean(0)).astype(np.float32)
    coord /= np.abs(coord).max() + 1e-6
    feat = rng.normal(size=(n_points, in_channels)).astype(np.float32)
    return {"coord": coord, "grid_coord": grid, "feat": feat}


def make_scene(n_points, in_channels=4, extent=256, seed=0, kind="surface"):
    """One scene with exactly ``n_points`` unique voxels. Returns dict of numpy arrays.
    extent=None: dense mode (see _dense_scene); otherwise a random subset of the voxels inside extent^3."""
    if extent is None:
        return _dense_scene(n_points, in_channels, seed, kind)
    rng = np.random.default_rng(seed)
    n_raw = int(n_points * 2.2) + 1024
    for _ in range(8):
        p = (_surface_points if kind == "surface" else _lidar_points)(rng, n_raw, extent)
        g = np.floor(p).astype(np.int64)
        g -= g.min(0)
        keep = np.all(g < extent, axis=1)
        g, p = g[keep], p[keep]
        _, first = np.unique(g, axis=0, return_index=True)
        if len(first) >= n_points:
            break
        n_raw *= 2
    else:
        raise RuntimeError(f"cannot draw {n_points} unique voxels in extent {extent}")
    sel = first[rng.permutation(len(first))[:n_points]]
    grid = g[sel]
    coord = (p[sel] - p[sel].mean(0)).astype(np.float32)
    coord /= np.abs(coord).max() + 1e-6
    feat = rng.normal(size=(n_points, in_channels)).astype(np.float32)
    return {"coord": coord, "grid_coord": grid, "feat": feat}


def collate(scenes, device="cpu", grid_size=0.02, with_target=0, seed=0):
    """point_collate_fn equivalent: concat + cumulative offset (datasets/utils.py:16-69)."""
    n = [len(s["coord"]) for s in scenes]
    d = {
        "coord": torch.from_numpy(np.concatenate([s["coord"] for s in scenes])).to(device),
        "grid_coord": torch.from_numpy(np.concatenate([s["grid_coord"] for s in scenes])).to(device),
        "feat": torch.from_numpy(np.concatenate([s["feat"] for s in scenes])).to(device),
        "offset": torch.tensor(np.cumsum(n), dtype=torch.int64, device=device),
        "grid_size": torch.full((len(scenes),), grid_size, device=device),
    }
    if with_target:
        g = torch.Generator().manual_seed(seed)
        N = sum(n)
        t = torch.randn(N, with_target, 4, generator=g) * 0.1
        t[..., 3] = (torch.rand(N, with_target, generator=g) > 0.7).float()
        d["target"] = t.to(device)
    return d


def make_batch(sizes, in_channels=4, extent=256, seed=0, kind="surface", device="cpu", with_target=0):
    scenes = [make_scene(n, in_channels, extent, seed + 17 * i, kind) for i, n in enumerate(sizes)]
    return collate(scenes, device=device, with_target=with_target, seed=seed)
