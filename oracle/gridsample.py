"""Oracle (test infrastructure): numpy restatement of GridSample (pointcept/datasets/transform.py:848-964).

Differs from the reference in one documented point: the reference's `np.argsort(key)` (introsort) leaves the order
of EQUAL keys unspecified, so which member of a voxel `idx_sort[start + r % count]` names is not reproducible from
the source; this restatement (and the device path) use a stable sort.  Everything that does not depend on that
order - voxel set and order, grid_coord, inverse, min_coord, counts - is pinned to the reference's own outputs in
tests/golden/gridsample.npz."""
import numpy as np


def fnv_hash_vec(arr):
    arr = arr.astype(np.uint64)
    h = np.uint64(14695981039346656037) * np.ones(arr.shape[0], dtype=np.uint64)
    for j in range(arr.shape[1]):
        h *= np.uint64(1099511628211)
        h = np.bitwise_xor(h, arr[:, j])
    return h


def ravel_hash_vec(arr):
    arr = arr - arr.min(0)
    arr = arr.astype(np.uint64)
    arr_max = arr.max(0).astype(np.uint64) + 1
    keys = np.zeros(arr.shape[0], dtype=np.uint64)
    for j in range(arr.shape[1] - 1):
        keys += arr[:, j]
        keys *= arr_max[j + 1]
    keys += arr[:, -1]
    return keys


def grid_sample_plan(coord, grid_size, hash_type="fnv"):
    scaled = coord / np.array(grid_size)              # float32 / float64 0-d array -> float64 (:850)
    grid = np.floor(scaled).astype(int)
    mn = grid.min(0)
    grid = grid - mn
    scaled = scaled - mn
    key = (fnv_hash_vec if hash_type == "fnv" else ravel_hash_vec)(grid)
    idx_sort = np.argsort(key, kind="stable")
    key_sort = key[idx_sort]
    _, inverse, count = np.unique(key_sort, return_inverse=True, return_counts=True)
    inv = np.zeros_like(inverse)
    inv[idx_sort] = inverse
    return dict(grid=grid, scaled=scaled, min_coord=(mn * np.array(grid_size)).reshape(1, 3), key=key,
                idx_sort=idx_sort, inverse=inv, count=count,
                start=np.cumsum(np.insert(count, 0, 0)[0:-1]))


def grid_sample_train(coord, grid_size, hash_type, rand):
    """rand = np.random.randint(0, count.max(), count.size) as drawn by the reference (:862-865)."""
    p = grid_sample_plan(coord, grid_size, hash_type)
    idx_unique = p["idx_sort"][p["start"] + rand % p["count"]]
    return dict(idx_unique=idx_unique, grid_coord=p["grid"][idx_unique], inverse=p["inverse"],
                min_coord=p["min_coord"], displacement=(p["scaled"] - p["grid"] - 0.5)[idx_unique], count=p["count"])
