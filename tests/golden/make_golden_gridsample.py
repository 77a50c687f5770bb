"""Build-container-only: run the reference's GridSample (pointcept/datasets/transform.py:826-964, imported in
place with a torchvision stub) on seeded synthetic clouds and store inputs + outputs in gridsample.npz.
usage: python tests/golden/make_golden_gridsample.py"""
import importlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("PTV3_REFERENCE_ROOT", "/root/reference")


def load_transform():
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvt.Compose = object
    tv.transforms = tvt
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt})
    pc = os.path.join(REF, "pointcept")
    for name, path in (("pointcept", pc), ("pointcept.datasets", os.path.join(pc, "datasets")),
                       ("pointcept.utils", os.path.join(pc, "utils"))):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    return importlib.import_module("pointcept.datasets.transform")


def cloud(seed, n, spread):
    rng = np.random.default_rng(seed)
    c = rng.normal(size=(n, 3)).astype(np.float32) * spread
    c /= np.abs(c).max()            # unit-sphere-like normalisation of the dataset (offset_keypoint_dataset.py)
    return c


def main():
    T = load_transform()
    out = {}
    for ci, (n, gs, hash_type) in enumerate([(6000, 0.02, "fnv"), (3000, 0.05, "fnv"), (2000, 0.1, "ravel"),
                                             (1, 0.02, "fnv")]):
        coord = cloud(ci, n, 1.0)
        feat = np.arange(n, dtype=np.float32)[:, None] * np.ones((1, 2), np.float32)
        t = f"c{ci}_"
        out[t + "coord"], out[t + "grid_size"], out[t + "hash"] = coord, np.float64(gs), np.array(hash_type)
        np.random.seed(100 + ci)
        tr = T.GridSample(grid_size=gs, hash_type=hash_type, mode="train", return_inverse=True,
                          return_grid_coord=True, return_min_coord=True, return_displacement=True)
        d = tr(dict(coord=coord.copy(), color=feat.copy(), index_valid_keys=["coord", "color"]))
        out[t + "train_coord"], out[t + "train_point_id"] = d["coord"], d["color"][:, 0].astype(np.int64)
        out[t + "train_grid_coord"], out[t + "train_inverse"] = d["grid_coord"], d["inverse"]
        out[t + "train_min_coord"], out[t + "train_displacement"] = d["min_coord"], d["displacement"]
        te = T.GridSample(grid_size=gs, hash_type=hash_type, mode="test", return_grid_coord=True)
        parts = te(dict(coord=coord.copy(), color=feat.copy(), index_valid_keys=["coord", "color"]))
        out[t + "test_nparts"] = np.int64(len(parts))
        out[t + "test_grid_coord"] = parts[0]["grid_coord"]
        out[t + "test_cover"] = np.unique(np.concatenate([p["index"] for p in parts]))
    np.savez_compressed(os.path.join(HERE, "gridsample.npz"), **out)
    print("wrote gridsample.npz", {k: v.shape for k, v in out.items() if k.startswith("c0_")})


if __name__ == "__main__":
    main()
