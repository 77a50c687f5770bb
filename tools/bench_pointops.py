"""kNN (libs/pointops knn_query) on the GPU: time + vector-fp32 roofline fraction.
usage: python tools/bench_pointops.py [n] [m] [nsample]   -> one JSON line
Algorithmic work = 8 flops per (query, candidate) pair (3 sub, 3 mul, 2 add: the reference's expression, which
must not be fma-contracted for bit-exact indices).  Peak: 157.3 TFLOP/s is the packed-FMA fp32 vector rate
(MI355X_MICROARCH.md); single non-fused fp32 ops issue at a quarter of it (39.3 Tops/s) - both are reported."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pointcept-keypointdetection_amd")]


def bench_cells(ops, k, xyz, off, new_xyz, noff):
    """the cell-grid search on the same input (uniform random points in the unit cube), grid build included"""
    for _ in range(3):
        ops.knn_query_cells(k, xyz, off, new_xyz, noff)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        idx, d2 = ops.knn_query_cells(k, xyz, off, new_xyz, noff)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    ref_i, ref_d = ops.knn_query(k, xyz, off, new_xyz, noff)
    print(json.dumps({"op": "knn_query_cells", "n": xyz.shape[0], "m": new_xyz.shape[0], "nsample": k,
                      "ms_with_grid_build": round(ms, 3), "queries_per_s": round(new_xyz.shape[0] / (ms * 1e-3)),
                      "same_distances_as_scan": bool(torch.equal(torch.sort(ref_d, 1).values, d2)),
                      "same_indices_as_scan": bool(torch.equal(ref_i, idx))}))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 25000
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    cells = len(sys.argv) > 4 and sys.argv[4] == "cells"
    from ptv3_hip import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    xyz = torch.rand(n, 3, generator=g).to(dev)
    new_xyz = (xyz[torch.randperm(n, generator=g)[:m].to(dev)] if m <= n else torch.rand(m, 3, generator=g).to(dev)).contiguous()
    m = new_xyz.shape[0]
    off = torch.tensor([n], dtype=torch.int32, device=dev)
    noff = torch.tensor([m], dtype=torch.int32, device=dev)
    if cells:
        return bench_cells(ops, k, xyz, off, new_xyz, noff)
    for _ in range(3):
        ops.knn_query(k, xyz, off, new_xyz, noff)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        idx, d2 = ops.knn_query(k, xyz, off, new_xyz, noff)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = 8.0 * n * m
    tf = flops / (ms * 1e-3) / 1e12
    print(json.dumps({"op": "knn_query", "n": n, "m": m, "nsample": k, "ms": round(ms, 3),
                      "queries_per_s": round(m / (ms * 1e-3)), "achieved_tflops": round(tf, 3),
                      "roofline": {"bound": "valu_fp32", "peak_packed_fma": 157.3, "frac_packed_fma": round(tf / 157.3, 4),
                                   "peak_unfused_ops": 39.3, "frac_unfused": round(tf / 39.3, 4), "unit": "TFLOP/s"}}))


if __name__ == "__main__":
    main()
