"""Time ptv3_gemm_tn (weight gradients) on the shapes of the fork model's training step at 100k points.
usage: python tools/bench_gemm_tn.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pointcept-keypointdetection_amd")]
from ptv3_hip import ops  # noqa: E402
import ptv3_scenes as S  # noqa: E402


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    dev = torch.device("cuda:0")
    scene = S.make_scene(100000, 4, None, seed=0)
    grid = torch.from_numpy(scene["grid_coord"]).int()
    idx = torch.cat([torch.zeros(len(grid), 1, dtype=torch.int32), grid], 1).contiguous().to(dev)
    n = idx.shape[0]
    nbr3, table = ops.subm_neighbors(idx, 3)
    nbr5, _ = ops.subm_neighbors(idx, 5, table)
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda m, c: torch.randn(m, c, device=dev, generator=g).bfloat16()  # noqa: E731
    total = 0.0
    for name, m, cout, cin, nbr in (("stem 5^3 conv", n, 32, 8, nbr5), ("conv c=32", n, 32, 32, nbr3),
                                    ("conv c=64", n, 64, 64, nbr3), ("qkv c=32", n, 96, 32, None),
                                    ("fc1 c=32", n, 128, 32, None), ("fc2 c=32", n, 32, 128, None),
                                    ("fc1 c=64", n, 256, 64, None), ("proj c=64", n, 64, 64, None),
                                    ("fc1 c=128 25k", 25000, 512, 128, None), ("fc2 c=256 6k", 6000, 256, 1024, None)):
        dy, x = rnd(m, cout), rnd(n if nbr is not None else m, cin)
        kv = 1 if nbr is None else nbr.shape[1]
        us = timed(lambda: ops.gemm_tn(dy, x, nbr, kv, with_bias=True))
        fl = 2.0 * m * cout * cin * kv
        total += us
        print(f"{name:16s} m={m:6d} cout={cout:4d} cin={cin:4d} kvol={kv:3d}  {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s")
    print(f"sum {total:.1f} us")


if __name__ == "__main__":
    main()
