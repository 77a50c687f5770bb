"""ptv3_gemm on the chip-filling shapes of BASELINE configs[2] (120k-point LiDAR scan): 64-point tile vs the large tile.
usage: python tools/bench_gemm_big.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointcept-keypointdetection_amd"))
import torch
from ptv3_hip import ops
import ptv3_scenes as S
dev = torch.device("cuda:0")


def timeit(f, it=10):
    for _ in range(2): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(it): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / it * 1e6


shapes = [("qkv", 56504, 256, 768), ("proj", 56504, 256, 256), ("fc1", 56504, 256, 1024), ("fc2", 56504, 1024, 256),
          ("fc1", 27743, 512, 2048), ("fc2", 27743, 2048, 512), ("fc1", 80168, 128, 512), ("proj", 120000, 64, 64)]
print(f"{'shape':34s}{'64pt us':>9s}{'128x128 us':>12s}{'TF/s':>8s}{'GB/s':>8s}")
for name, m, k, n in ([] if os.environ.get("GEMM_BENCH_CONV_ONLY") else shapes):
    x = torch.randn(m, k, device=dev).bfloat16(); w = torch.randn(n, k, device=dev).bfloat16(); b = torch.randn(n, device=dev)
    os.environ["PTV3_GEMM_BIG"] = "0"
    t0 = timeit(lambda: ops.gemm(x, w, bias=b))
    os.environ["PTV3_GEMM_BIG"] = "2"
    ts = [timeit(lambda: ops.gemm(x, w, bias=b))]
    fl, by = 2.0 * m * k * n, 2.0 * (m * k + n * k + m * n)
    print(f"{name:5s} M={m:<7d}K={k:<5d}N={n:<6d}{t0:9.1f}" + "".join(f"{t:12.1f}" for t in ts) +
          f"{fl / ts[0] / 1e6:8.1f}{by / ts[0] / 1e3:8.0f}", flush=True)
# the sparse convs of the 120k-point LiDAR scan (bench.py's scene): levels 2 / 3 / 4 = strides 4 / 8 / 16, C = 128 / 256 / 512
sc = S.make_scene(120000, 4, 2048, 1000, "lidar")
grid = torch.from_numpy(sc["grid_coord"]).int()
for lvl, C in ((2, 128), (3, 256), (4, 512)):
    idx = torch.unique(torch.cat([torch.zeros(len(grid), 1, dtype=torch.int32), grid >> lvl], 1), dim=0).contiguous().to(dev)
    nbr, _ = ops.subm_neighbors(idx, 3)
    n = idx.shape[0]
    x = torch.randn(n, C, device=dev).bfloat16(); w = torch.randn(C, 27 * C, device=dev).bfloat16()
    act = float((nbr >= 0).float().mean())
    t = timeit(lambda: ops.gemm(x, w, nbr=nbr, kvol=27))
    print(f"conv  M={n:<7d}C={C:<5d}active={act:5.2f} {t:9.1f} us  {2.0 * n * 27 * C * C / t / 1e6:8.1f} dense-equivalent TFLOP/s "
          f"{2.0 * n * 27 * act * C * C / t / 1e6:8.1f} on active pairs", flush=True)
