"""Time ptv3_gemm on the model's actual shapes for each column-tile choice (PTV3_GEMM_NT override)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointcept-keypointdetection_amd"))
os.environ["PTV3_GEMM_TUNE"] = "1"
import torch
from ptv3_hip import ops
dev = torch.device("cuda:0")
Ns = [100000, 21823, 5590, 1388, 245]
shapes = []
encC = [32, 64, 128, 256, 512]; decC = [64, 64, 128, 256]
for s, n in enumerate(Ns):
    for C in sorted(set([encC[s]] + ([decC[s]] if s < 4 else []))):
        shapes += [("qkv", n, C, 3 * C), ("proj", n, C, C), ("fc1", n, C, 4 * C), ("fc2", n, 4 * C, C)]
def timeit(f, it=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(it): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / it * 1e6
print(f"{'shape':28s}" + "".join(f"NT={nt:<8d}" for nt in (2, 4, 6, 8, 12, 16)))
for name, m, k, n in shapes:
    x = torch.randn(m, k, device=dev).bfloat16(); w = torch.randn(n, k, device=dev).bfloat16(); b = torch.randn(n, device=dev)
    row = f"{name:5s} M={m:<7d}K={k:<5d}N={n:<5d}"
    for nt in (2, 4, 6, 8, 12, 16):
        os.environ["PTV3_GEMM_NT"] = str(nt)
        row += f"{timeit(lambda: ops.gemm(x, w, bias=b)):<11.1f}"
    print(row, flush=True)
