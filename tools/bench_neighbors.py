import os, sys, time, torch
sys.path[:0] = ["/root/repo", "/root/repo/pointcept-keypointdetection_amd"]
from ptv3_hip import ops
import ptv3_scenes as S
dev = torch.device("cuda:0")
sc = S.make_scene(100000, 4, None, seed=0)
g = torch.from_numpy(sc["grid_coord"]).int()
idx = torch.cat([torch.zeros(len(g), 1, dtype=torch.int32), g], 1).contiguous().to(dev)
def t(k):
    for _ in range(3): ops.subm_neighbors(idx, k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ops.subm_neighbors(idx, k)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 20 * 1e6
for k in (5, 3):
    print(os.environ.get("PTV3_NBR_SYMMETRIC", "1"), k, round(t(k), 1), "us (table build + neighbours)")
