"""Host-side (Python) cost of one forward: cProfile top functions. python tools/profile_host.py [points]"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pointcept-keypointdetection_amd"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch
import ptv3_scenes as S
from pointcept.models import build_model
from make_golden_cfg import FORK_CFG

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build_model(dict(type="OffsetKeypointPTv3", num_keypoints=6, backbone_conf=dict(type="PT-v3m1", **FORK_CFG))).eval().to(dev)
model.backbone.compute_dtype = torch.bfloat16
batch = {k: v.to(dev) for k, v in S.make_batch([n], in_channels=4, extent=None, seed=1000).items()}
for _ in range(3):
    with torch.no_grad():
        model(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    with torch.no_grad():
        model(batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/10:.2f} ms/forward, +drain {1e3*(t2-t1):.2f} ms total", flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    with torch.no_grad():
        model(batch)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
