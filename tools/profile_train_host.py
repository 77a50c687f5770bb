"""Host-side (Python) cost of one training step: cProfile top functions.  python tools/profile_train_host.py"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pointcept-keypointdetection_amd"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch
import ptv3_scenes as S
from pointcept.models import build_model
from make_golden_cfg import FORK_CFG
from ptv3_hip.optim import FusedAdamW

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build_model(dict(type="OffsetKeypointPTv3", num_keypoints=6, backbone_conf=dict(type="PT-v3m1", **FORK_CFG))).to(dev).train()
model.backbone.compute_dtype = torch.bfloat16
opt = FusedAdamW(model.parameters(), lr=1e-3)
batch = {k: v.to(dev) for k, v in S.collate([S.make_scene(100000, 4, None, 1000, "surface")], with_target=6).items()}


def step():
    opt.zero_grad()
    out = model(batch)
    out["loss"].backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host {1e3*(t1-t0)/5:.2f} ms/step, drain {1e3*(t2-t1):.2f} ms", flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(32)
