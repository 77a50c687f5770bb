"""Per-module wall time of one forward (debug aid): python tools/trace_forward.py [points] [dtype]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pointcept-keypointdetection_amd"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch
import ptv3_scenes as S
from pointcept.models import build_model
from make_golden_cfg import FORK_CFG

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dtype = torch.bfloat16 if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else torch.float32
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build_model(dict(type="OffsetKeypointPTv3", num_keypoints=6, backbone_conf=dict(type="PT-v3m1", **FORK_CFG))).eval().to(dev)
model.backbone.compute_dtype = dtype
batch = {k: v.to(dev) for k, v in S.make_batch([n], in_channels=4, extent=512, seed=1000).items()}
print("built", flush=True)
t_in = {}
def pre(name):
    def f(m, i):
        torch.cuda.synchronize(); t_in[name] = time.perf_counter()
    return f
def post(name):
    def f(m, i, o):
        torch.cuda.synchronize()
        print(f"{name:40s} {1e3*(time.perf_counter()-t_in[name]):9.3f} ms  n={o.feat.shape[0] if hasattr(o,'feat') else ''}", flush=True)
    return f
for name, mod in model.backbone.named_modules():
    if name.count(".") == 2 and (name.startswith("enc.") or name.startswith("dec.")) or name == "embedding":
        mod.register_forward_pre_hook(pre(name)); mod.register_forward_hook(post(name))
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        out = model(batch)
    torch.cuda.synchronize()
    print(f"== forward {it}: {1e3*(time.perf_counter()-t0):.2f} ms", flush=True)
