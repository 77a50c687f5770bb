"""Level-by-level comparison of the "Swin3D-v1m1" forward with oracle/swin3d.py (stem, every stage, every down / upsampling):
prints the relative L2 error and the number of rows off by more than 1e-3 per tap - how the reciprocal-multiply division in
the voxelisation was found.  usage (GPU box): python tools/trace_swin_levels.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pointcept-keypointdetection_amd"))
import importlib.util
spec = importlib.util.spec_from_file_location("t", os.path.join(ROOT, "tests/test_hip_swin3d.py")); t = importlib.util.module_from_spec(spec); spec.loader.exec_module(t)
from ptv3_hip import configs
from pointcept.models import build_model
cfg = dict(configs.TINY_SWIN3D_CFG)
model = build_model(cfg); t._randomise(model, 5)
batch = t._swin_batch([2600, 1500], seed=3)
tr = {}
o = t.O.Swin3DOracle({k: v.numpy() for k, v in model.state_dict().items()}, cfg)
want = o.forward(batch, tr)
dev = torch.device("cuda:0")
model = model.to(dev).eval()
got = {}
model.stem_layer.register_forward_hook(lambda m, i, out: got.__setitem__("stem", out.feat.cpu().numpy()))
for i, l in enumerate(model.layers):
    def hk(m, inp, out, i=i):
        got[f"layer{i}"] = out[0].feat.cpu().numpy()
        if m.downsample is not None:
            got[f"down{i}"] = out[1].feat.cpu().numpy(); got[f"down{i}_cfeat"] = out[1].cfeat.cpu().numpy()
    l.register_forward_hook(hk)
for j, u in enumerate(model.upsamples):
    u.register_forward_hook(lambda m, i, out, j=j: got.__setitem__(f"up{j}", out.feat.cpu().numpy()))
with torch.no_grad():
    y = model(t._to_dev(batch, dev)).cpu().numpy()
for k in tr:
    a, b = got[k], tr[k]
    print(k, a.shape, b.shape, "rel", t._rel(a, b) if a.shape == b.shape else None, "rows>1e-3:", int((np.abs(a-b).max(1) > 1e-3*np.abs(b).max()).sum()) if a.shape==b.shape else None)
print("final", t._rel(y, want))
