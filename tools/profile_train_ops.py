"""Which torch (aten) operators a training step still issues, with the Python line that called them.
usage: python tools/profile_train_ops.py"""
import os
import sys
from collections import Counter

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pointcept-keypointdetection_amd")]
import ptv3_scenes as S  # noqa: E402
from ptv3_hip.configs import FORK_CFG  # noqa: E402
from ptv3_hip.optim import FusedAdamW  # noqa: E402
from pointcept.models import build_model  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = build_model(dict(type="OffsetKeypointPTv3", num_keypoints=6, hidden_dim=256,
                             backbone_conf=dict(type="PT-v3m1", **FORK_CFG))).to(dev).train()
    model.backbone.compute_dtype = torch.bfloat16
    opt = FusedAdamW(model.parameters(), lr=2e-3, weight_decay=5e-3, shadow_dtype=torch.bfloat16)
    batch = {k: v.to(dev) for k, v in S.collate([S.make_scene(100000, 4, None, seed=0)], with_target=6).items()}

    def step():
        opt.zero_grad()
        out = model(batch)
        out["loss"].backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    cfg = None
    try:
        cfg = torch._C._profiler._ExperimentalConfig(verbose=True)
    except Exception:
        pass
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU], with_stack=True,
                                experimental_config=cfg) as prof:
        step()
    torch.cuda.synchronize()
    launching = ("aten::fill_", "aten::zero_", "aten::mul", "aten::mul_", "aten::add", "aten::add_", "aten::copy_",
                 "aten::div", "aten::div_", "aten::sub", "aten::neg", "aten::rsqrt", "aten::sqrt", "aten::index",
                 "aten::cat", "aten::where", "aten::lt", "aten::sum", "aten::mean", "aten::clone", "aten::contiguous",
                 "aten::_to_copy", "aten::addcmul", "aten::rand", "aten::uniform_", "aten::bernoulli_", "aten::sigmoid",
                 "aten::abs", "aten::max", "aten::min", "aten::cumsum", "aten::repeat_interleave", "aten::zeros")
    by_site = Counter()
    for ev in prof.events():
        if ev.name in launching:
            site = next((f for f in ev.stack if "ptv3_hip/" in f or "pointcept/" in f or "bench.py" in f), None)
            if site is None:   # an autograd node without Python frames: name the node that issued it
                par = ev.cpu_parent
                while par is not None and par.cpu_parent is not None and not par.name.startswith(("autograd::", "torch::autograd", "Optimizer")) \
                        and "Backward" not in par.name:
                    par = par.cpu_parent
                site = "<" + (par.name if par is not None else "?") + ">"
            by_site[(ev.name, site.split("pointcept-keypointdetection_amd/")[-1][:110])] += 1
    for (name, site), n in by_site.most_common(70):
        print(f"{n:5d}  {name:22s} {site}")


if __name__ == "__main__":
    main()
