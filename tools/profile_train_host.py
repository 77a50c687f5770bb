"""Host-side (Python) profile of the training step: cProfile over 4 steps after warm-up, and the wall time per step with
and without waiting for the GPU.  usage: python tools/profile_train_host.py"""
import cProfile
import io
import os
import pstats
import sys
import time

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
    groups = [dict(params=[p for n, p in model.named_parameters() if "block" in n], lr=2e-4),
              dict(params=[p for n, p in model.named_parameters() if "block" not in n])]
    opt = FusedAdamW(groups, lr=2e-3, weight_decay=5e-3, shadow_dtype=torch.bfloat16)
    batch = {k: v.to(dev) for k, v in S.collate([S.make_scene(100000, 4, None, seed=0)], with_target=6).items()}

    def step():
        opt.zero_grad()
        out = model(batch)
        out["loss"].backward()
        opt.step()

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    t_issue = (time.perf_counter() - t0) / 10
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / 10
    print(f"host issue time {t_issue * 1e3:.2f} ms/step, with the GPU drained {t_all * 1e3:.2f} ms/step")
    # host time per phase (no waiting for the GPU inside a phase except the model's own syncs)
    acc = [0.0, 0.0, 0.0, 0.0]
    for _ in range(10):
        t = [time.perf_counter()]
        opt.zero_grad(); t.append(time.perf_counter())
        out = model(batch); t.append(time.perf_counter())
        out["loss"].backward(); t.append(time.perf_counter())
        opt.step(); t.append(time.perf_counter())
        for i in range(4):
            acc[i] += t[i + 1] - t[i]
    torch.cuda.synchronize()
    print("host ms/step: zero_grad %.2f forward %.2f backward %.2f optimizer %.2f" % tuple(a * 100 for a in acc))
    # the same with a drain after every phase: GPU time of each phase when it starts from an empty queue
    acc = [0.0, 0.0, 0.0]
    for _ in range(5):
        opt.zero_grad(); torch.cuda.synchronize()
        t0 = time.perf_counter(); out = model(batch); torch.cuda.synchronize(); t1 = time.perf_counter()
        out["loss"].backward(); torch.cuda.synchronize(); t2 = time.perf_counter()
        opt.step(); torch.cuda.synchronize(); t3 = time.perf_counter()
        acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2
    print("drained ms/step: forward %.2f backward %.2f optimizer %.2f" % tuple(a * 200 for a in acc))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(4):
        step()
    pr.disable()
    torch.cuda.synchronize()
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats("cumtime").print_stats(45)
    print(out.getvalue()[:9000])


if __name__ == "__main__":
    main()
