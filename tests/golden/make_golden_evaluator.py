"""tests/golden/evaluator.npz: the reference's OffsetKeypointEvaluator hook (engines/hooks/offset_keypoint_evaluator.py
:19-123) run on seeded batches, in the build container (imports /root/reference in place).

    python tests/golden/make_golden_evaluator.py

The hook file is plain torch; it is imported through ref_loader's bare `pointcept` packages with the reference's own
engines/hooks/builder.py, engines/hooks/default.py (HookBase), utils/comm.py and utils/registry.py.  Two things the CPU
container lacks are shimmed HERE (not in the hook): `Tensor.cuda` is the identity (the hook moves its batches with
.cuda()), and the trainer is a stub object (model = a callable that returns the stored `pred`, val_loader = the list of
batches, logger / writer record what the hook reports).  Stored: the batches and everything the hook reported
(val/MeanDist, val/KP_k_MeanDist, the SaveBest metric, the per-keypoint sample counts from its log lines).
"""
import importlib
import os
import re
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import ref_loader  # noqa: E402


def batch(seed, sizes, K=6, with_scale=True):
    g = torch.Generator().manual_seed(seed)
    n = sum(sizes)
    coord = torch.randn(n, 3, generator=g)
    pred = torch.randn(n, K, 4, generator=g) * 0.2
    pred[..., 3] = torch.rand(n, K, generator=g)
    target = torch.randn(n, K, 4, generator=g) * 0.2
    target[..., 3] = (torch.rand(n, K, generator=g) > 0.9).float()
    target[: sizes[0], 2, 3] = 0          # keypoint 2 has no valid point in the first scene
    d = dict(coord=coord, target=target, offset=torch.tensor(sizes).cumsum(0), _pred=pred)
    if with_scale:
        d["scale"] = torch.rand(len(sizes), generator=g) + 0.5
    return d


def main():
    ref_loader.load()
    # the reference's own hook infrastructure, imported in place (replacing ref_loader's placeholder HookBase)
    hooks_pkg = sys.modules["pointcept.engines.hooks"]
    default = importlib.import_module("pointcept.engines.hooks.default")
    hooks_pkg.HookBase = default.HookBase
    ev = importlib.import_module("pointcept.engines.hooks.offset_keypoint_evaluator")
    torch.Tensor.cuda = lambda self, *a, **k: self      # no GPU in the build container
    out = {}
    runs = [("r0", [batch(1, [400, 600]), batch(2, [50, 700, 300])]),          # two batches, scale given
            ("r1", [batch(3, [1000], with_scale=False)]),                      # no "scale" key: 1.0
            ("r2", [batch(4, [3, 5, 900]), batch(5, [200, 1])])]               # tiny scenes
    for tag, loader in runs:
        logs, scalars = [], {}

        class M:
            def eval(self):
                return self

            def __call__(self, d):
                return {"pred": d["_pred"]}
        hook = ev.OffsetKeypointEvaluator(num_keypoints=6)
        hook.trainer = types.SimpleNamespace(
            val_loader=[dict(b) for b in loader], model=M(), logger=types.SimpleNamespace(info=logs.append),
            writer=types.SimpleNamespace(add_scalar=lambda k, v, e: scalars.__setitem__(k, v)), epoch=0, comm_info={})
        hook.after_epoch()
        counts = [int(re.search(r"Valid Samples Evaluated: (\d+)", s).group(1)) for s in logs if "Valid Samples" in s]
        out[tag + "_nbatch"] = np.array(len(loader))
        for i, b in enumerate(loader):
            for k, v in b.items():
                out[f"{tag}_b{i}_{k.lstrip('_')}"] = v.numpy()
        out[tag + "_mean_dist"] = np.array(scalars["val/MeanDist"], dtype=np.float64)
        out[tag + "_kp_mean_dist"] = np.array([scalars[f"val/KP_{k}_MeanDist"] for k in range(6)], dtype=np.float64)
        out[tag + "_kp_counts"] = np.array(counts)
        out[tag + "_metric"] = np.array(hook.trainer.comm_info["current_metric_value"], dtype=np.float64)
        print(tag, scalars["val/MeanDist"], counts)
    np.savez_compressed(os.path.join(HERE, "evaluator.npz"), **out)
    print("evaluator.npz", os.path.getsize(os.path.join(HERE, "evaluator.npz")))


if __name__ == "__main__":
    main()
