"""Window attention alone at the shapes of the fork config's levels, over (waves per workgroup, QT) launch
configurations (PTV3_ATTN_WAVES / PTV3_ATTN_QT force one; unset = the library's cost model).
usage: python tools/bench_attn.py            -> one line per (shape, config), times in us"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pointcept-keypointdetection_amd")]

SHAPES = [("enc0", 100000, 32, 2), ("dec0", 100000, 64, 4), ("lvl1", 22500, 64, 4), ("lvl2", 6000, 128, 8),
          ("lvl3", 2000, 256, 16), ("lvl4", 250, 512, 32)]
CONFIGS = [(0, 0), (8, 4), (8, 2), (8, 1), (4, 4), (4, 2), (4, 1)]


def main():
    from ptv3_hip import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    for name, n, c, h in SHAPES:
        patch = min(1024, n)
        qkv = torch.randn(n, 3 * c, generator=g).to(dev).to(torch.bfloat16)
        order = torch.randperm(n, generator=g).to(dev)
        inverse = torch.empty_like(order)
        inverse[order] = torch.arange(n, device=dev)
        off = torch.tensor([n], dtype=torch.int64, device=dev)
        wo, wi = ops.window_plan(order[None].contiguous(), inverse[None].contiguous(), off, [n], patch)
        wo, wi = wo[0].contiguous(), wi[0].contiguous()
        ref = None
        res = {}
        for waves, qt in CONFIGS:
            for k, v in (("PTV3_ATTN_WAVES", waves), ("PTV3_ATTN_QT", qt)):
                if v:
                    os.environ[k] = str(v)
                else:
                    os.environ.pop(k, None)
            for _ in range(3):
                out = ops.window_attention(qkv, wo, wi, h, patch, (c // h) ** -0.5)
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            assert torch.equal(out, ref), (name, waves, qt)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20
            e0.record()
            for _ in range(reps):
                ops.window_attention(qkv, wo, wi, h, patch, (c // h) ** -0.5)
            e1.record()
            torch.cuda.synchronize()
            res[f"w{waves}q{qt}"] = round(e0.elapsed_time(e1) / reps * 1e3, 1)
        print(json.dumps({"shape": name, "n": n, "c": c, "heads": h, "us": res}), flush=True)


if __name__ == "__main__":
    main()
