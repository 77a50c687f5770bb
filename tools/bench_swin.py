"""Time ptv3_swin_attn_fwd (Swin3D cRSE window attention, row A19) on S3DIS-like stages.
usage: python tools/bench_swin.py [n_voxels]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointcept-keypointdetection_amd"))
from ptv3_hip import ops  # noqa: E402


def surface(n, extent, seed):
    rng = np.random.default_rng(seed)
    xy = rng.uniform(0, extent, size=(3 * n, 2))
    z = extent / 2 + 6.0 * np.sin(xy[:, 0] / 19.0) * np.cos(xy[:, 1] / 17.0) + rng.normal(0, 0.4, 3 * n)   # a room-like sheet
    g = np.unique(np.floor(np.concatenate([xy, z[:, None]], 1)).astype(np.int64), axis=0)
    g = g[rng.permutation(len(g))[:n]]
    return np.concatenate([np.zeros((len(g), 1), np.int64), g], 1).astype(np.int32)


def bench_model(n, dev):
    """Whole "Swin3D-v1m1" forward, Swin3D-S / S3DIS config (BASELINE configs[4] shape: 9 input channels, colour +
    normal signals, 5^3 / 7^3 windows), one scene of n points on a room-like sheet, random weights, fp32."""
    from ptv3_hip import configs
    from pointcept.models import build_model
    rng = np.random.default_rng(0)
    extent = int((n / 1.2) ** 0.5)
    g = surface(n, extent, 1)[:, 1:].astype(np.int64)
    n = len(g)
    batch = {"coord": torch.from_numpy(((g + rng.random(g.shape)) * 0.02).astype(np.float32)).to(dev),
             "grid_coord": torch.from_numpy(g).to(dev),
             "feat": torch.from_numpy(rng.normal(size=(n, 9)).astype(np.float32)).to(dev),
             "coord_feat": torch.from_numpy(rng.uniform(-1, 1, (n, 6)).astype(np.float32)).to(dev),
             "offset": torch.tensor([n], device=dev)}
    torch.manual_seed(0)
    model = build_model(configs.SWIN3D_S3DIS_CFG).to(dev).eval()
    if os.environ.get("SWIN_DTYPE") == "bf16":
        model.compute_dtype = torch.bfloat16
    with torch.no_grad():
        for _ in range(2):
            y = model(dict(batch))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            y = model(dict(batch))
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    print(json.dumps({"model": "Swin3D-v1m1 (Swin3D-S, S3DIS config)", "points": n, "dtype": os.environ.get("SWIN_DTYPE", "float32"),
                      "ms_per_forward": round(ms, 2), "Mpoints_per_s": round(n / ms / 1e3, 3),
                      "finite": bool(torch.isfinite(y).all().item())}))


def bench_train(n, dev):
    """Training step (forward + backward + FusedAdamW) of the fork's Swin3D offset model
    (configs/my_dataset/offset_keypoint_swin3d.py: quant 50, XYZ_RGB, 4 levels), one scene of n points, fp32."""
    from ptv3_hip import configs
    from ptv3_hip.optim import FusedAdamW
    from pointcept.models import build_model
    rng = np.random.default_rng(0)
    extent = int((n / 1.2) ** 0.5)
    g = surface(n, extent, 1)[:, 1:].astype(np.int64)
    n = len(g)
    batch = {"coord": torch.from_numpy(((g + rng.random(g.shape)) * 0.02).astype(np.float32)).to(dev),
             "grid_coord": torch.from_numpy(g).to(dev),
             "feat": torch.from_numpy(np.clip(rng.normal(size=(n, 4)) * 0.5, -1, 1).astype(np.float32)).to(dev),
             "offset": torch.tensor([n], device=dev),
             "target": torch.from_numpy(np.concatenate([rng.normal(size=(n, 6, 3)) * 0.3, rng.random((n, 6, 1)) > 0.5],
                                                       -1).astype(np.float32)).to(dev)}
    torch.manual_seed(0)
    model = build_model(configs.OFFSET_SWIN3D_CFG).to(dev).train()
    opt = FusedAdamW(model.parameters(), lr=1e-3, weight_decay=0.01)

    def step():
        opt.zero_grad()
        loss = model(dict(batch))["loss"]
        loss.backward()
        opt.step()
        return loss

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        loss = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    print(json.dumps({"model": "OffsetKeypointSwin3D (fork config, quant 50, XYZ_RGB) train step", "points": n,
                      "dtype": "float32", "ms_per_step": round(ms, 1), "Mpoints_per_s": round(n / ms / 1e3, 3),
                      "loss": round(float(loss), 4)}))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
    dev = torch.device("cuda:0")
    if len(sys.argv) > 2 and sys.argv[2] == "model":
        return bench_model(n, dev)
    if len(sys.argv) > 2 and sys.argv[2] == "train":
        return bench_train(n, dev)
    extent = int((n / 1.2) ** 0.5)
    coords = torch.from_numpy(surface(n, extent, 0)).to(dev)
    n = coords.shape[0]
    g = torch.Generator(device=dev).manual_seed(0)
    for heads, hd, ws, quant, crse, dtype in ((6, 8, 5, 4, "XYZ_RGB_NORM", torch.float32),
                                              (6, 16, 7, 4, "XYZ_RGB_NORM", torch.float32),
                                              (6, 16, 7, 4, "XYZ_RGB_NORM", torch.bfloat16),
                                              (4, 16, 5, 50, "XYZ_RGB", torch.float32)):
        t0 = time.perf_counter()
        w_w_id, w_w_xyz, w_sizes, n2n, inv, w_start = ops.swin_window_mapping(coords, 1, ws, 0)
        torch.cuda.synchronize()
        t_map = (time.perf_counter() - t0) * 1e3
        nsig = {"XYZ": 3, "XYZ_RGB": 6, "XYZ_RGB_NORM": 9}[crse]
        rows = [2 * ws * quant] + [2 * 2 * quant * 2] * (nsig // 3 - 1)
        offs = [r * heads * hd for r in rows for _ in range(3)]
        tabs = [torch.randn(sum(offs), device=dev, generator=g) * 0.02 for _ in range(3)]
        q, k, v = (torch.randn(n, heads, hd, device=dev, generator=g).to(dtype) for _ in range(3))
        sig = torch.rand(n, nsig, device=dev, generator=g)
        sig[:, :3] = w_w_xyz.float() + sig[:, :3]
        sig[:, 3:] = sig[:, 3:] * 2 - 1
        scale = torch.tensor([quant] * 3 + [quant * 2] * (nsig - 3), device=dev, dtype=torch.float32)
        cr = (sig * scale).contiguous()
        mx = int(w_sizes.max().item())      # the model passes the largest window of the mapping (swin3d_layers.py:46)
        run = lambda: ops.swin_attention(q, k, v, *tabs, offs, n2n, w_start, cr, mx)
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        pairs = float((w_sizes.double() ** 2).sum().item()) * heads
        print(json.dumps({"op": "swin_attn_fwd", "voxels": n, "windows": int(w_sizes.numel()),
                          "mean_tokens": round(float(w_sizes.double().mean().item()), 1),
                          "max_tokens": int(w_sizes.max().item()), "heads": heads, "head_dim": hd, "window": ws,
                          "quant": quant, "crse": crse, "dtype": str(dtype).split(".")[-1], "ms": round(ms, 3),
                          "Gpairs_per_s": round(pairs / ms / 1e6, 2), "mapping_ms_cold": round(t_map, 2)}))


if __name__ == "__main__":
    main()
