"""Headline benchmark: PTv3 (fork config) + keypoint-offset head forward, Mpoints/s.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

One step = one eval forward of OffsetKeypointPTv3 (configs/my_dataset/offset_keypoint_ptv3.py shape,
46.2 M parameters, 1024-point windows) over one synthetic scene batch already resident in HBM.
Scenes are independent, so N GPUs run N replicas on their own scenes (weak scaling, no data-path
collective); value = total points / max-over-ranks time.
Prints ONE JSON line on rank 0 with `roofline` (dominant matrix-core kernel family, HIP-event timed
inside the timed region) and, at N = 1, `cpu_baseline` (the oracle on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "pointcept-keypointdetection_amd"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK = {"bf16": 2500.0, "fp32": 157.3}  # dense MFMA TFLOP/s, MI355X_MICROARCH.md chip table
PEAK_HBM = 8000.0                       # GB/s, same table (6.3 TB/s is the measured achievable copy rate)


def rank_scene_seeds(rank, scenes_per_rank):
    """Scene shard of one rank: disjoint seeds, `scenes_per_rank` scenes each (weak scaling, no exchange)."""
    return [1000 + rank * scenes_per_rank + i for i in range(scenes_per_rank)]


def reduce_over_ranks(value, device, op):
    """MAX of the timed region / SUM of the points over the job (the only collectives of the bench)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=op)
    return float(t.item())


def pmc_traffic(args, kernel_family):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (tools/measure_round.sh ->
    profiles/<round>/pmc_traffic.json), if they were taken on this workload.  linear and subm_conv are the
    same kernel (gemm_kernel), so the PMC figure covers both families."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_traffic.json")))
    if not files:
        return None, None
    data = json.load(open(files[-1]))
    w = data.get("workload", {})
    if (w.get("points"), w.get("scenes"), w.get("dtype"), w.get("kind")) != (args.points, args.scenes, args.dtype, args.kind):
        return None, None
    key = {"linear": "gemm_kernel<bf16,64ch>", "subm_conv": "gemm_kernel<bf16,64ch>",
           "window_attn": "window_attn_full_kernel"}.get(kernel_family)
    k = data["kernels"].get(key)
    if not k:
        return None, None
    return k["hbm_bytes_per_launch"], {"source": os.path.relpath(files[-1], ROOT), "kernel": key,
                                       "hbm_mb_per_step": round(k["hbm_bytes_per_step"] / 1e6, 1),
                                       "launches_per_step": k["launches_per_step"]}


def pmc_sq():
    """Matrix-core utilisation and stall shares of the two main kernels from the committed SQ counter pass
    (tools/measure_round.sh -> profiles/<round>/pmc_sq.json; default workload only)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_sq.json")))
    if not files:
        return None
    data = json.load(open(files[-1])).get("kernels", {})
    keep = ("mfma_util", "share_issuing", "share_issue_stall", "share_parked")
    out = {k: {f: v[f] for f in keep if f in v} for k, v in data.items()
           if k in ("window_attn_full_kernel", "gemm_kernel<bf16,64ch>")}
    return {"source": os.path.relpath(files[-1], ROOT),
            "definition": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES)", **out} if out else None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points", type=int, default=100000, help="points per scene")
    ap.add_argument("--scenes", type=int, default=1, help="scenes per GPU per step")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--kind", default="surface", choices=["surface", "lidar"])
    ap.add_argument("--cpu-sample", type=int, default=100000, help="points of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-overlap", action="store_true",
                    help="forward mode: do not let consecutive forwards overlap (one feature pipeline in flight)")
    ap.add_argument("--mode", default="forward", choices=["forward", "train"],
                    help="forward = the headline metric (default); train = forward + backward + fused AdamW per "
                         "step, DistributedDataParallel over RCCL when --gpus > 1 (BASELINE configs[3] shape)")
    return ap.parse_args()


def build_model(device):
    from pointcept.models import build_model
    from make_golden_cfg import FORK_CFG
    torch.manual_seed(1234)
    model = build_model(dict(type="OffsetKeypointPTv3", num_keypoints=6,
                             backbone_conf=dict(type="PT-v3m1", **FORK_CFG))).eval()
    gen = torch.Generator().manual_seed(99)
    for n, b in model.named_buffers():  # non-trivial eval BatchNorm statistics
        if n.endswith("running_mean"):
            b.copy_(torch.randn(b.shape, generator=gen) * 0.1)
        if n.endswith("running_var"):
            b.copy_(torch.rand(b.shape, generator=gen) + 0.5)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    return model.to(device), sd, FORK_CFG


def host_cores():
    """CPU threads this process may really use: affinity mask, cgroup quota, capped at the box share."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("PTV3_BENCH_CPU_THREADS", "16"))))


def cpu_baseline(sd, cfg, model, device, n_points):
    """Oracle (CPU restatement of the reference path) on the host cores + parity of the HIP fp32 path."""
    from oracle import ptv3 as O
    import ptv3_scenes as S
    cores = host_cores()
    torch.set_num_threads(cores)
    data = S.make_batch([n_points], in_channels=4, extent=None, seed=7)
    orc = O.OffsetKeypointOracle(cfg, sd)
    torch.manual_seed(11)
    t0 = time.perf_counter()
    with torch.no_grad():
        ref = orc.forward(data)
    dt = time.perf_counter() - t0
    model.backbone.compute_dtype = torch.float32
    torch.manual_seed(11)
    with torch.no_grad():
        out = model({k: v.to(device) for k, v in data.items()})
    pred = out["pred"].float().cpu()
    l2 = (pred[..., :3] - ref["pred"][..., :3]).norm(dim=-1).max().item()
    lg = (pred[..., 3] - ref["pred"][..., 3]).abs().max().item()
    return ({"value": round(n_points / dt / 1e6, 5), "unit": "Mpoints/s", "cores": cores, "kind": "port",
             "sample": f"oracle (torch-CPU fp32 restatement) forward of one {n_points}-point surface scene, "
                       f"{dt:.1f} s wall, {cores} threads"},
            {"offset_l2_max": l2, "mask_prob_abs_max": lg, "tolerance": 1e-4, "sample_points": n_points,
             "mode": "fp32"})


def train_bench(args, model, device, world, rank, local_rank):
    """Train step of the fork config: forward + backward (HIP Functions) + fused AdamW; gradients are averaged by
    DistributedDataParallel over RCCL (one bucketed all-reduce of the 46.2M fp32 gradients, overlapped with
    backward) when world > 1.  One scene per GPU per step = BASELINE configs[3]'s batch-8-on-8-GPUs shape."""
    import ptv3_scenes as S
    from ptv3_hip.optim import FusedAdamW
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    model.backbone.compute_dtype = dtype
    model.train()
    net = model
    if world > 1:
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], find_unused_parameters=False,
                                                        gradient_as_bucket_view=True)
    # the fork's optimizer config: AdamW lr 0.002 wd 0.005, "block" parameters at lr 0.0002
    groups = [dict(params=[p for n, p in model.named_parameters() if "block" in n], lr=2e-4),
              dict(params=[p for n, p in model.named_parameters() if "block" not in n])]
    opt = FusedAdamW(groups, lr=2e-3, weight_decay=5e-3, shadow_dtype=dtype)
    scenes = [S.make_scene(args.points, 4, None, seed, args.kind) for seed in rank_scene_seeds(rank, args.scenes)]
    batch = {k: v.to(device) for k, v in S.collate(scenes, with_target=6).items()}
    n_points = args.points * args.scenes

    def step():
        opt.zero_grad()
        out = net(batch)
        out["loss"].backward()
        opt.step()
        return out["loss"]

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = reduce_over_ranks(elapsed, device, dist.ReduceOp.MAX)
    total_points = reduce_over_ranks(n_points, device, dist.ReduceOp.SUM)
    if rank == 0:
        print(json.dumps({
            "metric": "Mpoints/sec PTv3 train step (fwd+bwd+AdamW) @100k pts/scene, 1024-pt window",
            "value": round(total_points * args.steps / elapsed / 1e6, 4), "unit": "Mpoints/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"OffsetKeypointPTv3 (PT-v3m1 fork config, 46.2M params) train step, "
                                   f"{args.scenes} x {args.points}-point synthetic {args.kind} scene(s) per GPU, "
                                   f"patch 1024, drop_path 0.3, fused AdamW (2 param groups)",
                       "points_per_gpu": n_points,
                       "parallelism": f"dp{world} (DDP over RCCL, 184.7 MB fp32 gradient all-reduce)" if world > 1
                       else "dp1"},
            "roofline": None, "cpu_baseline": None, "final_loss": float(loss.item())}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (the HIP path has no CPU fallback)"
    # one rank per GPU; PTV3_BENCH_BACKEND=gloo lets several ranks rehearse the multi-rank path on ONE GPU (RCCL
    # needs a device per rank) - used only to test this script, never for reported numbers
    backend = os.environ.get("PTV3_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)  # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend=backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    import ptv3_scenes as S
    from ptv3_hip import ops
    model, sd, cfg = build_model(device)

    cpu, parity = None, None
    if args.mode == "train":
        return train_bench(args, model, device, world, rank, dev_index)
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        cpu, parity = cpu_baseline(sd, cfg, model, device, args.cpu_sample)

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    model.backbone.compute_dtype = dtype
    model.backbone.inputs_resident = True  # the scene batch sits in HBM before the timed region (bench contract)
    # throughput mode: two forwards in flight (the small deep levels of step i run under the chip-filling level-0
    # kernels of step i+1); every step is still one complete forward of the scene and all of them finish inside
    # the timed region (synchronize on both sides)
    model.backbone.overlap_calls = not args.no_overlap
    # every rank owns its own scene(s): shard = scene, no exchange on the data path
    scenes = [S.make_scene(args.points, 4, None, seed, args.kind) for seed in rank_scene_seeds(rank, args.scenes)]
    batch = {k: v.to(device) for k, v in S.collate(scenes).items()}
    n_points = args.points * args.scenes

    def step():
        with torch.no_grad():
            return model(batch)["pred"]

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # ---- timed region: K steps, nothing but the forward (barrier + synchronize on both sides)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = reduce_over_ranks(elapsed, device, dist.ReduceOp.MAX)
    total_points = reduce_over_ranks(n_points, device, dist.ReduceOp.SUM)

    # ---- kernel durations: the SAME K steps again with HIP events on the launch stream around every
    # matrix-core launch.  Kept out of the timed region above because the ~340 event records per step are
    # queue markers that cost ~1.5 ms per step (measured: 5.0 ms vs 3.4 ms) - they would falsify `value`.
    if not args.no_kernel_events:
        # one forward in flight here: with two overlapped forwards an event bracket would time the kernel while it
        # shares the chip with the other forward's kernels; the roofline entry describes the kernel alone
        model.backbone.overlap_calls = False
        ops.profile_enable(True)
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()

    # ---- per-kernel-family device time from the events recorded during the timed steps
    roofline = None
    if not args.no_kernel_events:
        fam = ops.profile_collect()
        ops.profile_enable(False)
        dom = max(fam, key=lambda k: fam[k]["ms"])
        d = fam[dom]
        tf = d["flops"] / (d["ms"] * 1e-3) / 1e12
        gbs = d["bytes"] / (d["ms"] * 1e-3) / 1e9
        # which roof binds this family: algorithmic intensity against the machine balance of the dtype
        intensity = d["flops"] / max(d["bytes"], 1.0)
        hbm_bound = intensity * PEAK_HBM * 1e9 < PEAK[args.dtype] * 1e12
        roofline = {"kernel": dom, "measured": f"HIP events on the launch stream, {args.steps} extra steps after the timed region, one forward "
                                "in flight",
                    "bound": "hbm" if hbm_bound else "mfma",
                    "achieved": round(gbs if hbm_bound else tf, 3),
                    "peak": PEAK_HBM if hbm_bound else PEAK[args.dtype],
                    "unit": "GB/s" if hbm_bound else "TFLOP/s",
                    "frac": round((gbs / PEAK_HBM) if hbm_bound else (tf / PEAK[args.dtype]), 5),
                    "traffic": None, "traffic_detail": None,
                    "intensity_flop_per_byte": round(intensity, 1),
                    "avg_launch_us": round(d["ms"] * 1e3 / max(1, d["launches"]), 2),
                    "launches_per_step": d["launches"] // args.steps,
                    "algorithmic_gflop_per_step": round(d["flops"] / args.steps / 1e9, 3),
                    "algorithmic_mb_per_step": round(d["bytes"] / args.steps / 1e6, 3),
                    "families_ms_per_step": {k: round(v["ms"] / args.steps, 3) for k, v in fam.items()},
                    "families_tflops": {k: round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 2) for k, v in fam.items()},
                    "families_gbps": {k: round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1) for k, v in fam.items()}}

    if roofline is not None and rank == 0:
        roofline["traffic"], roofline["traffic_detail"] = pmc_traffic(args, roofline["kernel"])
        if (args.points, args.scenes, args.dtype, args.kind) == (100000, 1, "bf16", "surface"):
            roofline["pmc_sq"] = pmc_sq()
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        line = {
            "metric": "Mpoints/sec PTv3 fwd @100k pts/scene, 1024-pt window; keypoint offset L2 vs ref",
            "value": round(total_points * args.steps / elapsed / 1e6, 4), "unit": "Mpoints/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"OffsetKeypointPTv3 (PT-v3m1 fork config, 46.2M params) eval forward, "
                                   f"{args.scenes} x {args.points}-point synthetic {args.kind} scene(s) per GPU, "
                                   f"patch 1024, serialization + sparse conv + attention + head included",
                       "points_per_gpu": n_points, "parallelism": f"replicas x{world} (scene-sharded, no collective)",
                       "forwards_in_flight": 1 if args.no_overlap else 2},
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
