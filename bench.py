"""Headline benchmark: PTv3 (fork config) + keypoint-offset head forward, Mpoints/s.

    python bench.py --gpus N --steps K --warmup W

One step = one eval forward of OffsetKeypointPTv3 (configs/my_dataset/offset_keypoint_ptv3.py shape,
46.2 M parameters, 1024-point windows) over one synthetic scene batch already resident in HBM.
Scenes are independent, so N GPUs run N replicas on their own scenes (weak scaling, no data-path
collective); value = total points / max-over-ranks time.

Launch: under `torch.distributed.run` (RANK / LOCAL_RANK / WORLD_SIZE in the environment) every process is one
rank.  Without them, `--gpus N` with N > 1 starts N child processes ITSELF before anything touches the GPU - one
process per GPU, `init_process_group("nccl")` over tcp://127.0.0.1:<free port>, the process model of the
reference's pointcept/engines/launch.py:36-113 - and returns their exit code.

Prints ONE JSON line on rank 0 with
  value / ms_per_step     SURVEY 8(d)'s figure: K single forwards, each followed by torch.cuda.synchronize(), timed as
                          one region between barrier + synchronize (one forward in flight at any time)
  latency_ms_median       the median of those K single forwards (and latency_mpoints_per_s)
  throughput_2_in_flight  the same K forwards WITHOUT the per-step synchronize, two in flight (the executor's
                          throughput mode); --no-overlap: one in flight, back to back
  fp32                    the same figures in fp32 (exact-fp32 MFMA), the arithmetic of the 1e-4 parity bar
  parity                  fp32 AND bf16 outputs of the HIP path vs the oracle on the cpu_baseline scene
  roofline                dominant KERNEL, HIP-event timed on the launch stream in this run (traffic: null - the PMC
                          passes are separate runs; their committed summaries are quoted under pmc_replayed)
  cpu_baseline            the oracle on the host cores (N = 1 only, bounded sample: median of 5 forwards)
  other_workloads         BASELINE configs[2] (PTv3 semseg on a 120k-point LiDAR-like scan, with its own roofline),
                          configs[4] (Swin3D-S), configs[3]'s per-GPU share (one training step, with its roofline)
"""
import argparse
import json
import os
import socket
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "pointcept-keypointdetection_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK = {"bf16": 2500.0, "fp32": 157.3}  # dense MFMA TFLOP/s, MI355X_MICROARCH.md chip table
PEAK_HBM = 8000.0                       # GB/s, same table (6.3 TB/s is the measured achievable copy rate)
GRAD_BYTES = 46181592 * 4               # fp32 gradient of OffsetKeypointPTv3 (backbone 46,158,272 + head 23,320)


def rank_scene_seeds(rank, scenes_per_rank):
    """Scene shard of one rank: disjoint seeds, `scenes_per_rank` scenes each (weak scaling, no exchange)."""
    return [1000 + rank * scenes_per_rank + i for i in range(scenes_per_rank)]


def reduce_over_ranks(value, device, op):
    """MAX of the timed region / SUM of the points over the job (the only collectives of the forward bench)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=op)
    return float(t.item())


def allreduce_probe(device, nbytes, reps=5):
    """Timed all-reduce of one fp32 buffer the size of the model's gradient: the collective a DDP
    step issues.  bus GB/s = 2 (n-1)/n x bytes / time (the rccl-tests convention)."""
    world = dist.get_world_size()
    buf = torch.ones(nbytes // 4, dtype=torch.float32, device=device)
    sync = torch.cuda.synchronize if device.type == "cuda" else (lambda: None)
    dist.all_reduce(buf)
    sync()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        dist.all_reduce(buf)
    sync()
    dt = (time.perf_counter() - t0) / reps
    dt = reduce_over_ranks(dt, device, dist.ReduceOp.MAX)
    return {"bytes": nbytes, "ms": round(dt * 1e3, 3), "algbw_GBps": round(nbytes / dt / 1e9, 2),
            "busbw_GBps": round(2 * (world - 1) / world * nbytes / dt / 1e9, 2), "ranks": world}


def _pmc_label(kernel, dtype):
    """bench.py's kernel name -> the label tools/kernel_names.py gives the same kernel in the PMC / trace summaries"""
    dt = "bf16" if dtype == "bf16" else "f32"
    for tag in ("64ch", "32ch"):
        kernel = kernel.replace(f"gemm_kernel<{tag}>", f"gemm_kernel<{dt},{tag}>")
    return kernel.replace("gemm_big_kernel ", f"gemm_big_kernel<{dt},128ch> ")


def pmc_replayed(tag, kernel):
    """Counter summaries of the committed rocprofv3 --pmc passes over the SAME command (tools/measure_round.sh ->
    profiles/<round>/pmc_traffic*.json, pmc_sq*.json; tag = "default" | "lidar").  They were NOT measured in this run:
    the block says where it was replayed from and sits outside `roofline`."""
    import glob
    suffix = "" if tag == "default" else "_" + tag
    out = {}
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", f"pmc_traffic{suffix}.json")))
    if files and kernel:
        data = json.load(open(files[-1]))
        want = _pmc_label(kernel, "bf16")
        hits = [v for k, v in data.get("kernels", {}).items()
                if k.replace(",pd4", "") == want or (want.startswith("window_attn_full_kernel") and k.startswith(want))
                or k.startswith(want.split(" (")[0])]
        if hits:
            step = sum(h["hbm_bytes_per_step"] for h in hits)
            launches = sum(h["launches_per_step"] for h in hits)
            out["traffic"] = {"replayed_from": os.path.relpath(files[-1], ROOT), "kernel": want,
                              "hbm_mb_per_launch": round(step / max(launches, 1e-9) / 1e6, 3),
                              "hbm_mb_per_step": round(step / 1e6, 1), "launches_per_step": launches}
            raw = [h.get("fetch_bytes_per_step_raw") for h in hits]
            if all(r is not None for r in raw):   # gathered reads: FETCH_SIZE x1 .. x2 (see tools/pmc_traffic.py)
                out["traffic"]["hbm_mb_per_step_low"] = round((step - sum(raw)) / 1e6, 1)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", f"pmc_sq{suffix}.json")))
    if files:
        data = json.load(open(files[-1])).get("kernels", {})
        keep = ("mfma_util", "share_issuing", "share_issue_stall", "share_parked")
        sq = {k: {f: v[f] for f in keep if f in v} for k, v in data.items()
              if k.startswith(("window_attn_full_kernel", "gemm_kernel<bf16,64ch", "gemm_big_kernel", "block_", "conv_tile"))}
        if sq:
            out["sq"] = {"replayed_from": os.path.relpath(files[-1], ROOT),
                         "definition": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES)", **sq}
    return out or None


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points", type=int, default=100000, help="points per scene")
    ap.add_argument("--scenes", type=int, default=1, help="scenes per GPU per step")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--kind", default="surface", choices=["surface", "lidar"])
    ap.add_argument("--model", default="offset", choices=["offset", "semseg"],
                    help="offset = OffsetKeypointPTv3 over the fork config (headline); semseg = DefaultSegmentorV2 "
                         "(19 classes) over the upstream PTv3 semseg backbone (enable_flash=True), BASELINE configs[2]")
    ap.add_argument("--cpu-sample", type=int, default=100000, help="points of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the fp32 line and the configs[2] workload")
    ap.add_argument("--no-overlap", action="store_true",
                    help="forward mode: do not let consecutive forwards overlap (one feature pipeline in flight)")
    ap.add_argument("--mode", default="forward", choices=["forward", "train"],
                    help="forward = the headline metric (default); train = forward + backward + fused AdamW per "
                         "step, DistributedDataParallel over RCCL when --gpus > 1 (BASELINE configs[3] shape)")
    ap.add_argument("--rehearse-launch", action="store_true",
                    help="no GPU work: exercise launch, rendezvous (gloo), sharding and reductions only; the JSON "
                         "line then carries value null and rehearsal true (never a reported number)")
    return ap.parse_args(argv)


# --------------------------------------------------------------------------------------------------
# launch: one process per GPU (pointcept/engines/launch.py:36-113)
# --------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_entry(local_rank, world, port, argv):
    os.environ.update(RANK=str(local_rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    run(parse(argv))


def self_launch(args, argv):
    """Start args.gpus fresh processes (spawn: new interpreters, nothing inherited from a GPU context - this parent
    never initialises the GPU) and wait for them.  Rank 0 prints the JSON line on the shared stdout."""
    import torch.multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_rank_entry, args=(r, args.gpus, port, argv)) for r in range(args.gpus)]
    for p in procs:
        p.start()
    # poll: a rank that dies early (e.g. during init) leaves the others inside init_process_group or a collective until
    # the communicator's timeout; on the first non-zero exit the remaining ranks are terminated
    code = 0
    while any(p.is_alive() for p in procs) and not code:
        for p in procs:
            p.join(timeout=0.2)
            if p.exitcode not in (None, 0):
                code = p.exitcode
                break
    if code:
        for p in procs:
            if p.is_alive():
                p.terminate()
        for p in procs:
            p.join(timeout=10)
        return code
    return max((p.exitcode or 0) for p in procs)


# --------------------------------------------------------------------------------------------------
# models and scenes
# --------------------------------------------------------------------------------------------------
def perturb_bn(model):
    gen = torch.Generator().manual_seed(99)
    for n, b in model.named_buffers():  # non-trivial eval BatchNorm statistics
        if n.endswith("running_mean"):
            b.copy_(torch.randn(b.shape, generator=gen) * 0.1)
        if n.endswith("running_var"):
            b.copy_(torch.rand(b.shape, generator=gen) + 0.5)


def build_model(device, kind="offset"):
    from pointcept.models import build_model
    from ptv3_hip.configs import FORK_CFG, SEMSEG_CFG
    torch.manual_seed(1234)
    if kind == "semseg":
        cfg = SEMSEG_CFG
        model = build_model(dict(type="DefaultSegmentorV2", num_classes=19, backbone_out_channels=64,
                                 backbone=dict(type="PT-v3m1", **cfg))).eval()
    else:
        cfg = FORK_CFG
        model = build_model(dict(type="OffsetKeypointPTv3", num_keypoints=6,
                                 backbone_conf=dict(type="PT-v3m1", **cfg))).eval()
    perturb_bn(model)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    return model.to(device), sd, cfg


def make_scene(points, kind, seed):
    """surface: dense synthetic surface scan (pools ~4x per stage); lidar: 64-ring scan at its natural 0.05 m voxel
    size in a 2048^3 grid (serialization depth 11, pools ~1.3x per stage) - the SemanticKITTI-like shape."""
    import ptv3_scenes as S
    return S.make_scene(points, 4, None if kind == "surface" else 2048, seed, kind)


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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(sd, cfg, model, device, n_points):
    """Oracle (CPU restatement of the reference path) on the host cores + parity of the HIP path - in fp32 (the
    1e-4 bar) AND in bf16 (the arithmetic `value` is measured in) - on the same scene."""
    from oracle import ptv3 as O
    import ptv3_scenes as S
    cores = host_cores()
    torch.set_num_threads(cores)
    data = S.make_batch([n_points], in_channels=4, extent=None, seed=7)
    orc = O.OffsetKeypointOracle(cfg, sd)
    times = []
    for _ in range(int(os.environ.get("PTV3_BENCH_CPU_REPS", "5"))):   # median of 5
        torch.manual_seed(11)
        t0 = time.perf_counter()
        with torch.no_grad():
            ref = orc.forward(data)
        times.append(time.perf_counter() - t0)
    dt = statistics.median(times)
    parity = {"tolerance_fp32": 1e-4, "sample_points": n_points,
              "reference": "oracle (torch-CPU fp32 restatement, pinned to the reference by tests/golden)",
              "logit_scale": round(ref["logits"].abs().max().item(), 4)}
    datad = {k: v.to(device) for k, v in data.items()}
    for name, dt_ in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        model.backbone.compute_dtype = dt_
        torch.manual_seed(11)
        with torch.no_grad():
            out = model(datad)
        pred = out["pred"].float().cpu()
        d3 = (pred[..., :3] - ref["pred"][..., :3]).norm(dim=-1)
        dm = (pred[..., 3] - ref["pred"][..., 3]).abs()
        parity[name] = {"offset_l2_max": d3.max().item(), "offset_l2_mean": d3.mean().item(),
                        "mask_prob_abs_max": dm.max().item(), "mask_prob_abs_mean": dm.mean().item()}
    return ({"value": round(n_points / dt / 1e6, 5), "unit": "Mpoints/s", "cores": cores, "kind": "port",
             "cpu_model": cpu_model(), "threads": cores,
             "sample": f"oracle (torch-CPU fp32 restatement) forward of one {n_points}-point surface scene: median of "
                       f"{len(times)} runs = {dt:.2f} s wall (min {min(times):.2f}, max {max(times):.2f}), {cores} threads"},
            parity)


# --------------------------------------------------------------------------------------------------
# timing helpers
# --------------------------------------------------------------------------------------------------
def timed_steps(step, steps, warmup, world, device):
    """W untimed steps, then EXACTLY K steps bracketed by barrier + synchronize on both sides; MAX over ranks."""
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    return reduce_over_ranks(time.perf_counter() - t0, device, dist.ReduceOp.MAX)


def synced_steps(step, steps, warmup, world, device):
    """W untimed forwards, then K single forwards with torch.cuda.synchronize() after EACH, the K of them
    timed as one region between barrier + synchronize; (elapsed MAX over ranks, median single-forward ms)."""
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ts = []
    t0 = time.perf_counter()
    for _ in range(steps):
        t1 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t1) * 1e3)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    return reduce_over_ranks(time.perf_counter() - t0, device, dist.ReduceOp.MAX), statistics.median(ts)


def forward_figures(model, batch, key, args, world, device, overlap):
    """elapsed of K synchronised single forwards, their median (ms), elapsed of K forwards in throughput mode, step fn"""
    def step():
        with torch.no_grad():
            return model(batch)[key]
    model.backbone.overlap_calls = False
    elapsed, lat = synced_steps(step, args.steps, args.warmup, world, device)
    model.backbone.overlap_calls = overlap
    thr = timed_steps(step, args.steps, args.warmup, world, device)
    model.backbone.overlap_calls = False
    step()
    torch.cuda.synchronize()
    return elapsed, lat, thr, step


def kernel_roofline(step, steps, dtype_name, rank=0):
    """Per-KERNEL durations of `steps` more steps with HIP events on the launch stream around every bracketed launch
    (ptv3_profile_enable; one forward in flight, so a bracket times its kernel alone), and the roofline entry of the
    dominant kernel: its algorithmic bytes / flops per launch over its average launch duration - the quantity a
    rocprofv3 --kernel-trace --stats summary of this command gives for the same kernel name.  Kept out of the timed
    region: the ~340 event records per step are queue markers that cost ~1.5 ms per step."""
    from ptv3_hip import ops
    ops.profile_enable(True)
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    kern = ops.profile_collect_kernels()   # per KERNEL (one launch per bracket); before the family collect
    fam = ops.profile_collect()
    ops.profile_enable(False)
    if not kern:
        return None
    dom = max(kern, key=lambda k: kern[k]["ms"])
    d = kern[dom]
    tf = d["flops"] / (d["ms"] * 1e-3) / 1e12
    gbs = d["bytes"] / (d["ms"] * 1e-3) / 1e9
    # which roof binds it: algorithmic intensity against the machine balance of the dtype
    intensity = d["flops"] / max(d["bytes"], 1.0)
    hbm_bound = intensity * PEAK_HBM * 1e9 < PEAK[dtype_name] * 1e12

    def per_kernel(v):
        n = max(1, v["launches"])
        return {"ms_per_step": round(v["ms"] / steps, 4), "launches_per_step": round(v["launches"] / steps, 1),
                "avg_launch_us": round(v["ms"] * 1e3 / n, 2),
                "algorithmic_mb_per_launch": round(v["bytes"] / n / 1e6, 4),
                "algorithmic_gflop_per_launch": round(v["flops"] / n / 1e9, 4),
                "gbps": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1),
                "tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 2)}
    fam = {k: v for k, v in fam.items() if v["launches"]}
    return {"kernel": dom, "measured": f"HIP events on the launch stream around every launch of the kernel, {steps} extra "
                                       "steps after the timed region, one step in flight",
            "bound": "hbm" if hbm_bound else "mfma",
            "achieved": round(gbs if hbm_bound else tf, 3),
            "peak": PEAK_HBM if hbm_bound else PEAK[dtype_name],
            "unit": "GB/s" if hbm_bound else "TFLOP/s",
            "frac": round((gbs / PEAK_HBM) if hbm_bound else (tf / PEAK[dtype_name]), 5),
            "traffic": None,
            "intensity_flop_per_byte": round(intensity, 1),
            "avg_launch_us": round(d["ms"] * 1e3 / max(1, d["launches"]), 2),
            "launches_per_step": round(d["launches"] / steps, 1),
            "algorithmic_gflop_per_step": round(d["flops"] / steps / 1e9, 3),
            "algorithmic_mb_per_step": round(d["bytes"] / steps / 1e6, 3),
            "families_ms_per_step": {k: round(v["ms"] / steps, 3) for k, v in fam.items()},
            "families_tflops": {k: round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 2) for k, v in fam.items()},
            "families_gbps": {k: round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1) for k, v in fam.items()},
            "kernels": {k: per_kernel(v) for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["ms"])}}


def semseg_lidar_workload(device, args, world):
    """BASELINE configs[2] beside the headline: PTv3 semseg forward on one 120k-point LiDAR-like scan."""
    import ptv3_scenes as S
    model, _, _ = build_model(device, "semseg")
    model.backbone.compute_dtype = torch.bfloat16
    model.backbone.inputs_resident = True
    batch = {k: v.to(device) for k, v in S.collate([make_scene(120000, "lidar", 1000)]).items()}
    elapsed, lat, thr, step = forward_figures(model, batch, "seg_logits", args, world, device, True)
    roof = kernel_roofline(step, args.steps, "bf16")
    with torch.no_grad():
        pts = model(batch, return_point=True)["point"]["_stage_points"]
    torch.cuda.synchronize()
    return {"workload": "DefaultSegmentorV2 (19 classes) over PT-v3m1 (fork widths, enable_flash=True, patch 1024), "
                        "1 x 120000-point LiDAR-like scan, 2048^3 grid (depth 11), bf16",
            "value": round(120000 * args.steps / elapsed / 1e6, 4), "unit": "Mpoints/s",
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "latency_ms_median": round(lat, 3),
            "latency_mpoints_per_s": round(120000 / lat / 1e3, 4),
            "throughput_2_in_flight": {"value": round(120000 * args.steps / thr / 1e6, 4), "unit": "Mpoints/s",
                                       "ms_per_step": round(thr / args.steps * 1e3, 3)},
            "stage_points": pts, "roofline": roof,
            "pmc_replayed": pmc_replayed("lidar", roof["kernel"] if roof else None)}


def train_step_workload(device, args):
    """BASELINE configs[3]'s per-GPU share beside the headline: one training step (forward + backward + fused AdamW) of the
    fork config on one 100k-point scene, bf16 activations, fp32 masters, drop_path 0.3 - what `--mode train` times."""
    import ptv3_scenes as S
    from ptv3_hip.optim import FusedAdamW
    model, _, _ = build_model(device, "offset")
    model.backbone.compute_dtype = torch.bfloat16
    model.train()
    groups = [dict(params=[p for n, p in model.named_parameters() if "block" in n], lr=2e-4),
              dict(params=[p for n, p in model.named_parameters() if "block" not in n])]
    opt = FusedAdamW(groups, lr=2e-3, weight_decay=5e-3, shadow_dtype=torch.bfloat16)
    batch = {k: v.to(device) for k, v in S.collate([make_scene(100000, "surface", 1000)], with_target=6).items()}
    last = {}

    def step():
        opt.zero_grad()
        out = model(batch)
        out["loss"].backward()
        opt.step()
        last["loss"] = out["loss"]

    steps = 10
    elapsed = timed_steps(step, steps, 3, 1, device)
    roof = kernel_roofline(step, steps, "bf16")
    return {"workload": "OffsetKeypointPTv3 (fork config, 46.2M params) train step: forward + backward + fused AdamW, "
                        "1 x 100000-point scene, bf16 activations, drop_path 0.3", "value": round(1e5 * steps / elapsed / 1e6, 4),
            "unit": "Mpoints/s", "ms_per_step": round(elapsed / steps * 1e3, 3), "final_loss": float(last["loss"].item()),
            "roofline": roof}


def swin3d_workload(device, points=1000000):
    """BASELINE configs[4] beside the headline: "Swin3D-v1m1" (Swin3D-S, the S3DIS config: 9 input channels, colour + normal
    signals, 5^3 / 7^3-voxel windows with cRSE tables) forward on one ~1M-point room-like scene (an S3DIS Area-5 room at
    the 2 cm base grid).  fp32, random weights; PARITY UNPINNED (MinkowskiEngine / microsoft/Swin3D are not in the
    reference tree - DESIGN.md section 10)."""
    import numpy as np
    from ptv3_hip import configs
    from pointcept.models import build_model as build
    rng = np.random.default_rng(0)
    extent = int((points / 1.2) ** 0.5)
    xy = rng.uniform(0, extent, size=(3 * points, 2))
    z = extent / 2 + 6.0 * np.sin(xy[:, 0] / 19.0) * np.cos(xy[:, 1] / 17.0) + rng.normal(0, 0.4, 3 * points)
    g = np.unique(np.floor(np.concatenate([xy, z[:, None]], 1)).astype(np.int64), axis=0)
    g = g[rng.permutation(len(g))[:points]]
    n = len(g)
    batch = {"coord": torch.from_numpy(((g + rng.random(g.shape)) * 0.02).astype(np.float32)).to(device),
             "grid_coord": torch.from_numpy(g).to(device),
             "feat": torch.from_numpy(rng.normal(size=(n, 9)).astype(np.float32)).to(device),
             "coord_feat": torch.from_numpy(rng.uniform(-1, 1, (n, 6)).astype(np.float32)).to(device),
             "offset": torch.tensor([n], device=device)}
    torch.manual_seed(0)
    model = build(configs.SWIN3D_S3DIS_CFG).to(device).eval()
    out = {}

    def step():
        with torch.no_grad():
            out["y"] = model(dict(batch))
    reps = 5
    ts = []
    for i in range(2 + reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        if i >= 2:
            ts.append((time.perf_counter() - t0) * 1e3)
    ms = statistics.median(ts)
    roof = kernel_roofline(step, 3, "fp32")
    finite = bool(torch.isfinite(out["y"]).all().item())
    # the reference's S3DIS configs set enable_amp = True: the same forward with bf16 features / GEMM operands / q, k, v
    model.compute_dtype = torch.bfloat16
    tb = []
    for i in range(2 + reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        if i >= 2:
            tb.append((time.perf_counter() - t0) * 1e3)
    mb = statistics.median(tb)
    return {"workload": f"Swin3D-v1m1 (Swin3D-S, S3DIS config, 28.2M params) eval forward, 1 x {n}-point room-like scene, "
                        "fp32, random weights; parity unpinned", "value": round(n / ms / 1e3, 4), "unit": "Mpoints/s",
            "ms_per_step": round(ms, 2), "latency_ms_median": round(ms, 2), "finite": finite,
            "amp_bf16": {"value": round(n / mb / 1e3, 4), "unit": "Mpoints/s", "ms_per_step": round(mb, 2),
                         "finite": bool(torch.isfinite(out["y"]).all().item()),
                         "note": "model.compute_dtype = torch.bfloat16 (the reference config runs under enable_amp)"},
            "roofline": roof}


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
    scenes = [make_scene(args.points, args.kind, seed) for seed in rank_scene_seeds(rank, args.scenes)]
    batch = {k: v.to(device) for k, v in S.collate(scenes, with_target=6).items()}
    n_points = args.points * args.scenes
    last = {}

    def step():
        opt.zero_grad()
        out = net(batch)
        out["loss"].backward()
        opt.step()
        last["loss"] = out["loss"]

    elapsed = timed_steps(step, args.steps, args.warmup, world, device)
    total_points = reduce_over_ranks(n_points, device, dist.ReduceOp.SUM)
    probe = allreduce_probe(device, GRAD_BYTES) if world > 1 else None
    roof = kernel_roofline(step, args.steps, args.dtype, rank) if not args.no_kernel_events else None
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
                       "parallelism": f"dp{world} (DDP over RCCL, {GRAD_BYTES / 1e6:.1f} MB fp32 gradient all-reduce)"
                       if world > 1 else "dp1"},
            "rccl_ranks": world, "allreduce_probe": probe,
            "roofline": roof, "cpu_baseline": None, "final_loss": float(last["loss"].item())}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def rehearse(args, world, rank):
    """--rehearse-launch: everything around the GPU work (used by the CPU tests of the N > 1 path)."""
    import ptv3_scenes as S
    device = torch.device("cpu")
    seeds = rank_scene_seeds(rank, args.scenes)
    scenes = [S.make_scene(min(args.points, 2000), 4, 64, s) for s in seeds]
    n_points = int(S.collate(scenes)["offset"][-1])
    if world > 1:
        dist.barrier()
    elapsed = reduce_over_ranks(1.0 + rank, device, dist.ReduceOp.MAX)
    total = reduce_over_ranks(n_points, device, dist.ReduceOp.SUM)
    seen = int(reduce_over_ranks(1, device, dist.ReduceOp.SUM))
    probe = allreduce_probe(device, 1 << 20, reps=2) if world > 1 else None
    if rank == 0:
        print(json.dumps({"metric": "launch rehearsal (no GPU work)", "value": None, "rehearsal": True,
                          "n_gpus": world, "rccl_ranks": seen, "elapsed_max": elapsed, "total_points": total,
                          "scene_seeds_rank0": seeds, "allreduce_probe": probe, "mode": args.mode}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def run(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.rehearse_launch:
        if world > 1:
            dist.init_process_group(backend="gloo")
        return rehearse(args, world, rank)
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
    seen = int(reduce_over_ranks(1, device, dist.ReduceOp.SUM))   # ranks that really joined the communicator

    import ptv3_scenes as S
    from ptv3_hip import ops
    model, sd, cfg = build_model(device, args.model)
    out_key = "pred" if args.model == "offset" else "seg_logits"

    cpu, parity = None, None
    if args.mode == "train":
        return train_bench(args, model, device, world, rank, dev_index)
    if rank == 0 and world == 1 and args.cpu_sample > 0 and args.model == "offset":
        cpu, parity = cpu_baseline(sd, cfg, model, device, args.cpu_sample)

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    model.backbone.compute_dtype = dtype
    model.backbone.inputs_resident = True  # the scene batch sits in HBM before the timed region (bench contract)
    # every rank owns its own scene(s): shard = scene, no exchange on the data path
    scenes = [make_scene(args.points, args.kind, seed) for seed in rank_scene_seeds(rank, args.scenes)]
    batch = {k: v.to(device) for k, v in S.collate(scenes).items()}
    n_points = args.points * args.scenes

    # ---- timed region: K steps, nothing but the forward (barrier + synchronize on both sides).  Throughput mode:
    # two forwards in flight (the small deep levels of step i run under the chip-filling level-0 kernels of step
    # i+1); every step is still one complete forward of the scene and all of them finish inside the timed region.
    # Then SURVEY 8(d)'s figure: the median of single, synchronize-bracketed forwards (one in flight).
    elapsed, latency, thr, step = forward_figures(model, batch, out_key, args, world, device, not args.no_overlap)
    total_points = reduce_over_ranks(n_points, device, dist.ReduceOp.SUM)

    # ---- kernel durations: the SAME K steps again with HIP events on the launch stream around every matrix-core launch
    roofline = None
    if not args.no_kernel_events:
        model.backbone.overlap_calls = False
        roofline = kernel_roofline(step, args.steps, args.dtype, rank)

    # ---- the same workload in fp32 (the arithmetic of the 1e-4 parity bar), and BASELINE configs[2]
    fp32, extra = None, None
    default_run = world == 1 and not args.no_extra and args.model == "offset"
    if default_run and args.dtype == "bf16":
        model.backbone.compute_dtype = torch.float32
        e32, l32, t32, _ = forward_figures(model, batch, out_key, args, world, device, not args.no_overlap)
        fp32 = {"value": round(n_points * args.steps / e32 / 1e6, 4), "unit": "Mpoints/s",
                "ms_per_step": round(e32 / args.steps * 1e3, 3), "latency_ms_median": round(l32, 3),
                "throughput_2_in_flight": {"value": round(n_points * args.steps / t32 / 1e6, 4),
                                           "ms_per_step": round(t32 / args.steps * 1e3, 3)}}
        model.backbone.compute_dtype = dtype
    if default_run and rank == 0:
        del model, batch
        torch.cuda.empty_cache()
        extra = {"semseg_lidar_120k": semseg_lidar_workload(device, args, world)}
        torch.cuda.empty_cache()
        extra["swin3d_s3dis_1m"] = swin3d_workload(device)
        torch.cuda.empty_cache()
        extra["train_step_100k"] = train_step_workload(device, args)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        name = ("OffsetKeypointPTv3 (PT-v3m1 fork config, 46.2M params)" if args.model == "offset" else
                "DefaultSegmentorV2 (19 classes, PT-v3m1 fork widths, enable_flash=True)")
        line = {
            "metric": "Mpoints/sec PTv3 fwd @100k pts/scene, 1024-pt window; keypoint offset L2 vs ref",
            "value": round(total_points * args.steps / elapsed / 1e6, 4), "unit": "Mpoints/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"{name} eval forward, "
                                   f"{args.scenes} x {args.points}-point synthetic {args.kind} scene(s) per GPU, "
                                   f"patch 1024, serialization + sparse conv + attention + head included",
                       "points_per_gpu": n_points, "parallelism": f"replicas x{world} (scene-sharded, no collective)",
                       "forwards_in_flight": 1,
                       "timing": "K single forwards, torch.cuda.synchronize() after each (SURVEY 8d), one timed region"},
            "latency_ms_median": round(latency, 3),
            "latency_mpoints_per_s": round(n_points / latency / 1e3, 4),
            "throughput_2_in_flight": {"value": round(total_points * args.steps / thr / 1e6, 4), "unit": "Mpoints/s",
                                       "ms_per_step": round(thr / args.steps * 1e3, 3),
                                       "forwards_in_flight": 1 if args.no_overlap else 2},
            "rccl_ranks": seen,
            "fp32": fp32,
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity, "other_workloads": extra,
            "pmc_replayed": pmc_replayed("lidar" if args.kind == "lidar" else "default", roofline["kernel"] if roofline else None)
            if (args.points, args.scenes, args.dtype) in ((100000, 1, "bf16"), (120000, 1, "bf16")) else None,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no external launcher: become one.  Nothing above this line touches the GPU.
        return self_launch(args, argv)
    run(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
