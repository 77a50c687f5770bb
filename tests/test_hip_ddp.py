"""Data-parallel training: two ranks, one scene each, DistributedDataParallel averages the gradients.
Both ranks share the single GPU of the test box, so the process group is gloo (RCCL needs one device per rank; the
8-GPU RCCL run is the driver's `bench.py --mode train --gpus N`).  Checked against the mean of the two
single-process gradients."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _paths():
    for p in (ROOT, os.path.join(ROOT, "pointcept-keypointdetection_amd"), os.path.join(ROOT, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _model():
    _paths()
    from make_golden_cfg import TINY_CFG
    from pointcept.models import build_model
    torch.manual_seed(1234)
    return build_model(dict(type="OffsetKeypointPTv3", num_keypoints=6, hidden_dim=32,
                            backbone_conf=dict(type="PT-v3m1", **dict(TINY_CFG, drop_path=0.0))))


def _scene(rank, dev):
    import ptv3_scenes as S
    return {k: v.to(dev) for k, v in S.make_batch([1100 + 300 * rank], in_channels=4, extent=96, seed=40 + rank,
                                                  with_target=6).items()}


def _grads(model, data):
    torch.manual_seed(5)
    model.zero_grad(set_to_none=True)
    model(data)["loss"].backward()
    return [p.grad.detach().clone() for p in model.parameters()]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    model = _model().to(dev).train()
    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], find_unused_parameters=False)
    torch.manual_seed(5)
    ddp(_scene(rank, dev))["loss"].backward()
    torch.save([p.grad.cpu() for p in model.parameters()], os.path.join(out_dir, f"g{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_two_ranks_average_gradients(tmp_path):
    assert torch.cuda.is_available()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    dev = torch.device("cuda", 0)
    model = _model().to(dev).train()
    ref = [(a + b) / 2 for a, b in zip(_grads(model, _scene(0, dev)), _grads(model, _scene(1, dev)))]
    for r in range(2):
        got = torch.load(os.path.join(str(tmp_path), f"g{r}.pt"), weights_only=True)
        for g, e in zip(got, ref):
            assert (g - e.cpu()).abs().max().item() <= 1e-6 + 1e-5 * e.abs().max().item()


# ---------------------------------------------------------------- sync_bn=True (engines/train.py:256-257)
def _bn_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _paths()
    from pointcept.models.utils.hip_layers import BatchNorm1d, adopt_sync_batchnorm
    from ptv3_hip import ops
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(3)
    seq = torch.nn.Sequential(BatchNorm1d(48, eps=1e-3, momentum=0.01)).to(dev).train()
    with torch.no_grad():
        seq[0].weight.copy_(torch.linspace(0.5, 1.5, 48))
        seq[0].bias.copy_(torch.linspace(-0.5, 0.5, 48))
    seq = torch.nn.SyncBatchNorm.convert_sync_batchnorm(seq)      # what tools/train.py does
    assert isinstance(seq[0], torch.nn.SyncBatchNorm)
    params = list(seq.parameters())
    assert adopt_sync_batchnorm(seq) == 1 and isinstance(seq[0], BatchNorm1d)
    assert all(a is b for a, b in zip(params, seq.parameters()))   # same Parameter objects
    g = torch.Generator().manual_seed(100)
    full = torch.randn(700 + 500, 48, generator=g) * 2 + 1
    w = torch.randn(700 + 500, 48, generator=g)
    lo, hi = (0, 700) if rank == 0 else (700, 1200)                # ragged split of one global batch
    x = full[lo:hi].to(dev).requires_grad_(True)
    y = seq[0](x, act=ops.ACT_GELU)
    (y * w[lo:hi].to(dev)).sum().backward()
    torch.save(dict(y=y.detach().cpu(), dx=x.grad.cpu(), dw=seq[0].weight.grad.cpu(), db=seq[0].bias.grad.cpu(),
                    rm=seq[0].running_mean.cpu(), rv=seq[0].running_var.cpu()), os.path.join(out_dir, f"bn{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_sync_batchnorm_two_ranks_match_one_global_batch(tmp_path):
    """BatchNorm statistics over ALL ranks: two ranks holding 700 / 500 rows of one batch reproduce torch's BatchNorm1d
    + GELU on the whole batch (output, input gradient, running statistics); weight / bias gradients are the local
    sums whose total is the single-process gradient."""
    assert torch.cuda.is_available()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_bn_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    torch.manual_seed(3)
    bn = torch.nn.BatchNorm1d(48, eps=1e-3, momentum=0.01).train()
    with torch.no_grad():
        bn.weight.copy_(torch.linspace(0.5, 1.5, 48))
        bn.bias.copy_(torch.linspace(-0.5, 0.5, 48))
    g = torch.Generator().manual_seed(100)
    full = (torch.randn(1200, 48, generator=g) * 2 + 1).requires_grad_(True)
    w = torch.randn(1200, 48, generator=g)
    y = torch.nn.functional.gelu(bn(full))
    (y * w).sum().backward()
    r = [torch.load(os.path.join(str(tmp_path), f"bn{k}.pt"), weights_only=True) for k in range(2)]
    close = lambda a, b, tol=2e-5: (a - b).abs().max().item() <= tol * (1 + b.abs().max().item())  # noqa: E731
    assert close(torch.cat([r[0]["y"], r[1]["y"]]), y.detach())
    assert close(torch.cat([r[0]["dx"], r[1]["dx"]]), full.grad)
    assert close(r[0]["dw"] + r[1]["dw"], bn.weight.grad, 1e-4) and close(r[0]["db"] + r[1]["db"], bn.bias.grad, 1e-4)
    for k in range(2):
        assert close(r[k]["rm"], bn.running_mean) and close(r[k]["rv"], bn.running_var)


# ---------------------------------------------------------------- DDP bucket views + fused AdamW, two full steps
def _groups(model):
    return [dict(params=[p for n, p in model.named_parameters() if "block" in n], lr=2e-4),
            dict(params=[p for n, p in model.named_parameters() if "block" not in n])]


def _opt_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _paths()
    from ptv3_hip.optim import FusedAdamW
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    model = _model().to(dev).train()
    model.backbone.compute_dtype = torch.bfloat16
    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], find_unused_parameters=False,
                                                    gradient_as_bucket_view=True)      # what bench.py --mode train does
    opt = FusedAdamW(_groups(model), lr=2e-3, weight_decay=5e-3, shadow_dtype=torch.bfloat16)
    data = _scene(rank, dev)
    grads = []
    for it in range(2):
        opt.zero_grad()                                   # set_to_none: the bucket views come back in backward
        torch.manual_seed(5 + it)
        ddp(data)["loss"].backward()
        grads.append([p.grad.detach().float().cpu().clone() for p in model.parameters()])
        opt.step()
    torch.cuda.synchronize()
    torch.save(dict(w=[p.detach().cpu() for p in model.parameters()], g=grads), os.path.join(out_dir, f"o{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_bucket_views_with_fused_adamw_two_steps(tmp_path):
    """The combination `bench.py --mode train --gpus N` runs: DistributedDataParallel(gradient_as_bucket_view=True) +
    FusedAdamW (gradient ADDRESSES as launch arguments, zero_grad(set_to_none=True), two parameter groups, bf16
    shadows), two full steps on two ranks.  Both ranks must end with the same weights, and those weights must be what
    torch.optim.AdamW gives in ONE process that is fed the rank-averaged gradients the ranks saw
    (engines/defaults.py:22-43,136)."""
    assert torch.cuda.is_available()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_opt_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [torch.load(os.path.join(str(tmp_path), f"o{k}.pt"), weights_only=True) for k in range(2)]
    for a, b in zip(r[0]["w"], r[1]["w"]):
        assert torch.equal(a, b)                           # identical replicas after two steps
    for it in range(2):                                   # DDP left the SAME averaged gradient on both ranks
        for a, b in zip(r[0]["g"][it], r[1]["g"][it]):
            assert torch.equal(a, b)
    model = _model().train()                              # CPU replica of the initial weights
    ref_opt = torch.optim.AdamW(_groups(model), lr=2e-3, weight_decay=5e-3)
    for it in range(2):
        for p, g in zip(model.parameters(), r[0]["g"][it]):
            p.grad = g.clone()
        ref_opt.step()
    for p, w in zip(model.parameters(), r[0]["w"]):
        assert (p.detach() - w).abs().max().item() <= 1e-6 + 1e-5 * p.detach().abs().max().item()


def test_bench_train_two_ranks_self_launch(tmp_path):
    """`bench.py --mode train --gpus 2` through its OWN launcher (spawn before any GPU call, tcp rendezvous on
    127.0.0.1, DDP, fused AdamW, max-over-ranks timing), both ranks on the one GPU of the test box over gloo
    (PTV3_BENCH_BACKEND: RCCL needs a device per rank) - the real train path before the first 8-GPU run."""
    import json
    import subprocess
    env = dict(os.environ, PTV3_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "train", "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--points", "20000", "--no-kernel-events"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')][-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["value"] > 0
    assert line["allreduce_probe"]["ranks"] == 2 and line["config"]["parallelism"].startswith("dp2")
    assert line["final_loss"] == line["final_loss"]      # not NaN
