"""N > 1 path of bench.py on CPU: 2 gloo ranks, scene sharding without exchange, MAX/SUM reductions."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "pointcept-keypointdetection_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    import ptv3_scenes as S
    seeds = bench.rank_scene_seeds(rank, 2)
    scenes = [S.make_scene(500, 4, 32, s) for s in seeds]
    batch = S.collate(scenes)
    n_points = int(batch["offset"][-1])
    elapsed = bench.reduce_over_ranks(1.0 + rank, torch.device("cpu"), dist.ReduceOp.MAX)
    total = bench.reduce_over_ranks(n_points, torch.device("cpu"), dist.ReduceOp.SUM)
    dist.barrier()
    q.put((rank, seeds, n_points, elapsed, total, batch["grid_coord"].sum().item()))
    dist.destroy_process_group()


def test_two_rank_sharding_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, n0, e0, t0, c0), (r1, s1, n1, e1, t1, c1) = res
    assert set(s0).isdisjoint(s1) and len(s0) == len(s1) == 2   # shards are disjoint scene sets
    assert c0 != c1                                              # ... of different data
    assert e0 == e1 == 2.0                                       # MAX over ranks of the timed region
    assert t0 == t1 == n0 + n1 == 2000                           # whole-job points


def test_bench_self_launch_two_gloo_ranks(capfd, monkeypatch):
    """`bench.py --gpus 2` with no launcher in the environment starts its own two ranks (spawned processes, tcp
    rendezvous on 127.0.0.1 - the process model of the reference's engines/launch.py) before any GPU call.
    --rehearse-launch swaps the GPU work for the surrounding machinery: sharding, barrier, MAX / SUM reductions,
    the all-reduce probe; rank 0 prints the one JSON line."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    rc = bench.main(["--gpus", "2", "--rehearse-launch", "--scenes", "2", "--points", "400", "--mode", "train"])
    assert rc == 0
    lines = [ln for ln in capfd.readouterr().out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    assert rec["rehearsal"] is True and rec["value"] is None
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2
    assert rec["elapsed_max"] == 2.0 and rec["total_points"] == 4 * 400
    assert rec["scene_seeds_rank0"] == bench.rank_scene_seeds(0, 2)
    assert rec["allreduce_probe"]["ranks"] == 2 and rec["allreduce_probe"]["busbw_GBps"] > 0
