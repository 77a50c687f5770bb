"""Is the throughput mode paced by the host?  bench.py's forward loop (two forwards in flight) with an extra
busy-wait of D microseconds on the host before every call: if the step time grows by D the host timeline sets
the period, if it does not the GPU does.   usage: python tools/probe_host_delay.py [D ...]   (default 0 100 200 400)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pointcept-keypointdetection_amd")]


def main():
    import bench
    import ptv3_scenes as S
    delays = [int(a) for a in sys.argv[1:]] or [0, 100, 200, 400]
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    model, _, _ = bench.build_model(device)
    model.backbone.compute_dtype = torch.bfloat16
    model.backbone.inputs_resident = True
    model.backbone.overlap_calls = True
    batch = {k: v.to(device) for k, v in S.collate([S.make_scene(100000, 4, None, bench.rank_scene_seeds(0, 1)[0],
                                                                  "surface")]).items()}

    def run(delay_us, steps=30):
        for _ in range(5):
            with torch.no_grad():
                model(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            if delay_us:
                t_end = time.perf_counter() + delay_us * 1e-6
                while time.perf_counter() < t_end:
                    pass
            with torch.no_grad():
                model(batch)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    base = None
    for d in delays:
        ms = run(d)
        base = ms if base is None else base
        print(f"host delay {d:4d} us per call: {ms:.3f} ms per forward ({ms - base:+.3f} ms vs no delay)", flush=True)


if __name__ == "__main__":
    main()
