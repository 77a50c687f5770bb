"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py` into profiles/<round>/pmc_traffic.json.

usage: python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <forwards> <out.json> [workload json]
Units / corrections exactly as /opt/skills/guides/MI355X_MICROARCH.md section HBM prescribes:
  FETCH_SIZE, WRITE_SIZE are in KiB (x1024); on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide (>= 128 B per row)
  coalesced reads -> doubled for the kernels whose reads are such streams (dense GEMM operands, block halves,
  LayerNorm ...).  The gathered launches (sparse-conv instantiations, window attention, neighbour search) read 32-128
  byte row pieces: their FETCH_SIZE is reported BOTH ways (x1 = `fetch_bytes_per_step_raw`, x2 =
  `fetch_bytes_per_step`), the truth lies between.  WRITE_SIZE is taken as is.
"""
import collections, csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def per_kernel(path, counter, nf):
    rows = list(csv.DictReader(open(path)))
    starts = sorted({int(r["Dispatch_Id"]) for r in rows if "sfc_encode" in r["Kernel_Name"]})[-nf:]
    lo = starts[0]
    tot = collections.defaultdict(float)
    cnt = collections.defaultdict(set)
    for r in rows:
        d = int(r["Dispatch_Id"])
        if d < lo or r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        tot[name] += float(r["Counter_Value"])
        cnt[name].add(d)
    return {k: (v / nf, len(cnt[k]) / nf) for k, v in tot.items()}


from kernel_names import label as short, is_gather  # noqa: E402


def main():
    fetch_csv, write_csv, nf, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    workload = json.loads(sys.argv[5]) if len(sys.argv) > 5 else {}
    f = per_kernel(fetch_csv, "FETCH_SIZE", nf)
    w = per_kernel(write_csv, "WRITE_SIZE", nf)
    agg = collections.defaultdict(lambda: {"fetch_bytes_per_step": 0.0, "write_bytes_per_step": 0.0, "launches_per_step": 0.0})
    for name, (v, n) in f.items():
        a = agg[short(name)]
        a["fetch_bytes_per_step"] += v * 1024 * 2   # KiB -> B, gfx950 half-count correction
        if is_gather(short(name)):
            a["fetch_bytes_per_step_raw"] = a.get("fetch_bytes_per_step_raw", 0.0) + v * 1024
        a["launches_per_step"] += n
    for name, (v, n) in w.items():
        agg[short(name)]["write_bytes_per_step"] += v * 1024
    res = {"workload": workload, "forwards_averaged": nf,
           "corrections": "FETCH_SIZE x1024 x2 (gfx950 counts 128-B requests at 64 B), WRITE_SIZE x1024",
           "kernels": {}}
    for k, a in sorted(agg.items(), key=lambda kv: -(kv[1]["fetch_bytes_per_step"] + kv[1]["write_bytes_per_step"])):
        a["hbm_bytes_per_step"] = a["fetch_bytes_per_step"] + a["write_bytes_per_step"]
        a["hbm_bytes_per_launch"] = a["hbm_bytes_per_step"] / max(a["launches_per_step"], 1e-9)
        res["kernels"][k] = {kk: round(vv, 1) for kk, vv in a.items()}
    json.dump(res, open(out, "w"), indent=1)
    for k, a in list(res["kernels"].items())[:8]:
        print(f"{k:32s} {a['hbm_bytes_per_step']/1e6:9.1f} MB/step  {a['launches_per_step']:6.1f} launches  "
              f"{a['hbm_bytes_per_launch']/1e6:8.2f} MB/launch")


if __name__ == "__main__":
    main()
