"""Per-kernel sums of rocprofv3 --pmc counters (counter_collection.csv), last `nf` forwards.
usage: python tools/summarize_pmc.py <counter_collection.csv> <forwards> [name-filter]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
nf = int(sys.argv[2]); flt = sys.argv[3] if len(sys.argv) > 3 else ""
# dispatch ids increase with launch order
disp = sorted({int(r["Dispatch_Id"]) for r in rows})
starts = sorted({int(r["Dispatch_Id"]) for r in rows if "sfc_encode" in r["Kernel_Name"]})[-nf:]
lo = starts[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for r in rows:
    d = int(r["Dispatch_Id"])
    if d < lo: continue
    name = r["Kernel_Name"].replace("void ", "").replace("ptv3::", "").split("(")[0][:48]
    if flt and flt not in name: continue
    agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
    if (d, name) not in seen:
        seen.add((d, name)); cnt[name] += 1
for name, c in sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:14]:
    print(f"{name:48s} n/fwd={cnt[name]/nf:6.1f} " + " ".join(f"{k}={v/nf:.4g}" for k, v in sorted(c.items())))
