"""Per-queue timeline of a rocprofv3 --kernel-trace CSV of bench.py (forward mode, --no-kernel-events):
busy time, gaps and the largest gap owners on every HIP queue, averaged over the last N forwards.
usage: python tools/timeline.py <kernel_trace.csv> <num_forwards>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
nf = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "sfc_encode" in r["Kernel_Name"]][-nf:]
sel = rows[starts[0]:]
t0 = int(sel[0]["Start_Timestamp"])
t1 = max(int(r["End_Timestamp"]) for r in sel)
print(f"wall {(t1 - t0) / 1e3 / nf:.1f} us/forward, {len(sel) / nf:.0f} launches/forward")
byq = collections.defaultdict(list)
for r in sel:
    byq[r["Queue_Id"]].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / 1e3
    gaps = collections.defaultdict(lambda: [0, 0.0])
    last = None
    for r in rs:
        s = int(r["Start_Timestamp"])
        if last is not None and s > int(last["End_Timestamp"]):
            g = (s - int(last["End_Timestamp"])) / 1e3
            nm = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ptv3::", "")[:48]
            gaps[nm][0] += 1
            gaps[nm][1] += g
        last = r
    tot_gap = sum(v[1] for v in gaps.values())
    print(f"queue {q}: {len(rs) / nf:.0f} launches, busy {busy / nf:.1f} us, gaps {tot_gap / nf:.1f} us per forward")
    for nm, (c, g) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:10]:
        print(f"    gap before {nm:48s} n={c / nf:5.1f}  {g / nf:8.1f} us/forward  avg {g / c:6.1f} us")
