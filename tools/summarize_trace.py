"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel totals for the last `steps` forwards and GPU idle time.
usage: python tools/summarize_trace.py <kernel_trace.csv> <num_forwards_in_trace>"""
import csv, sys, collections, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import label
rows = list(csv.DictReader(open(sys.argv[1])))
nf = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# forwards are delimited by sfc_encode launches
starts = [i for i, r in enumerate(rows) if "sfc_encode" in r["Kernel_Name"]]
starts = starts[-nf:]
sel = rows[starts[0]:]
t_begin, t_end = int(sel[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in sel)
agg = collections.defaultdict(lambda: [0, 0.0])
busy = 0.0
last_end = t_begin
idle = 0.0
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"]
    short = label(name)
    agg[short][0] += 1
    agg[short][1] += (e - s) / 1e3
    if s > last_end:
        idle += (s - last_end) / 1e3
    last_end = max(last_end, e)
print(f"forwards={nf} wall={(t_end - t_begin)/1e3/nf:.1f} us/forward  idle={idle/nf:.1f} us/forward  launches/forward={len(sel)/nf:.0f}")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"{k:60s} n={c/nf:6.1f} {t/nf:9.1f} us/forward  avg={t/c:7.1f} us")
