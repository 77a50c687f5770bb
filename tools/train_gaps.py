"""GPU busy / idle per training step from a rocprofv3 kernel trace of `bench.py --mode train`, and the kernels that
FOLLOW the largest idle gaps (= what the host was slow to issue).  usage: python tools/train_gaps.py <kernel_trace.csv>"""
import csv
import os
import sys
from collections import Counter

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
try:
    from kernel_names import label
except Exception:  # noqa: BLE001
    label = lambda s: s  # noqa: E731

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
marks = [i for i, e in enumerate(ev) if "adamw_kernel" in e[2]]
marks = [m for j, m in enumerate(marks) if j + 1 == len(marks) or ev[marks[j + 1]][0] - ev[m][0] > 5e6]   # last launch of a step
after = Counter()
for a, b in zip(marks[:-1], marks[1:]):
    seg = ev[a + 1:b + 1]
    span = (seg[-1][1] - seg[0][0]) / 1e6
    busy = sum(e[1] - e[0] for e in seg) / 1e6
    gaps = [(seg[i + 1][0] - seg[i][1], label(seg[i + 1][2])[:50], label(seg[i][2])[:40]) for i in range(len(seg) - 1)]
    big = [g for g in gaps if g[0] > 15000]
    print(f"step: span {span:.2f} ms busy {busy:.2f} idle {span - busy:.2f} launches {len(seg)}; "
          f"gaps > 15 us: {len(big)} totalling {sum(g[0] for g in big) / 1e6:.2f} ms")
    for g in big:
        after[(g[1], g[2])] += g[0]
print("idle time by (kernel after the gap <- kernel before), ms over all steps:")
for (k, prev), t in after.most_common(25):
    print(f"  {t / 1e6:7.2f}  {k:50s} <- {prev}")
