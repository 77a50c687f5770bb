"""Per-step summary of a rocprofv3 --kernel-trace --stats csv of `bench.py --mode train` (kernels grouped by label).
usage: python tools/train_kernel_stats.py <kernel_stats.csv> <steps incl. warmup>"""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
try:
    from kernel_names import label
except Exception:  # noqa: BLE001
    label = lambda s: s  # noqa: E731

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
acc = {}
for r in rows:
    k = label(r["Name"])[:72]
    a = acc.setdefault(k, [0, 0.0])
    a[0] += int(r["Calls"])
    a[1] += float(r["TotalDurationNs"])
tot = sum(t for _, t in acc.values())
print(f"kernel time {tot / 1e6 / steps:.2f} ms/step, {sum(c for c, _ in acc.values()) / steps:.0f} launches/step")
for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{k:72s} calls/step={c / steps:7.1f} ms/step={t / 1e6 / steps:7.3f} avg_us={t / c / 1e3:8.1f}")
