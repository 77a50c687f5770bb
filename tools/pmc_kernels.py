"""Per-kernel totals of one or more rocprofv3 --pmc passes over a micro-benchmark (no forward boundaries): counters
summed over the dispatches of each kernel label, divided by the number of dispatches.
usage: python tools/pmc_kernels.py <out.json> <counter_collection.csv> [more.csv ...] [--match substr]"""
import collections
import csv
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import label  # noqa: E402


def main():
    args = sys.argv[1:]
    match = None
    if "--match" in args:
        i = args.index("--match")
        match = args[i + 1]
        del args[i:i + 2]
    out, paths = args[0], args[1:]
    res = collections.defaultdict(dict)
    for path in paths:
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        for r in csv.DictReader(open(path)):
            key = label(r["Kernel_Name"])
            if match and match not in key:
                continue
            agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[key].add(int(r["Dispatch_Id"]))
        for k, c in agg.items():
            n = max(len(disp[k]), 1)
            res[k]["dispatches"] = n
            for kk, vv in c.items():
                res[k][kk] = round(vv / n, 1)
    for k, e in res.items():
        busy = e.get("SQ_BUSY_CU_CYCLES", 0.0)
        if busy > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in e:
            e["mfma_util"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * busy), 4)
        act, wi, wa = e.get("SQ_ACTIVE_INST_ANY", 0.0), e.get("SQ_WAIT_INST_ANY", 0.0), e.get("SQ_WAIT_ANY", 0.0)
        if act + wi + wa > 0:
            e["share_issuing"] = round(act / (act + wi + wa), 3)
            e["share_issue_stall"] = round(wi / (act + wi + wa), 3)
            e["share_parked"] = round(wa / (act + wi + wa), 3)
    json.dump(res, open(out, "w"), indent=1)
    for k, e in sorted(res.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CU_CYCLES", kv[1].get("dispatches", 0))):
        print(k[:60], json.dumps({a: b for a, b in e.items()}))


if __name__ == "__main__":
    main()
