"""Per-kernel SQ counters of a rocprofv3 --pmc pass over bench.py -> JSON + table (issue / stall / matrix-core shares).
usage: python tools/pmc_sq.py <counter_collection.csv> <forwards> <out.json>
Counters (gfx950, ROCm 7.2; SQ counters are summed over all CUs / XCDs by rocprofv3):
  SQ_BUSY_CU_CYCLES       cycles a CU had at least one wave (quad-cycle units x4 folded by rocprof)
  SQ_ACTIVE_INST_VALU     wave-cycles spent issuing VALU (incl. MFMA issue)      SQ_ACTIVE_INST_ANY  any instruction
  SQ_WAIT_INST_ANY        wave-cycles stalled on issue (dependency / pipe busy)   SQ_WAIT_ANY  parked on s_waitcnt / barrier
  SQ_VALU_MFMA_BUSY_CYCLES  cycles the matrix pipe was busy                      SQ_INSTS_VALU_TRANS_F32  v_exp / v_rcp ... count
MFMA utilisation is reported as SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES): share of the SIMD-cycles of
the CUs that held waves of this kernel during which the matrix pipe was executing."""
import collections
import csv
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import label  # noqa: E402


def main():
    path, nf, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    rows = list(csv.DictReader(open(path)))
    starts = sorted({int(r["Dispatch_Id"]) for r in rows if "sfc_encode" in r["Kernel_Name"]})[-nf:]
    lo = starts[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in rows:
        d = int(r["Dispatch_Id"])
        if d < lo:
            continue
        n = r["Kernel_Name"]
        key = label(n)
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[key].add(d)
    res = {}
    for k, c in agg.items():
        busy = c.get("SQ_BUSY_CU_CYCLES", 0.0)
        e = {kk: round(vv / nf, 1) for kk, vv in c.items()}
        e["launches_per_forward"] = round(len(disp[k]) / nf, 1)
        if busy > 0:
            e["mfma_util"] = round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * busy), 4)
        act, wi, wa = c.get("SQ_ACTIVE_INST_ANY", 0.0), c.get("SQ_WAIT_INST_ANY", 0.0), c.get("SQ_WAIT_ANY", 0.0)
        if act + wi + wa > 0:
            e["share_issuing"] = round(act / (act + wi + wa), 3)
            e["share_issue_stall"] = round(wi / (act + wi + wa), 3)
            e["share_parked"] = round(wa / (act + wi + wa), 3)
            e["valu_share_of_issue"] = round(c.get("SQ_ACTIVE_INST_VALU", 0.0) / max(act, 1.0), 3)
        res[k] = e
    json.dump({"forwards_averaged": nf, "kernels": res}, open(out, "w"), indent=1)
    top = sorted(res.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CU_CYCLES", 0.0))[:10]
    for k, e in top:
        print(f"{k[:40]:40s} mfma_util={e.get('mfma_util', 0):6.3f} issuing={e.get('share_issuing', 0):5.2f} "
              f"issue_stall={e.get('share_issue_stall', 0):5.2f} parked={e.get('share_parked', 0):5.2f} "
              f"valu/issue={e.get('valu_share_of_issue', 0):5.2f} n={e['launches_per_forward']}")


if __name__ == "__main__":
    main()
