#!/usr/bin/env python
"""Aggregate a rocprofv3 counter-collection CSV per kernel.

    rocprofv3 --kernel-trace --pmc <COUNTER...> --output-format csv -d <dir> -- <cmd>
    python tools/pmc_summary.py <dir>/**/<pid>_counter_collection.csv [out.csv]

Prints (and optionally writes) one row per kernel: dispatches, per-dispatch mean of every counter,
mean duration.  Keep every --pmc group in its own run (FETCH_SIZE and WRITE_SIZE do not fit in one
pass on gfx950; gpurun refuses --pmc together with the hip/hsa trace domains)."""
import csv
import re
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    rows = list(csv.DictReader(open(path)))
    per = defaultdict(lambda: {"n": set(), "ns": 0.0, "counters": defaultdict(float)})
    for r in rows:
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        rec = per[name]
        key = (r.get("Dispatch_Id"), r.get("Correlation_Id"))
        if key not in rec["n"]:
            rec["n"].add(key)
            rec["ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        rec["counters"][r["Counter_Name"]] += float(r["Counter_Value"])
    counters = sorted({c for rec in per.values() for c in rec["counters"]})
    out = [["kernel", "dispatches", "avg_us"] + [c + "_per_dispatch" for c in counters]]
    for name, rec in sorted(per.items(), key=lambda kv: -kv[1]["ns"]):
        n = len(rec["n"])
        out.append([name[:120], n, "%.2f" % (rec["ns"] / n / 1e3)]
                   + ["%.4g" % (rec["counters"].get(c, 0.0) / n) for c in counters])
    w = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
    w.writerows(out)


if __name__ == "__main__":
    main()
