#!/usr/bin/env python
"""Merge the per-kernel FETCH_SIZE / WRITE_SIZE summaries (tools/pmc_summary.py output of two
separate rocprofv3 --pmc passes) into the JSON bench.py reads for `roofline.traffic`:
    python tools/pmc_to_json.py fetch_by_kernel.csv write_by_kernel.csv <commit> out.json
The file is stamped with bench.kernel_source_hash() of the tree it is run in (run it on the GPU box,
next to the sources the counters were collected on): bench.py refuses a file whose hash differs."""
import csv
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def short(name):
    name = re.sub(r"^void\s+", "", name)
    depth, out = 0, []
    for ch in name:          # cut at the argument list, keep template arguments
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).strip()


def load(path, column):
    res = {}
    for r in csv.DictReader(open(path)):
        res[short(r["kernel"])] = (float(r[column]), int(r["dispatches"]), float(r["avg_us"]))
    return res


def main():
    fetch = load(sys.argv[1], "FETCH_SIZE_per_dispatch")
    write = load(sys.argv[2], "WRITE_SIZE_per_dispatch")
    kernels = {}
    for k, (f, n, us) in fetch.items():
        if k in write:
            kernels[k] = dict(fetch_kb_per_launch=f, write_kb_per_launch=write[k][0], launches=n,
                              avg_us=us)
    import bench

    json.dump(dict(commit=sys.argv[3], kernel_source_hash=bench.kernel_source_hash(), note="KB per launch; FETCH_SIZE is to be doubled on gfx950 "
                   "(MI355X_MICROARCH.md, HBM section)", kernels=kernels),
              open(sys.argv[4], "w"), indent=1)
    print("wrote", sys.argv[4], len(kernels), "kernels")


if __name__ == "__main__":
    main()
