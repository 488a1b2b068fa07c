#!/usr/bin/env python
"""Per-stream kernel sums of a rocprofv3 kernel trace: which kernels fill the TRAINING stream (the chain the
step's wall time follows) and which run beside it.  usage: tools/stream_breakdown.py <kernel_trace.csv> [steps]"""
import csv
import sys
from collections import defaultdict


def short(name, n=64):
    for junk in ("void ", "at::native::", "(anonymous namespace)::", "pv2::"):
        name = name.replace(junk, "")
    return name.split("(")[0][:n]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    per = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for r in rows:
        key = (r.get("Queue_Id", "0"), r.get("Stream_Id", "0"))
        e = per[key][short(r["Kernel_Name"])]
        e[0] += 1
        e[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for key, ks in sorted(per.items(), key=lambda kv: -sum(v[1] for v in kv[1].values())):
        tot = sum(v[1] for v in ks.values())
        cnt = sum(v[0] for v in ks.values())
        print("queue %s stream %s: %.2f ms and %.0f launches per step" % (key[0], key[1], tot / steps / 1e3, cnt / steps))
        for name, (c, t) in sorted(ks.items(), key=lambda kv: -kv[1][1])[:45]:
            print("   %8.1f us/step %6.1f launches/step %7.2f us avg  %s" % (t / steps, c / steps, t / c, name))


if __name__ == "__main__":
    main()
