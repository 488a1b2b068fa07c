#!/usr/bin/env python
"""Print a rocprofv3 kernel_stats.csv as ms/step: tools/kstats.py file.csv n_steps [rows]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("total %.2f ms/step over %g steps" % (tot / 1e6 / steps, steps))
for r in rows[:top]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    print("%-86s %6.1f/step %8.3f ms/step %8.1f us" % (n[:86], int(r["Calls"]) / steps,
          int(r["TotalDurationNs"]) / 1e6 / steps, float(r["AverageNs"]) / 1e3))
