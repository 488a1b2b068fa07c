#!/bin/bash
set -u
O=gpurun_out/r06; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
PV2_WGRAD_STREAM=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_amp -- python $R/bench.py --amp bf16 --scenes-per-gpu 8 --views 5 --steps 6 --warmup 2 --no-kernel-timing --no-cpu-baseline > $R/gpurun_out/prof_amp.log 2>&1
cd $R
find gpurun_out/prof_amp -name "*kernel_trace.csv" -delete
f=$(find gpurun_out/prof_amp -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_amp_shipped_single.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats_amp_shipped_single.csv")))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
n=8
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms per step", tot/n/1e6)
for r in rows[:32]:
    nm=r["Name"].replace("(anonymous namespace)::","").replace("void ","")
    print("%7.3f ms/step %6.1f calls/step %8.1f us  %s"%(float(r["TotalDurationNs"])/n/1e6, float(r["Calls"])/n, float(r["AverageNs"])/1e3, nm[:100]))
PY
