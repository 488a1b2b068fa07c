#!/bin/bash
# Round-2 trip J: 16-bit kernel tests + profile of the --amp bf16 step.
set -u
O=gpurun_out/r2j; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_half.py -q -x -s > $O/pytest_half.txt 2>&1; echo "pytest rc=$?"; grep -E "cosine|passed|failed|Error" $O/pytest_half.txt | cut -c1-300
bash tools/gpu_prof.sh amp16 --steps 10 --warmup 3 --amp bf16
cp gpurun_out/prof_amp16_kernel_stats.csv $O/kernel_stats_amp.csv; rm -rf gpurun_out/prof_amp16
python - <<'PY'
import csv,re
rows=list(csv.DictReader(open('gpurun_out/r2j/kernel_stats_amp.csv')))
print("total kernel ms/step %.2f"%(sum(float(r['TotalDurationNs']) for r in rows)/1e6/13))
for r in rows[:26]:
    n=re.sub(r'\(anonymous namespace\)::|void ','',r['Name'])
    print("%6.3f ms %6.1f/step %8.1f us  %s"%(float(r['TotalDurationNs'])/1e6/13,int(r['Calls'])/13,float(r['AverageNs'])/1e3,n[:90]))
PY
grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_amp16.log
