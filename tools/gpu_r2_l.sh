#!/bin/bash
set -u
O=gpurun_out/r2l; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_half.py tests/test_gpu_kernels.py -q -x -k "bn or spunet" -s > $O/pytest.txt 2>&1; echo "pytest rc=$?"; grep -E "cosine|passed|failed|Error" $O/pytest.txt | cut -c1-300
bash tools/gpu_prof.sh amp16 --steps 10 --warmup 3 --amp bf16
cp gpurun_out/prof_amp16_kernel_stats.csv $O/kernel_stats_amp.csv; rm -rf gpurun_out/prof_amp16
python - <<'PY'
import csv,re
for tag in ("amp",):
    rows=list(csv.DictReader(open('gpurun_out/r2l/kernel_stats_%s.csv'%tag)))
    print(tag,"total kernel ms/step %.2f"%(sum(float(r['TotalDurationNs']) for r in rows)/1e6/13))
    for r in rows:
        n=re.sub(r'\(anonymous namespace\)::|void ','',r['Name'])
        if re.search(r'col_|bn_',n): print("  %6.3f ms %6.1f/step %8.1f us  %s"%(float(r['TotalDurationNs'])/1e6/13,int(r['Calls'])/13,float(r['AverageNs'])/1e3,n[:70]))
PY
