#!/bin/bash
# Upper bound of what operands cut into bf16 pieces BEFOREHAND could buy: the same
# sources built with -DPV2_FAKE_SPLIT (bf16_rest = identity: the split costs one pack; results are wrong,
# timings are those of a kernel that finds its pieces ready - without the 1.5x gather bytes the real
# thing would add).  Build: make -C ponderv2_amd/csrc OUT=.../lib/libponderv2_fake.so OBJDIR=.../build/fake EXTRA=-DPV2_FAKE_SPLIT
set -u
mkdir -p gpurun_out/r06
{
echo "== per layer, real split =="; python tools/bench_spconv_kernels.py 2>&1 | grep -E "^L|levels"
echo "== per layer, split free (probe build) =="; PV2_PROBE_LIB=libponderv2_fake.so python tools/bench_spconv_kernels.py 2>&1 | grep -E "^L|levels"
for lib in "" libponderv2_fake.so; do
  echo "== bench.py, lib=${lib:-product} =="
  PV2_PROBE_LIB=$lib timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('ms_per_step', round(d['ms_per_step'],2))
        for k in d.get('kernels',[])[:14]: print('  ', k)"
done
} | tee gpurun_out/r06/fake_split.txt
