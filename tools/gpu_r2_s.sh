#!/bin/bash
# The multi-process code path of bench.py on a 1-GPU box: torchrun with one rank, RCCL process
# group + gradient sync + barriers forced on (PV2_BENCH_FORCE_DIST=1).
set -u
O=gpurun_out/r2s; mkdir -p $O
export PV2_BENCH_FORCE_DIST=1
port=29550
for gs in flat ddp; do for amp in "" "--amp bf16"; do
port=$((port+1))
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --grad-sync $gs $amp > $O/bench_${gs}.json 2> $O/bench_${gs}.err; echo "[$gs $amp] rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_${gs}.json) $(grep -o '"final_loss": [0-9.e-]*' $O/bench_${gs}.json)"; grep -i "error\|Traceback" $O/bench_${gs}.err | head -3
done; done
unset PV2_BENCH_FORCE_DIST
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $O/bench_plain.json 2>/dev/null; echo "[plain] $(grep -o '"ms_per_step": [0-9.]*' $O/bench_plain.json)"
