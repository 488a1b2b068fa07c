#!/bin/bash
# The multi-process code path of bench.py on a 1-GPU box: torchrun with one rank, RCCL process
# group + DDP wrapper + barriers forced on (PV2_BENCH_FORCE_DIST=1).
set -u
O=gpurun_out/r2s; mkdir -p $O
export PV2_BENCH_FORCE_DIST=1
port=29540
for mode in static find_unused; do for amp in "" "--amp bf16"; do
port=$((port+1))
PV2_DDP_MODE=$mode timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing $amp > $O/bench_ddp1.json 2> $O/bench_ddp1.err; echo "ddp[$mode $amp] rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_ddp1.json) $(grep -o '"final_loss": [0-9.e-]*' $O/bench_ddp1.json)"; grep -i "error\|Traceback" $O/bench_ddp1.err | head -3
done; done
