run() { echo -n "$1: "; shift; "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), round(d['host_enqueue_ms_per_step'],2))"; }
B="bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5"
run plain python $B
run plain_forcedist env PV2_BENCH_FORCE_DIST=1 python $B
run plain_forcedist_skipsync_nooverlap env PV2_BENCH_FORCE_DIST=1 PV2_BENCH_SKIP_SYNC=1 PV2_GSYNC_OVERLAP=0 python $B
run torchrun_nodist python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 $B --gpus 1
run torchrun_forcedist_omp8 env PV2_BENCH_FORCE_DIST=1 OMP_NUM_THREADS=8 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 $B --gpus 1
