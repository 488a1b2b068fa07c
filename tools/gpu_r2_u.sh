#!/bin/bash
# Round-2 trip U: GPU test suite + A/B of the backward side stream and the zero arenas.
set -u
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q > gpurun_out/u_pytest.txt 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/u_pytest.txt | tail -15
ab() {  # tag, env..., -- bench args
  tag=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 200 python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 "$@" \
      > gpurun_out/u_ab_$tag.json 2> gpurun_out/u_ab_$tag.err
  python - "$tag" <<'PY'
import json,sys
tag=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/u_ab_{tag}.json").read().strip().splitlines()[-1])
    print(f"{tag:28s} {d['ms_per_step']:7.2f} ms/step  host {d['host_enqueue_ms_per_step']:6.2f}  loss {d['final_loss']:.4f}  side={d.get('backward_side_stream')}")
except Exception as e:
    print(tag, "FAILED", e); print(open(f"gpurun_out/u_ab_{tag}.err").read()[-600:])
PY
}
OLD="PV2_WGRAD_STREAM=0 PV2_POINTWISE_CONV=0 PV2_PREFETCH_RAYS=0"
ab f32_old        $OLD -- --no-stage-thread
ab f32_side_pw    PV2_PREFETCH_RAYS=0 -- --no-stage-thread
ab f32_thread     PV2_PREFETCH_RAYS=0 --
ab f32_rays       PV2_PREFETCH_RAYS=1 -- --no-stage-thread
ab f32_thread_rays PV2_PREFETCH_RAYS=1 --
ab bf16_old       $OLD -- --no-stage-thread --amp bf16
ab bf16_side      PV2_PREFETCH_RAYS=0 -- --no-stage-thread --amp bf16
ab bf16_thread    PV2_PREFETCH_RAYS=0 -- --amp bf16
ab bf16_thread_rays PV2_PREFETCH_RAYS=1 -- --amp bf16
