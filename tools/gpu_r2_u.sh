#!/bin/bash
# Round-2 trip U: GPU test suite + A/B of the backward side stream and the zero arenas.
set -u
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_golden.py tests/test_gpu_sidestream.py tests/test_gpu_fused_head.py tests/test_gpu_half.py -m gpu -q > gpurun_out/u_pytest.txt 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/u_pytest.txt | tail -15
ab() {  # tag, env..., -- bench args
  tag=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 200 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 --warmup 5 "$@" \
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
ab f32_new_a      --
ab f32_norays_a   PV2_RAYS_STREAM=0 --
ab f32_nofused_a  PV2_FUSED_OPTIMIZER=0 --
ab f32_new_b      --
ab f32_norays_b   PV2_RAYS_STREAM=0 --
ab f32_nofused_b  PV2_FUSED_OPTIMIZER=0 --
OLD="PV2_WGRAD_STREAM=0 PV2_RAYS_STREAM=0 PV2_FUSED_OPTIMIZER=0"
ab bf16_old_a     $OLD -- --amp bf16
ab bf16_noside_a  PV2_WGRAD_STREAM=0 -- --amp bf16
ab bf16_new_a     -- --amp bf16
ab bf16_old_b     $OLD -- --amp bf16
ab bf16_noside_b  PV2_WGRAD_STREAM=0 -- --amp bf16
ab bf16_new_b     -- --amp bf16
