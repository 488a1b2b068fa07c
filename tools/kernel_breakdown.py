#!/usr/bin/env python
"""Per-step kernel time by category from a rocprofv3 `--kernel-trace --stats` summary
(profiles/*kernel_stats*.csv) - the breakdown quoted in DESIGN.md section 6.

    python tools/kernel_breakdown.py profiles/r02_rocprofv3_kernel_stats_v10_f32_single_stream.csv [steps=13]

`steps` = timed + warm-up steps of the profiled bench command (tools/gpu_prof.sh: 10 + 3)."""
import csv
import sys

RULES = (  # first match wins
    ("sparse conv weight gradient (partials + ordered reduce)", ("spconv_wgrad", "wgrad_reduce")),
    ("ordered row reduce (+ fused BN statistics)", ("row_reduce",)),
    ("weight packing (16-bit)", ("pack_weights",)),
    ("dense U-Net convs, hand-written (csrc/dense_conv.hip)", ("dconv",)),
    ("device GridSample (csrc/voxelize.hip)", ("voxel_",)),
    ("sparse conv forward / grad-input", ("spconv_",)),
    ("tall / skinny GEMMs (heads, 1x1x1 conv)", ("tall_gemm", "skinny_gemm")),
    ("dense U-Net convs + BatchNorm (MIOpen / CK / hipBLASLt)", ("ck::", "_ZN2ck", "Cijk", "MIOpen", "miopen")),
    ("fills", ("zero_words", "fillBuffer", "FillFunctor")),
    ("sparse BatchNorm + column sums", ("col_partials", "col_combine", "bn_", "col_sum")),
    ("fused ray march", ("field_", "coarse_sample", "volume_scatter", "weights_", "accumulate_", "fold_", "narrow_")),
    ("optimizer", ("multi_tensor", "sgd", "Sgd")),
    ("rulebook build", ("rocprim", "table", "hash", "down_", "tile_prefix", "fill_i32", "pair_positions")),
    ("dense max-pool / concat / split, ray set-up, losses (hand-written)", ("maxpool3d", "concat_rows", "split_rows", "small_inverse", "scene_bounds", "unit_cube", "ray_gen", "surface_loss")),
    ("ATen elementwise / copies / reductions", ("at::native", "copyBuffer")),
)


def category(name):
    for cat, keys in RULES:
        if any(k in name for k in keys):
            return cat
    return "other"


def main():
    path = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 13
    rows = list(csv.DictReader(open(path)))
    cats, launches = {}, {}
    for r in rows:
        c = category(r["Name"])
        cats[c] = cats.get(c, 0.0) + int(r["TotalDurationNs"]) / steps / 1e6
        launches[c] = launches.get(c, 0) + int(r["Calls"]) / steps
    print(f"{path}: {sum(cats.values()):.2f} ms of kernel time and {sum(launches.values()):.0f} launches per step")
    for c, v in sorted(cats.items(), key=lambda kv: -kv[1]):
        print(f"  {v:7.2f} ms  {launches[c]:6.0f} launches  {c}")


if __name__ == "__main__":
    main()
