#!/usr/bin/env python
"""Device timeline of the training step from a rocprofv3 kernel trace.

usage: tools/timeline.py <kernel_trace.csv> [n_steps]

Splits the trace into steps at the optimizer's launches, then reports for the last ``n_steps``:
wall per step, busy time per queue, time with NO kernel running on any queue (the device waits for
the host or for a dependency), the largest such gaps with the kernels either side, and - along the
step - which share of every millisecond was idle.  This is the picture that tells a launch-bound
step (many small gaps everywhere) from a kernel-bound one (no gaps, long kernels)."""
import csv
import sys
from collections import defaultdict


def short(name, n=60):
    for junk in ("void ", "at::native::", "(anonymous namespace)::", "pv2::", "at::cuda::detail::",
                 "elementwise_kernel_manual_unroll<128, 4, gpu_kernel_impl_nocast<", "rocprim::detail::"):
        name = name.replace(junk, "")
    return name.split("(")[0][:n]


def main():
    path = sys.argv[1]
    want = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    rows = []
    with open(path) as f:
        reader = csv.DictReader(f)
        if "Start_Timestamp" not in reader.fieldnames:
            sys.exit("unexpected header: %s" % reader.fieldnames)
        for r in reader:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                         r.get("Queue_Id", "0"), r.get("Stream_Id", "0")))
    rows.sort()
    # step boundaries: the end of every cluster of optimizer launches
    opt = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r[2]]
    ends = []
    for a, b in zip(opt, opt[1:] + [None]):
        if b is None or rows[b][0] - rows[a][1] > 2_000_000:   # > 2 ms apart: next step's cluster
            ends.append(rows[a][1])
    if len(ends) < want + 1:
        print("only", len(ends), "steps in the trace")
        want = len(ends) - 1
    ends = ends[-(want + 1):]
    tot = defaultdict(float)
    queue_busy = defaultdict(float)
    gaps_all = []
    n_launch = 0
    prof = defaultdict(lambda: [0.0, 0.0])   # ms bucket -> [idle, wall]
    bucket_names = defaultdict(lambda: defaultdict(int))
    for t0, t1 in zip(ends[:-1], ends[1:]):
        ks = [r for r in rows if r[0] >= t0 and r[1] <= t1 + 1]
        n_launch += len(ks)
        tot["wall"] += (t1 - t0) / 1e6
        for r in ks:
            bucket_names[int((r[0] - t0) / 1e6)][short(r[2])] += 1
            queue_busy[(r[3], r[4])] += (r[1] - r[0]) / 1e6
            tot["kernel_sum"] += (r[1] - r[0]) / 1e6
        # union of busy intervals
        cur_end, last_name = t0, "<step start>"
        for r in ks:
            if r[0] > cur_end:
                g = r[0] - cur_end
                gaps_all.append((g / 1e3, (cur_end - t0) / 1e6, last_name, short(r[2])))
                tot["idle"] += g / 1e6
                b0 = int((cur_end - t0) / 1e6)
                prof[b0][0] += g / 1e6
            if r[1] > cur_end:
                cur_end, last_name = r[1], short(r[2])
        if t1 > cur_end:
            tot["idle"] += (t1 - cur_end) / 1e6
    n = len(ends) - 1
    print(f"{path}: {n} steps, {n_launch / n:.0f} launches per step")
    print(f"  wall {tot['wall'] / n:8.3f} ms   kernel-time sum {tot['kernel_sum'] / n:8.3f} ms   "
          f"no kernel on any queue {tot['idle'] / n:8.3f} ms")
    for q, v in sorted(queue_busy.items(), key=lambda kv: -kv[1]):
        print(f"  queue {q[0]:>3} stream {q[1]:>3}: busy {v / n:8.3f} ms per step")
    sizes = [g[0] for g in gaps_all]
    for lo, hi in ((0, 2), (2, 5), (5, 10), (10, 20), (20, 50), (50, 200), (200, 1e9)):
        sel = [s for s in sizes if lo <= s < hi]
        print(f"  gaps {lo:>4}-{hi if hi < 1e9 else 'inf':>4} us: {len(sel) / n:7.1f} per step, {sum(sel) / n / 1e3:7.3f} ms per step")
    print("  per millisecond of the step (averaged): idle ms, launches, most frequent kernels")
    for b in range(int(tot["wall"] / n) + 1):
        names = sorted(bucket_names[b].items(), key=lambda kv: -kv[1])[:3]
        print(f"    {b:3d}  idle {prof[b][0] / n:.2f}  launches {sum(bucket_names[b].values()) / n:6.1f}  "
              + " | ".join(f"{k[:38]} x{v / n:.0f}" for k, v in names))
    print("  largest gaps (us, at ms into the step, after -> before):")
    for g in sorted(gaps_all, reverse=True)[:25]:
        print(f"    {g[0]:8.1f}  @{g[1]:6.2f}  {g[2]}  ->  {g[3]}")
    # which kernels follow the gaps most often, weighted by gap time
    after = defaultdict(float)
    for g in gaps_all:
        after[g[3]] += g[0]
    print("  gap time by the kernel that ends the gap (ms per step):")
    for k, v in sorted(after.items(), key=lambda kv: -kv[1])[:25]:
        print(f"    {v / n / 1e3:7.3f}  {k}")


if __name__ == "__main__":
    main()
