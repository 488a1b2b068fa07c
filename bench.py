#!/usr/bin/env python
"""PonderV2 pre-training throughput on MI355X: scenes/s (+ rays/s) of one training step
(forward + backward + SGD step) of PonderIndoor-v2 / SpUNet-v1m1 on synthetic ScanNet-shaped
scenes (BASELINE.json config 2: 2 scenes/GPU, 512 rays/scene; weak scaling over GPUs).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0).  Besides the contract keys it carries
  roofline     - the dominant hand-written kernel family, timed live with HIP events on the launch
                 stream over a second, identical pass of the K timed steps (so the events do not
                 perturb `value`), against the gfx950 peak that bounds it;
  kernels      - the same measurement for every instrumented kernel family;
  cpu_baseline - the same model code on the host cores with the oracle's CPU kernels
                 (rank 0, N=1 only; a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
# MIOpen's find results for the dense UNet3D convs ship in-tree (miopen_cache/): without them the
# first step spends ~95 s benchmarking solvers on every fresh box.  Must be set before MIOpen loads.
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(ROOT, "miopen_cache", "db"))
os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", os.path.join(ROOT, "miopen_cache", "cache"))

if int(os.environ.get("WORLD_SIZE", "1") or "1") > 1 or os.environ.get("PV2_BENCH_FORCE_DIST") == "1":
    # With an RCCL communicator in the process ROCclr's default of four hardware queues costs the step
    # +1.6 ms even when NO reduction is issued, +2.9 ms with one (one rank, MI355X; 21.6 against 18.7 ms);
    # three or two queues: +0.4 ms.  Round 6 measured that it is NOT the number of this program's streams
    # (four instead of six changed nothing, profiles/r06_one_rank_pg.txt), so the launchers keep round 5's
    # setting for ranks of a process group; the library no longer sets it on import.  Read at HIP
    # initialisation, hence up here; an explicit value in the environment wins.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16 / _f16, dense
# fp32 products computed as SIX bf16 MFMAs over three exact bf16 pieces per operand (csrc/mfma_split.h):
# the matrix roofline of those kernels, in fp32-equivalent (= algorithmic) TFLOP/s
SPLIT_MFMA_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 6.0
SPLIT_ON = os.environ.get("PV2_FP32_MFMA") != "1"
SKIP_SYNC = os.environ.get("PV2_BENCH_SKIP_SYNC") == "1"   # (diagnosis: process group present, no reduction)
SPLIT_FAMILIES = ("spconv_fwd_lds_kernel", "spconv_wgrad_split_kernel", "dconv_split_kernel",
                  "dconvT_split_kernel", "dconv_strided_split_kernel", "dconv_wgrad_split_kernel",
                  "field_fwd_rows_kernel")


def mfma_peak_of(family, default):
    """Dense matrix peak that bounds a kernel family, in algorithmic TFLOP/s."""
    if SPLIT_ON and family.startswith(SPLIT_FAMILIES):
        return SPLIT_MFMA_PEAK_TFLOPS
    return default
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s


PMC_FILE = "profiles/r06_pmc_fetch_write_per_kernel.json"


def kernel_source_hash():
    """sha256 (16 hex digits) over the sources of the sparse-conv kernels (csrc/sparse_conv*.hip +
    common.h, mfma_split.h) - the kernels the `roofline` line is about: what their per-kernel PMC measurement
    stays valid for.  tools/pmc_to_json.py stamps the same value into the PMC file."""
    import hashlib

    h = hashlib.sha256()
    src = os.path.join(ROOT, "ponderv2_amd", "csrc")
    files = sorted(os.path.join(src, f) for f in os.listdir(src)
                   if f.startswith("sparse_conv") and f.endswith(".hip"))
    for f in files + [os.path.join(src, "common.h"), os.path.join(src, "mfma_split.h")]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel_key):
    """(HBM-side bytes per launch, source) of a kernel from the rocprofv3 PMC passes committed with
    this round (tools/gpu_pmc.sh: separate --pmc FETCH_SIZE / WRITE_SIZE runs of this bench command;
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 16-B/lane streaming reads).
    The counters cannot be collected inside a timed run, so this is a read of that file - accepted
    only while the kernel sources are the ones it was measured on (``kernel_source_hash``): a file
    measured on other sources is REFUSED (traffic null) rather than quoted.  (None, reason) when
    absent or stale."""
    path = os.path.join(ROOT, PMC_FILE)
    try:
        with open(path) as f:
            doc = json.load(f)
        if doc.get("kernel_source_hash") != kernel_source_hash():
            return None, (f"{PMC_FILE} is stale: measured on kernel sources "
                          f"{doc.get('kernel_source_hash', '?')} (commit {doc.get('commit', '?')}), "
                          f"the sources here hash to {kernel_source_hash()}; re-run tools/gpu_pmc.sh")
        # every instantiation whose name starts with the key (e.g. "<4, false>" and "<4, true>":
        # forward and grad-input), weighted by launches
        recs = [r for k, r in doc["kernels"].items() if k.startswith(kernel_key)]
        n = sum(r.get("launches", 1) for r in recs)
        kb = sum((2.0 * r["fetch_kb_per_launch"] + r["write_kb_per_launch"]) * r.get("launches", 1)
                 for r in recs) / n
        return (1024.0 * kb,
                f"{PMC_FILE} (same kernel sources {doc['kernel_source_hash']}; measured at commit "
                f"{doc.get('commit', '?')}, {n} launches)")
    except Exception as e:  # noqa: BLE001
        return None, f"{PMC_FILE} unavailable ({type(e).__name__})"


def _backbone_route():
    """Which route the sparse U-Net took over the run (counters of the product modules)."""
    from ponderv2_amd import convbn, spunet_native

    return ("native executor: %d calls (csrc/spunet_exec.hip); conv + BatchNorm units outside it: %d"
            % (spunet_native.CALLS, getattr(convbn, "CALLS", 0)))


def spawn_ranks(n):
    """``python bench.py --gpus N`` without a launcher: re-run this command line as N ranks of one
    node under torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1), the way the
    reference's launch() spawns its workers (ponder/engines/launch.py:38-100, tools/train.py:34).
    Rank 0 of the children prints the JSON line; this process just forwards the exit code."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def launch_check(args):
    """--launch-check: the multi-process plumbing of this script without the model - process group
    (gloo when there is no GPU: this is what the CPU test runs), barrier, max-over-ranks reduction of
    a per-rank timing, one JSON line from rank 0.  NOT a measurement (says so in the line)."""
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    gpu = torch.cuda.is_available()
    if world > 1:
        if gpu:
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl" if gpu else "gloo")
        dist.barrier()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64, device="cuda" if gpu else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "gpus_requested": args.gpus,
                          "max_over_ranks": float(t), "backend": "nccl" if gpu else "gloo",
                          "note": "plumbing check only: no model ran, not a measurement"}))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="indoor", choices=["indoor", "outdoor", "ppt"],
                    help="indoor = BASELINE configs[1] (the headline metric); outdoor = the "
                         "PonderOutdoor-v2 / nuScenes-shaped step of configs[4] on this GPU's shard; "
                         "ppt = configs[3]: multi-dataset indoor pre-training (SpUNet-v1m3 PDNorm, "
                         "batches alternate Structured3D/ScanNet/S3DIS conditions 4:2:1)")
    ap.add_argument("--scenes-per-gpu", type=int, default=None,
                    help="default 2 (indoor, configs[1]) / 4 (outdoor, the reference's per-GPU batch)")
    ap.add_argument("--rays-per-camera", type=int, default=512, help="outdoor: RaySample.point_nsample")
    ap.add_argument("--views", type=int, default=2)
    ap.add_argument("--rays-per-view", type=int, default=256)
    ap.add_argument("--grad-sync", choices=["flat", "ddp"], default="flat",
                    help="multi-GPU gradient averaging: one flat RCCL all-reduce after backward "
                         "(utils/grad_sync.py) or torch DistributedDataParallel")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="build the sparse-conv geometry inside the step (one blocking read) "
                         "instead of one batch ahead on the side stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--dense-dtype", default="float32", choices=["bfloat16", "float16", "float32"],
                    help="dtype of the dense UNet3D projection (MIOpen).  float32 (default) = the "
                         "parity configuration, everything on the path in fp32; bfloat16 runs only "
                         "those dense convs under autocast (the reference config trains with "
                         "enable_amp=True)")
    ap.add_argument("--amp", default=None, choices=["bf16", "fp16"],
                    help="run the WHOLE model under torch.autocast, as the reference's shipped ScanNet "
                         "config does (enable_amp=True): a separate, labelled line - the fp32 run "
                         "stays the headline / parity number.  The model scopes the reduced "
                         "precision to the dense UNet3D (library convolutions); the hand-written "
                         "kernels keep fp32 arithmetic")
    ap.add_argument("--config", default=None,
                    help="config file to take the model / optimizer / scheduler sections from "
                         "(default: the repository's synthetic-data config of the workload; the "
                         "reference's own config file works unchanged)")
    ap.add_argument("--raw-points", action="store_true",
                    help="indoor / ppt: the batches hold RAW (dropout-thinned) points; every timed step "
                         "voxelises its batch on the device first (datasets.voxelize.device_grid_sample: "
                         "hash, sort, one representative per voxel - the device half of the "
                         "reference's GridSample) and picks the ray pixels, i.e. the input pipeline "
                         "the reference leaves to its dataloader workers is INSIDE the timed region")
    ap.add_argument("--launch-check", action="store_true",
                    help="only exercise the rank spawning / process group / rank-0 reporting path "
                         "(no model, works without a GPU over gloo); prints a JSON line saying so")
    ap.add_argument("--print-losses", action="store_true")
    ap.add_argument("--kernel-table", default=None, help="write a per-layer-shape kernel table here")
    args = ap.parse_args()
    if args.scenes_per_gpu is None:
        args.scenes_per_gpu = 4 if args.workload == "outdoor" else 2
    return args


CONFIGS = {  # the configs/ files whose model / optimizer / scheduler sections the bench builds from
    "indoor": "configs/scannet/pretrain-ponder-spunet-v1m1-synthetic.py",
    "ppt": "configs/scannet/pretrain-ponder-ppt-v1m1-synthetic.py",
    "outdoor": "configs/nuscenes/pretrain-ponder-spunet-v1m1-synthetic.py",
}
PPT_SCHEDULE = (0, 0, 0, 0, 1, 1, 2)                                 # sampling ratio 4:2:1


def load_config(workload, path=None):
    """Config.fromfile on the repository's config for the workload (model / optimizer / scheduler
    sections identical to the reference's; only the data section is synthetic) or on ``path`` -
    e.g. the reference's own configs/scannet/pretrain-ponder-spunet-v1m1-0-base.py, which loads
    unchanged."""
    from ponderv2_amd.ponder.utils.config import Config

    return Config.fromfile(path or os.path.join(ROOT, CONFIGS[workload]))


def model_cfg(rays_per_view, dense_dtype="float32", workload="indoor", path=None):
    cfg = load_config(workload, path).model.to_dict()
    if workload != "outdoor":
        cfg["ray_nsample"] = rays_per_view
    if isinstance(cfg.get("template"), (list, tuple)):
        cfg["template"] = cfg["template"][0]   # the stub text encoder ignores the template
    cfg["proj_autocast"] = None if dense_dtype == "float32" else dense_dtype
    return cfg


def ppt_model_cfg(rays_per_view, dense_dtype="float32"):
    return model_cfg(rays_per_view, dense_dtype, "ppt")


def outdoor_model_cfg(dense_dtype="float32"):
    return model_cfg(None, dense_dtype, "outdoor")


def make_ppt_batches(rank, scenes, views, device, conditions, valid_index):
    from ponderv2_amd.ponder.datasets import collate_fn, make_scene

    out = []
    for k, cond in enumerate(conditions):
        samples = [make_scene(1000 * rank + 100 * k + i, num_views=views, image_hw=(480, 640),
                              condition=cond, num_classes=len(valid_index[k])) for i in range(scenes)]
        batch = collate_fn(samples)
        out.append({k2: (v.to(device) if torch.is_tensor(v) else v) for k2, v in batch.items()})
    return out


def make_outdoor_batch(rank, scenes, rays_per_camera, device):
    from ponderv2_amd.ponder.datasets import lidar_collate_fn, make_lidar_scene

    batch = lidar_collate_fn([make_lidar_scene(1000 * rank + i, point_nsample=rays_per_camera)
                              for i in range(scenes)])
    return {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}


class KernelTimer:
    """Wraps the kernels.* launch functions with HIP events recorded on the launch stream and
    accumulates algorithmic flops / bytes per kernel family."""

    def __init__(self):
        self.records = {}   # family -> list of (start, end, flops, bytes)
        self._orig, self._orig_c, self._handle = {}, {}, None
        self.l2_bytes = {}
        self.survey = {}
        self.peaks = {}     # family -> dense MFMA peak of its operand type (TFLOP/s)
        self._depth = 0     # > 0 inside a wrapped call: nested wrapped calls are not recorded twice
        self._excl = None   # event pairs of the zero-fill launches inside the call being timed
        self._pending = None   # dict of the wrapped call in flight; the launch hook leaves "events"

    def _add(self, fam, s, e, flops, nbytes, shape=None, excl=()):
        self.records.setdefault(fam, []).append((s, e, flops, nbytes, shape, tuple(excl)))

    @staticmethod
    def _ms(rec):
        """Duration of the bracketed launch itself: the bracket minus the zero-fill launches that
        ran inside it (the scatter-add kernels' targets are cleared right before them; the fills
        are a kernel of their own in the rocprofv3 statistics and a family of their own here)."""
        return rec[0].elapsed_time(rec[1]) - sum(a.elapsed_time(b) for a, b in rec[5])

    def install(self):
        import ponderv2_amd.kernels as K

        timer = self

        def wrap(name, fam, cost, peak=F32_MFMA_PEAK_TFLOPS):
            orig = getattr(K, name)
            self._orig[name] = orig
            self.peaks[fam] = peak

            def fn(*a, **k):
                if timer._depth > 0:
                    return orig(*a, **k)
                s = torch.cuda.Event(enable_timing=True)
                e = torch.cuda.Event(enable_timing=True)
                s.record()
                timer._depth += 1
                timer._excl = excl = []
                timer._pending = pend = {}
                try:
                    out = orig(*a, **k)
                finally:
                    timer._depth -= 1
                    timer._excl = None
                    timer._pending = None
                e.record()
                if pend.get("skip"):
                    return out
                if "events" in pend:   # the C-ABI launch itself was bracketed (launch hook below):
                    s, e = pend["events"]   # events right around the kernel launch, nothing else
                    excl = []
                c = cost(*a, **k)
                name_ = c[3] if len(c) > 3 else fam     # the instantiation this call ran on
                timer.peaks.setdefault(name_, peak)
                timer._add(name_, s, e, c[0], c[1], c[2] if len(c) > 2 else None, excl)
                return out

            setattr(K, name, fn)

        # the product-row conv path (kernels._pr_conv): two C-ABI launches per conv, each bracketed
        # by its own event pair.  The MFMA kernel (stage 1) carries the conv's flops; both stages
        # are priced by the bytes THEY have to move (stage 1: sources, weights, pair list in, one
        # product row per pair out; stage 2: the product rows in, the result rows out, the table) -
        # together SURVEY 8d's per-conv bytes plus one write and one read of the product rows.
        orig_pr = K._pr_conv
        self._orig["_pr_conv"] = orig_pr
        self._pr_ctx = None

        def timed_pr(feats, weight, rb, c_in, c_out, *a, **k):
            if timer._pending is not None:
                timer._pending["skip"] = True     # recorded here, not by the outer wrapper
            n_rows = a[4] if len(a) > 4 else k["n_rows"]
            timer._pr_ctx = dict(p=rb.n_pairs, c_in=c_in, c_out=c_out, K=rb.K,
                                 n_src=feats.shape[0], n_rows=n_rows)
            try:
                return orig_pr(feats, weight, rb, c_in, c_out, *a, **k)
            finally:
                timer._pr_ctx = None

        K._pr_conv = timed_pr

        orig_fill = K._zero_fill
        self._orig["_zero_fill"] = orig_fill

        def timed_fill(t):
            if timer._excl is None:
                return orig_fill(t)
            a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a_.record()
            orig_fill(t)
            b_.record()
            timer._excl.append((a_, b_))
            timer._add("zero_words_kernel (clears of the scatter-add targets)", a_, b_, 0.0,
                       float(t.numel() * t.element_size()))
            return t

        K._zero_fill = timed_fill

        def conv_family(c_in, c_out, rb, out):
            """Which kernel kernels.spconv_forward runs this call on (same rules as over there)."""
            if out is None and K._use_os(rb):
                return "spconv_os_kernel (fwd: the 125-offset stem, odd channel counts)"
            if c_in % 32 == 0:
                nb = min(4, (c_out + 31) // 32)
                return "spconv_fwd_lds_kernel<%d> (fwd+dgrad, %s output channels per workgroup)" % (
                    nb, {1: "32", 2: "64", 3: "96", 4: "128"}[nb])
            return "spconv_fwd_kernel (fwd: stem, any channel count)"

        def conv_cost(feats, w, rb, out=None, bias=None):
            c_out, kk, c_in = w.shape
            p = rb.n_pairs
            return (2.0 * p * c_in * c_out,
                    4.0 * (rb.n_in * c_in + rb.n_out * c_out + kk * c_in * c_out) + 8.0 * p,
                    (c_in, c_out, kk, p), conv_family(c_in, c_out, rb, out))

        def wgrad_cost(feats, gout, rb, c_out):
            c_in = feats.shape[1]
            p = rb.n_pairs
            return (2.0 * p * c_in * c_out,
                    4.0 * (rb.n_in * c_in + rb.n_out * c_out + rb.K * c_in * c_out) + 8.0 * p,
                    (c_in, c_out, rb.K, p), wgrad_family(c_in, c_out))

        def dgrad_cost(gout, w, rb):
            return conv_cost(gout, w.permute(2, 1, 0), rb.transposed())

        def wgrad_family(c_in, c_out):
            if c_in % 4 or c_out % 4:
                return "spconv_wgrad_kernel (stem, any channel count)"
            big_n, big_c = c_out > 64, c_in > 64
            return "%s<%s>%s" % (
                "spconv_wgrad_split_kernel" if SPLIT_ON and K.USE_WGRAD_DET else "spconv_wgrad_lds_kernel",
                "2, 2, 1" if big_n and big_c else "2, 1, 2" if big_n else "1, 2, 2" if big_c else "1, 1, 4",
                " + wgrad_reduce_kernel (two-stage, deterministic)" if K.USE_WGRAD_DET else "")

        # 16-bit kernels (csrc/sparse_conv16.hip): same algorithmic flops, 2-byte features and
        # weights, fp32 dW; the gather table instead of pair lists on the output-stationary pass
        def conv16_cost(feats, packed, kk, c_out, nbr, stride, perm, kflip, n_out, bias=None, n_pairs=0):
            n_in, c_in = feats.shape
            return (2.0 * n_pairs * c_in * c_out,
                    2.0 * (n_in * c_in + n_out * c_out + kk * c_in * c_out) + 4.0 * kk * n_out,
                    (c_in, c_out, kk, n_pairs))

        def wgrad16_cost(feats, gout, rb, c_out):
            c_in = feats.shape[1]
            p = rb.n_pairs
            return (2.0 * p * c_in * c_out,
                    2.0 * (rb.n_in * c_in + rb.n_out * c_out) + 4.0 * rb.K * c_in * c_out + 8.0 * p,
                    (c_in, c_out, rb.K, p))

        def tri_cost_factory(mult):
            def cost(*a, **k):
                grid = a[1] if mult == 1 else (a[2] if mult == 2 else a[3])
                inp = a[0] if mult == 1 else (a[1] if mult == 2 else a[2])
                pts = grid.numel() // 3
                c = inp.shape[1]
                return (0.0, pts * 8.0 * c * inp.element_size() * mult)
            return cost

        # fused ray-march entry points (csrc/raymarch_fused.hip): events around the C-ABI calls.
        # flops: the MLP GEMMs (2H*F + G*H + F*H MACs per sample forward, as many again backward,
        # weight gradients run in pv2_gemm_tn and are not counted here); bytes: everything the
        # launch writes plus the volume rows it gathers, each counted once (compulsory - the 8x
        # corner re-reads are served by L2/MALL and reported separately as l2_gather_gbs)
        import ponderv2_amd._lib as LIB

        handle = LIB.lib()
        self._handle = handle
        H_, F_, G_, C_, NV_ = 128, 64, 64, 128, 140

        def wrap_c(name, fam, cost):
            orig = getattr(handle, name)
            self._orig_c[name] = orig

            def fn(*a):
                s_ = torch.cuda.Event(enable_timing=True)
                e_ = torch.cuda.Event(enable_timing=True)
                s_.record()
                rc = orig(*a)
                e_.record()
                c = cost(*a)
                timer._add(fam, s_, e_, c[0], c[1], None)
                timer.l2_bytes[fam] = timer.l2_bytes.get(fam, 0.0) + c[2]
                return rc

            setattr(handle, name, fn)

        def hook_launch(cname):
            """Events directly around the C-ABI call that launches the kernel of a wrapped Python
            op (the wrapper above also spans the output allocation, the clear of a scatter-add
            target and stream bookkeeping: this is the pair rocprofv3's kernel duration must agree
            with)."""
            orig_c = getattr(handle, cname)
            self._orig_c[cname] = orig_c

            def fnc(*a):
                pend = timer._pending
                if pend is None or "events" in pend:
                    return orig_c(*a)
                s_ = torch.cuda.Event(enable_timing=True)
                e_ = torch.cuda.Event(enable_timing=True)
                s_.record()
                rc = orig_c(*a)
                e_.record()
                pend["events"] = (s_, e_)
                return rc

            setattr(handle, cname, fnc)

        for cname in ("pv2_spconv_forward", "pv2_spconv_forward_wt", "pv2_spconv_os_forward",
                      "pv2_spconv_backward_weight", "pv2_spconv_backward_weight_det",
                      "pv2_spconv16_os_forward", "pv2_spconv16_backward_weight"):
            hook_launch(cname)

        def hook_pr(cname, cost):
            orig_c = getattr(handle, cname)
            self._orig_c[cname] = orig_c

            def fnc(*a):
                ctx = timer._pr_ctx
                if ctx is None:
                    return orig_c(*a)
                s_ = torch.cuda.Event(enable_timing=True)
                e_ = torch.cuda.Event(enable_timing=True)
                s_.record()
                rc = orig_c(*a)
                e_.record()
                flops, nbytes, shape, fam = cost(ctx)
                timer.peaks.setdefault(fam, F32_MFMA_PEAK_TFLOPS)
                timer._add(fam, s_, e_, flops, nbytes, shape)
                return rc

            setattr(handle, cname, fnc)

        def products_cost(c):
            nb = min(4, (c["c_out"] + 31) // 32)
            fam = ("spconv_fwd_lds_kernel<%d> (product rows: fwd+dgrad, %s output channels per "
                   "workgroup)" % (nb, {1: "32", 2: "64", 3: "96", 4: "128"}[nb]))
            # bytes THIS kernel has to move: every source row and the weights once, the pair
            # list, and one product row per pair written (the output rows of SURVEY 8d's formula
            # are written by stage 2, which is priced separately below)
            # SURVEY 8(d)'s bytes of the whole conv (each feature row in once, each output row out
            # once, weights once, the pair list): what `traffic` is ALSO reported against
            timer.survey[fam] = timer.survey.get(fam, 0.0) + (
                4.0 * (c["n_src"] * c["c_in"] + c["n_rows"] * c["c_out"]
                       + c["K"] * c["c_in"] * c["c_out"]) + 8.0 * c["p"])
            return (2.0 * c["p"] * c["c_in"] * c["c_out"],
                    4.0 * (c["n_src"] * c["c_in"] + c["p"] * c["c_out"]
                           + c["K"] * c["c_in"] * c["c_out"]) + 8.0 * c["p"],
                    (c["c_in"], c["c_out"], c["K"], c["p"]), fam)

        def reduce_cost(c):
            return (0.0, 4.0 * c["c_out"] * (c["p"] + c["n_rows"]) + 4.0 * c["K"] * c["n_rows"],
                    (c["c_in"], c["c_out"], c["K"], c["p"]),
                    "row_reduce_kernel (ordered sum of product rows: fwd+dgrad)")

        hook_pr("pv2_spconv_products", products_cost)
        hook_pr("pv2_spconv_reduce_rows", reduce_cost)

        def vol_bytes(a):
            return 4.0 * a[1] * a[2] * a[3] * a[4] * a[5]

        def field_fwd_cost(*a):
            n = a[10] * a[11]
            gather = n * 8.0 * (C_ + F_) * 4
            return (n * 2.0 * (2 * H_ * F_ + G_ * H_ + F_ * H_),
                    n * 4.0 * (NV_ + F_ + 2 * H_ + F_ + 2) + min(vol_bytes(a), n * 8.0 * C_ * 4),
                    gather)

        def field_bwd_cost(*a):
            n = a[10] * a[11]
            wrote = n * 4.0 * (C_ + 4 + 2 * H_ + H_ + F_ + 68 + 4)
            read = n * 4.0 * (NV_ + H_ + F_ + C_ + 8)
            scatter = n * 8.0 * C_ * 4          # read-modify-write of the volume gradient
            return (n * 2.0 * (F_ * H_ + G_ * H_ + 2 * H_ * F_), wrote + read
                    + min(2 * vol_bytes(a), 2 * scatter), n * 8.0 * F_ * 4 + 2 * scatter)

        def coarse_cost(*a):
            n = a[10] * a[11]
            return (n * 2.0 * (2 * H_ * F_), min(vol_bytes(a) / 2, n * 8.0 * F_ * 4)
                    + a[10] * 4.0 * 3 * (a[11] + a[12] + 1), n * 8.0 * F_ * 4)

        # the same head with the final 1x1x1 convolution folded in (fused_head.FoldedVolume): the
        # field kernels read per-sample rows, a 32-channel gather / scatter pair does the sampling
        X_, XP_ = 32, 40

        def vol32_bytes(b, z, y, x):
            return 4.0 * b * z * y * x * X_

        def field_fwd_rows_cost(*a):
            n = a[6] * a[7]
            return (n * 2.0 * (2 * H_ * F_ + G_ * H_ + F_ * H_),
                    n * 4.0 * (NV_ + F_ + 2 * H_ + F_ + 2) + n * 4.0 * (C_ + 3 * F_), 0.0)

        def field_bwd_rows_cost(*a):
            n = a[5] * a[6]
            wrote = n * 4.0 * (C_ + 4 + 2 * H_ + H_ + F_ + 68 + 4)
            read = n * 4.0 * (NV_ + H_ + 3 * F_ + 8)
            return (n * 2.0 * (F_ * H_ + G_ * H_ + 2 * H_ * F_), wrote + read, 0.0)

        def fold_gather_cost(*a):
            n = a[9] * a[10]
            gather = n * 8.0 * X_ * 4
            return (0.0, n * 4.0 * 4 * XP_ + min(vol32_bytes(*a[1:5]), gather), gather)

        def fold_scatter_cost(*a):
            n = a[8] * a[9]
            scatter = n * 8.0 * X_ * 4
            return (0.0, n * 4.0 * (2 * X_ + 4) + min(2 * vol32_bytes(*a[0:4]), 2 * scatter), 2 * scatter)

        def coarse_folded_cost(*a):
            n = a[11] * a[12]
            return (n * 2.0 * (2 * H_ * F_ + F_ * XP_), min(vol32_bytes(*a[1:5]), n * 8.0 * X_ * 4)
                    + a[11] * 4.0 * 3 * (a[12] + a[13] + 1), n * 8.0 * X_ * 4)

        rows_split = os.environ.get("PV2_FIELD_ROWS_SPLIT") != "0"
        wrap_c("pv2_neus_field_forward_rows",
               "field_fwd_rows_kernel + field_pack_kernel (rows mode: folded final conv; bf16 pieces, "
               "transposed products)" if rows_split else "field_fwd_kernel (rows mode: folded final conv)",
               field_fwd_rows_cost)
        wrap_c("pv2_neus_field_backward_rows",
               "field_bwd_kernel + sums_reduce_kernel x 2 (rows mode: folded final conv)",
               field_bwd_rows_cost)
        wrap_c("pv2_neus_fold_gather", "fold_gather_kernel", fold_gather_cost)
        wrap_c("pv2_neus_fold_scatter", "fold_scatter_kernel", fold_scatter_cost)
        wrap_c("pv2_neus_coarse_sample_folded", "coarse_sample_kernel (folded final conv)",
               coarse_folded_cost)
        wrap_c("pv2_neus_field_forward", "field_fwd_kernel", field_fwd_cost)
        wrap_c("pv2_neus_field_backward", "field_bwd_kernel + volume_scatter_kernel", field_bwd_cost)
        wrap_c("pv2_neus_coarse_sample", "coarse_sample_kernel", coarse_cost)

        # dense 3x3x3 convolutions of the projection U-Net (csrc/dense_conv.hip): flops 2 * cells * 27 *
        # c_in * c_out; bytes: input + output grids once, the weights once (mask / addend reads on top)
        def dconv_fwd_cost(*a):
            b, z, y, xx, c_in, c_out, mode = a[1], a[2], a[3], a[4], a[5], a[7], a[8]
            cells_in = float(b * z * y * xx)
            cells_out = cells_in * (8.0 if mode == 1 else 0.125 if mode == 2 else 1.0)
            taps_cells = cells_in if mode == 1 else cells_out
            extra = (cells_in * c_in if a[11] else 0.0) + (cells_out * c_out if a[13] else 0.0)
            return (2.0 * taps_cells * 27 * c_in * c_out,
                    4.0 * (cells_in * c_in + cells_out * c_out + 27 * c_in * c_out + extra), 0.0)

        def dconv_wgrad_cost(*a):
            b, z, y, xx, c_x, c_g, mode = a[1], a[2], a[3], a[4], a[5], a[9], a[11]
            cells = float(b * z * y * xx)
            g_cells = cells * (8.0 if mode == 1 else 1.0)
            return (2.0 * cells * 27 * c_x * c_g,
                    4.0 * (cells * c_x + g_cells * c_g * (2.0 if a[10] else 1.0) + 27 * c_x * c_g), 0.0)

        orig_dconv = handle.pv2_dconv3_forward
        self._orig_c["pv2_dconv3_forward"] = orig_dconv
        fam_by_mode = {0: ("dconv_split_kernel" if SPLIT_ON else "dconv_kernel") +
                          " (dense 3x3x3 conv: fwd + grad-input)",
                       1: ("dconvT_split_kernel" if SPLIT_ON else "dconvT_kernel") +
                          " (dense transposed conv k3 s2: fwd)",
                       2: ("dconv_strided_split_kernel" if SPLIT_ON else "dconv_kernel") +
                          " (strided k3 s2: grad-input of the transposed conv)"}

        def timed_dconv(*a):
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record()
            rc = orig_dconv(*a)
            e_.record()
            c = dconv_fwd_cost(*a)
            fam = fam_by_mode[a[8]]
            timer.peaks.setdefault(fam, F32_MFMA_PEAK_TFLOPS)
            timer._add(fam, s_, e_, c[0], c[1], None)
            return rc

        handle.pv2_dconv3_forward = timed_dconv
        orig_dwgrad = handle.pv2_dconv3_backward_weight
        self._orig_c["pv2_dconv3_backward_weight"] = orig_dwgrad
        wfam_by_mode = {0: ("dconv_wgrad_split_kernel" if SPLIT_ON else "dconv_wgrad_kernel") +
                           " + dconv_wgrad_reduce_kernel (dense conv weight gradient, two-stage, deterministic)",
                        1: "dconv_wgrad_kernel + dconv_wgrad_reduce_kernel (transposed conv weight gradient, "
                           "two-stage, deterministic)"}

        def timed_dwgrad(*a):
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record()
            rc = orig_dwgrad(*a)
            e_.record()
            c = dconv_wgrad_cost(*a)
            timer._add(wfam_by_mode[a[11]], s_, e_, c[0], c[1], None)
            return rc

        handle.pv2_dconv3_backward_weight = timed_dwgrad
        wrap("spconv_forward", "spconv_fwd_kernel (fwd+dgrad)", conv_cost)
        wrap("spconv_grad_input", "spconv_fwd_kernel (fwd+dgrad)", dgrad_cost)
        wrap("spconv_backward_weight", "spconv_wgrad_kernel", wgrad_cost)
        wrap("spconv16_forward", "spconv_os16_kernel (fwd+dgrad, 16-bit)", conv16_cost, BF16_MFMA_PEAK_TFLOPS)
        wrap("spconv16_backward_weight", "spconv_wgrad16_kernel (16-bit)", wgrad16_cost,
             BF16_MFMA_PEAK_TFLOPS)
        wrap("trilinear_forward", "tri_fwd_kernel", tri_cost_factory(1))
        wrap("trilinear_backward", "tri_bwd_kernel", tri_cost_factory(2))
        wrap("trilinear_backward_backward", "tri_bwdbwd_kernel", tri_cost_factory(3))

    def uninstall(self):
        import ponderv2_amd.kernels as K

        for name, orig in self._orig.items():
            setattr(K, name, orig)
        for name, orig in self._orig_c.items():
            setattr(self._handle, name, orig)

    def reset(self):
        self.records = {}
        self.l2_bytes = {}
        self.survey = {}

    def summary(self):
        torch.cuda.synchronize()
        out = []
        for fam, recs in self.records.items():
            ms = sum(self._ms(r) for r in recs)
            flops = sum(r[2] for r in recs)
            nbytes = sum(r[3] for r in recs)
            out.append(dict(kernel=fam, launches=len(recs), total_ms=ms,
                            avg_us=1e3 * ms / max(len(recs), 1),
                            tflops=flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
                            alg_gbs=nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0,
                            alg_flops_per_launch=flops / max(len(recs), 1),
                            alg_bytes_per_launch=nbytes / max(len(recs), 1),
                            mfma_peak_tflops=mfma_peak_of(fam, self.peaks.get(fam, F32_MFMA_PEAK_TFLOPS))))
            out[-1]["frac_of_mfma_peak"] = out[-1]["tflops"] / out[-1]["mfma_peak_tflops"]
            out[-1]["frac_of_hbm_peak"] = out[-1]["alg_gbs"] / HBM_PEAK_GBS
            if fam in getattr(self, "survey", {}):
                out[-1]["alg_bytes_survey_per_launch"] = self.survey[fam] / max(len(recs), 1)
            if fam in self.l2_bytes:
                out[-1]["l2_gather_gbs"] = self.l2_bytes[fam] / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
                out[-1]["frac_of_f32_mfma_peak"] = out[-1]["tflops"] / F32_MFMA_PEAK_TFLOPS
                # these kernels gather from a volume that was just written and sits in the 256 MiB
                # Infinity Cache / the L2s: bytes / time is a CACHE rate (it can exceed what HBM
                # delivers), so no HBM roofline fraction is quoted for them
                del out[-1]["frac_of_hbm_peak"]
                out[-1]["cache_resident_gbs"] = out[-1].pop("alg_gbs")
        return sorted(out, key=lambda r: -r["total_ms"])

    def shape_table(self, steps):
        """Per (kernel family, c_in, c_out, K, pairs): launches/step, us/launch, TFLOP/s."""
        torch.cuda.synchronize()
        rows = {}
        for fam, recs in self.records.items():
            for r in recs:
                if r[4] is None:
                    continue
                key = (fam,) + tuple(r[4])
                acc = rows.setdefault(key, [0, 0.0, 0.0])
                acc[0] += 1
                acc[1] += self._ms(r)
                acc[2] += r[2]
        lines = ["%-32s %5s %5s %4s %9s %7s %9s %8s %8s" % ("kernel", "c_in", "c_out", "K", "pairs",
                 "n/step", "us/launch", "TFLOP/s", "ms/step")]
        for key, (n, ms, fl) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
            lines.append("%-32s %5d %5d %4d %9d %7.1f %9.1f %8.2f %8.3f" % (
                key[0][:32], key[1], key[2], key[3], key[4], n / steps, 1e3 * ms / n,
                fl / (ms * 1e-3) / 1e12 if ms > 0 else 0, ms / steps))
        return "\n".join(lines)


def make_batch(rank, scenes, views, device, voxelize=True):
    from ponderv2_amd.ponder.datasets import collate_fn, make_scene

    samples = [make_scene(1000 * rank + i, num_views=views, image_hw=(480, 640), voxelize=voxelize)
               for i in range(scenes)]
    batch = collate_fn(samples)
    return {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}


def clone_batch(batch):
    return {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}


def cpu_baseline(args):
    """SURVEY 8(d) / BASELINE.md section 3: the same model code on the host cores through the oracle's
    CPU kernels (kind "port"; spconv itself is absent, so there is no runnable reference CPU backbone,
    and the GPU box has no reference checkout).  Round 5:
      * THREAD SWEEP {8, 16, 32, 64, all}: the per-offset GEMMs of the oracle are tiny, and
        ``set_num_threads(cpu_count)`` on a 128-core box oversubscribes them (11.0 s for a forward that
        takes 3.4 s on 8 cores); the SparseUNet forward is timed per thread count (1 warm-up + 2 timed)
        and everything below runs at the best one, which is what ``cores`` reports;
      * 2 warm-up + 5 timed iterations, median, as BASELINE.md section 3 says;
      * the train step of configs[0] (1 scene, 20 000 voxels, 128 rays: the reference's own
        CPU-runnable case) = ``value``; the SparseUNet forward ALSO on the benched 46 842-voxel batch of
        configs[1] (``sparse_unet_forward_bench_s``), next to the GPU's number for the same batch."""
    import statistics

    from oracle import cpu_backend, rulebook as orb
    from ponderv2_amd.ponder.datasets import collate_fn, make_scene
    from ponderv2_amd.ponder.models import build_model
    from ponderv2_amd.ponder.utils.config import ConfigDict

    def timed(fn, warm, n, budget_s):
        """median of up to ``n`` timed calls after ``warm`` warm-ups, cut short (never below one timed call)
        when the section has used ``budget_s`` - the default bench run must stay within minutes whatever
        the host is"""
        start = time.perf_counter()
        for _ in range(warm):
            fn()
            if time.perf_counter() - start > budget_s / 2:
                break
        out = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            out.append(time.perf_counter() - t0)
            if time.perf_counter() - start > budget_s:
                break
        return statistics.median(out), len(out)

    ncpu = os.cpu_count() or 1
    threads_before = torch.get_num_threads()
    torch.manual_seed(0)
    try:
        with cpu_backend.installed():
            model = build_model(ConfigDict(model_cfg(64))).train()
            opt = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.9, nesterov=True,
                                  weight_decay=1e-4)
            batch = collate_fn([make_scene(0, num_views=2, image_hw=(480, 640), n_voxels=20000)])

            def backbone_fwd(b=batch):
                with torch.no_grad():
                    model.backbone(clone_batch(b))

            sweep = {}
            for nt in sorted({t for t in (8, 16, 32, 64, ncpu) if t <= ncpu}):
                torch.set_num_threads(nt)
                sweep[nt] = timed(backbone_fwd, 1, 1, 12.0)[0]
                if sweep[nt] > 1.5 * min(sweep.values()):
                    break      # past the knee: more threads only oversubscribe the per-offset GEMMs
            best = min(sweep, key=sweep.get)
            torch.set_num_threads(best)
            t_backbone, n_backbone = timed(backbone_fwd, 2, 5, 20.0)
            fwd_times = []

            def step():
                t0 = time.perf_counter()
                out = model(clone_batch(batch))
                fwd_times.append(time.perf_counter() - t0)
                opt.zero_grad()
                out["loss"].backward()
                opt.step()

            t_step, n_step = timed(step, 2, 5, 45.0)
            t_fwd = statistics.median(fwd_times[-n_step:])
            # the benched batch (configs[1]: 2 scenes, 46 842 voxels) through the same backbone
            bench_batch = collate_fn([make_scene(1000 * 0 + i, num_views=2, image_hw=(480, 640))
                                      for i in range(2)])
            n_bench = int(bench_batch["offset"][-1])
            t_backbone_bench, n_bench_it = timed(lambda: backbone_fwd(bench_batch), 2, 5, 30.0)
        coords = torch.cat([torch.zeros(len(batch["grid_coord"]), 1, dtype=torch.long),
                            batch["grid_coord"]], 1).int().numpy()
        t_rb = timed(lambda: orb.subm_rulebook(coords, 3), 0, 3, 5.0)[0]
    finally:
        torch.set_num_threads(threads_before)
    return dict(value=1.0 / t_step, unit="scenes/s", cores=best, kind="port",
                sample="configs[0]: 1 scene (20000 voxels, 2 views x 64 = 128 rays), train step "
                       "fwd+bwd+SGD, fp32, product model code on oracle CPU kernels; thread sweep "
                       f"{ {k: round(v, 2) for k, v in sweep.items()} } s per SparseUNet forward -> {best} "
                       f"threads of {ncpu}; 2 warm-up + median of up to 5 within a time budget (SparseUNet "
                       f"forward alone {t_backbone:.2f} s [{n_backbone} timed], on the benched {n_bench}-voxel "
                       f"batch {t_backbone_bench:.2f} s [{n_bench_it}]; model forward {t_fwd:.2f} s, step "
                       f"{t_step:.2f} s [{n_step}], level-0 k3 rulebook build {t_rb:.2f} s)",
                rays_per_s=128.0 / t_step, sparse_unet_forward_s=t_backbone,
                sparse_unet_forward_bench_s=t_backbone_bench, bench_batch_voxels=n_bench,
                thread_sweep_s={str(k): v for k, v in sweep.items()}, host_cores=ncpu,
                forward_s=t_fwd, step_s=t_step, rulebook_build_s=t_rb)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)      # does not return
    if args.launch_check:
        return launch_check(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and int(os.environ.get("RANK", "0")) == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); reporting "
              f"n_gpus={world}", file=sys.stderr)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); "
                         "there is no CPU fallback for the measured path")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # PV2_BENCH_FORCE_DIST=1: take the multi-process path (RCCL process group, DDP wrapper, barriers,
    # max-over-ranks timing) also with a single rank - the way to exercise it on a 1-GPU box
    dist_on = world > 1 or os.environ.get("PV2_BENCH_FORCE_DIST") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=device)  # RCCL

    from ponderv2_amd import fused_head
    from ponderv2_amd.ponder.models import build_model
    from ponderv2_amd.ponder.utils.config import ConfigDict

    torch.manual_seed(0)
    # MIOpen: "benchmark" = run the solver search now; otherwise immediate mode, which reads the
    # find-db shipped in miopen_cache/ (written by an earlier search on this GPU / MIOpen version)
    torch.backends.cudnn.benchmark = os.environ.get("PV2_MIOPEN_SEARCH", "0") == "1"
    outdoor = args.workload == "outdoor"
    ppt = args.workload == "ppt"
    cfg = model_cfg(args.rays_per_view, args.dense_dtype, args.workload, args.config)
    full = load_config(args.workload, args.config)
    model = build_model(ConfigDict(cfg)).to(device).train()
    step_model = model
    gsync = None
    if dist_on and args.grad_sync == "flat":
        # one flat all-reduce after backward instead of DDP's per-parameter hooks (utils/grad_sync.py)
        from ponderv2_amd.ponder.utils.grad_sync import FlatGradSync

        # overlap: the sparse executor's parameter-gradient arena is all-reduced IN PLACE, slab by slab,
        # behind the events its backward records - while the units below still run (PV2_GSYNC_OVERLAP=0:
        # the round-4 form, everything after backward)
        gsync = FlatGradSync(model.parameters(), uniform_usage=not ppt,
                             overlap=os.environ.get("PV2_GSYNC_OVERLAP", "1") != "0" and not SKIP_SYNC).attach()
    elif dist_on:
        # find_unused_parameters: the ppt head is reported, never trained (quirk Q10), and the
        # multi-dataset model trains a different condition's BatchNorms per step.  static_graph
        # (PV2_DDP_MODE=static) is rejected by DDP for this model ("graph has changed", measured on
        # MI355X).  Gradients live in the buckets (no copy into them: -1.9 ms per step).
        ddp_mode = os.environ.get("PV2_DDP_MODE", "find_unused")
        from ponderv2_amd import sidestream
        sidestream.disable("DistributedDataParallel reads gradients during the backward pass")
        step_model = torch.nn.parallel.DistributedDataParallel(
            model, device_ids=[local_rank], broadcast_buffers=False, gradient_as_bucket_view=True,
            find_unused_parameters=ddp_mode == "find_unused", static_graph=ddp_mode == "static")
    # optimiser + scheduler from the config's own sections, stepped every iteration as the
    # reference's run_step does (engines/train.py:185-203); the lr follows the config's rule
    # lr = base * total_batch / config_batch
    from ponderv2_amd.ponder.utils.optimizer import build_optimizer, build_scheduler

    ocfg = full.optimizer.to_dict()
    ocfg["lr"] = ocfg["lr"] * (args.scenes_per_gpu * world) / float(full.batch_size)
    opt = build_optimizer(ocfg, model, full.get("param_dicts"))
    scfg = full.scheduler.to_dict()
    scfg.update(total_steps=100000)
    if "max_lr" in scfg:
        scfg["max_lr"] = ocfg["lr"]
    sched = build_scheduler(scfg, opt)
    if outdoor:
        batch = make_outdoor_batch(rank, args.scenes_per_gpu, args.rays_per_camera, device)
    else:
        batch = make_batch(rank, args.scenes_per_gpu, args.views, device, voxelize=not args.raw_points)
    n_vox = int(batch["offset"][-1])
    batches, counter = [batch], [0]
    if ppt:  # one resident batch per condition, visited in the loader's 4:2:1 order
        per_cond = make_ppt_batches(rank, args.scenes_per_gpu, args.views, device, cfg["conditions"],
                                    cfg["valid_index"])
        batches = [per_cond[k] for k in PPT_SCHEDULE]

    amp_dtype = {None: None, "bf16": torch.bfloat16, "fp16": torch.float16}[args.amp]
    if amp_dtype is not None:   # the 16-bit backbone runs output-stationary: group rows by offset mask
        import ponderv2_amd.kernels as K_amp
        K_amp.MASK_ORDER = True
    scaler = torch.amp.GradScaler("cuda", enabled=amp_dtype == torch.float16)

    # The trainer's one-batch lookahead (engines/train.py staged_batches): the NEXT step's batch is
    # staged - here: cloned; in training: copied to the device and voxelised - and its sparse-conv
    # geometry launched on the side stream before THIS step is enqueued.  Every step still builds
    # one batch's geometry inside the timed region; it only no longer stalls the host.
    raw_model = model.module if hasattr(model, "module") else model
    lookahead = hasattr(raw_model, "prefetch") and not args.no_prefetch

    def stage(i):
        # on the input stream, as engines/train.py stages a loader batch: the batch's own short work
        # (here a clone; with --raw-points the device half of the input pipeline: GridSample of the raw
        # points, same voxels and order as the host transform, tests/test_gpu_voxelize.py) and the
        # launch of its sparse-conv geometry must not queue behind the step the host has enqueued
        from ponderv2_amd.ponder.datasets.voxelize import device_grid_sample, input_stream

        with input_stream(device) as pipe:
            b = clone_batch(batches[i % len(batches)])
            if args.raw_points and "grid_coord" not in b:
                b = device_grid_sample(b, grid_size=0.02, hash_type="fnv")
            if lookahead:
                b = raw_model.prefetch(b)
            return pipe.adopt(b)

    staged = [stage(0)]

    def step():
        cur = staged.pop()
        counter[0] += 1
        staged.append(stage(counter[0]))
        with torch.autocast("cuda", dtype=amp_dtype or torch.bfloat16, enabled=amp_dtype is not None):
            out = step_model(cur)
        opt.zero_grad(set_to_none=True)
        if scaler.is_enabled():   # engines/train.py:185-196 of the reference
            scaler.scale(out["loss"]).backward()
            if gsync is not None:
                gsync.sync()
            scaler.step(opt)
            # the scheduler advances only when the optimizer step was not skipped - the reference's
            # get_scale() comparison (ponder/engines/train.py:191-195; one device read, as there)
            before = scaler.get_scale()
            scaler.update()
            if before <= scaler.get_scale():
                sched.step()
        else:
            out["loss"].backward()
            if gsync is not None and not SKIP_SYNC:
                gsync.sync()
            opt.step()
            sched.step()
        if args.print_losses and rank == 0:
            print("loss", {k: round(float(v.detach()), 5) for k, v in out.items()}, flush=True)
        return out

    def timed_pass(n_steps):
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t_cpu = 0.0
        for _ in range(n_steps):
            out = step()
        t_cpu = time.perf_counter() - t0       # host done enqueuing (device may still be busy)
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist_on:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t)
        return dt, t_cpu, out

    first_loss_t = None
    for _ in range(args.warmup):
        out = step()
        if first_loss_t is None:
            first_loss_t = out["loss"].detach().clone()
    # pass 1: the measurement (no instrumentation inside the timed region)
    seg0 = torch.cuda.memory_stats(device).get("segment.all.allocated", 0)
    elapsed, host_enqueue, out = timed_pass(args.steps)
    new_segments = torch.cuda.memory_stats(device).get("segment.all.allocated", 0) - seg0
    loss = float(out["loss"].detach())
    first_loss = float(first_loss_t) if first_loss_t is not None else loss
    # a training step that is fast but computes garbage is not a measurement: say so in the line
    loss_sane = bool(loss == loss and abs(loss) < 50.0 * max(abs(first_loss), 1.0))
    # pass 2: the SAME K steps again with HIP events around every hand-written kernel launch
    # (weight gradients back on the main stream for this pass: an event pair around a launch that
    # shares the GPU with another stream's kernels would time the sharing, not the kernel)
    from ponderv2_amd import sidestream
    side_state = sidestream.status()
    timer, elapsed_instr = None, None
    if not args.no_kernel_timing:
        timer = KernelTimer()
        timer.install()
        # (and conv + BatchNorm units taken apart: pv2_convbn_* launch several kernels per C call,
        # which events recorded from Python cannot separate; the same kernels run either way - the
        # row reduce without its statistics epilogue, the statistics as col_partials)
        import ponderv2_amd.kernels as K_
        was_on, was_fused = sidestream.ENABLED, K_.USE_CONVBN
        sidestream.ENABLED, K_.USE_CONVBN = False, False
        try:
            elapsed_instr, _, _ = timed_pass(args.steps)
        finally:
            sidestream.ENABLED, K_.USE_CONVBN = was_on, was_fused

    kernels = timer.summary() if timer else []
    if timer:
        if args.kernel_table and rank == 0:
            os.makedirs(os.path.dirname(args.kernel_table) or ".", exist_ok=True)
            with open(args.kernel_table, "w") as f:
                f.write(timer.shape_table(args.steps) + "\n")
        timer.uninstall()
    result = None
    if rank == 0:
        scenes = args.scenes_per_gpu * world * args.steps
        rays_per_scene = args.views * args.rays_per_view
        if outdoor:
            rays_per_scene = int(batch["ray_offset"][-1]) / args.scenes_per_gpu
        value = scenes / elapsed
        result = {
            "metric": ("pretrain scenes/sec (+ rendered rays/sec), SpUNet-v1m1 nuScenes-shaped"
                       if outdoor else
                       "pretrain scenes/sec (+ rendered rays/sec), SpUNet-v1m3 PDNorm multi-dataset"
                       if ppt else
                       "pretrain scenes/sec (+ rendered rays/sec), SpUNet-v1m1 ScanNet-shaped"),
            "value": value, "unit": "scenes/s", "rays_per_s": value * rays_per_scene,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            # device allocations (hipMalloc, each a device-wide stall) the caching allocator still made INSIDE the
            # timed region: > 0 means the warm-up was too short for the allocator to have seen the batch sizes
            "allocator_segments_in_timed_region": int(new_segments),
            "ms_per_step": 1e3 * elapsed / args.steps,
            "host_enqueue_ms_per_step": 1e3 * host_enqueue / args.steps,
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": (f"{args.amp} autocast entered around the whole model (reference enable_amp=True): "
                      f"sparse backbone on the {args.amp} MFMA kernels through the native executor (fp32 "
                      "accumulation, statistics and master weights); dense UNet3D products on the leading "
                      "bf16 piece of each operand, fp32 sums and results (csrc/dense_conv.hip one-term mode); "
                      "ray march and losses f32 on the hand-written kernels (above the reference's 16-bit "
                      "precision there)"
                      if args.amp else
                      "f32" if args.dense_dtype == "float32" else
                      f"f32 sparse conv + render head; {args.dense_dtype} autocast for the dense "
                      "UNet3D convs (reference enable_amp=True)"),
            "data": "synthetic",
            "config": {"workload": (("configs[4]: PonderV2-outdoor nuScenes pretrain, SpUNet-v1m1, "
                                     f"bs={args.scenes_per_gpu}/GPU, {rays_per_scene:.0f} rays/scene"
                                     " (6 cameras), mask 0.8, train step fwd+bwd+AdamW") if outdoor
                                    else
                                    ("configs[3]: multi-dataset indoor pretrain (Structured3D/ScanNet/"
                                     "S3DIS conditions 4:2:1), SpUNet-v1m3 PDNorm, "
                                     f"bs={args.scenes_per_gpu}/GPU, {rays_per_scene} rays/scene, "
                                     "train step fwd+bwd+SGD") if ppt
                                    else
                                    ("configs[1]: PonderV2-indoor ScanNet pretrain, SpUNet-v1m1, "
                                     f"bs={args.scenes_per_gpu}/GPU, {rays_per_scene} rays/scene, "
                                     "train step fwd+bwd+SGD")),
                       "scenes_per_gpu": args.scenes_per_gpu, "rays_per_scene": rays_per_scene,
                       ("raw_points_per_gpu" if args.raw_points else "voxels_per_gpu"): n_vox,
                       "input_pipeline": ("device voxelisation (GridSample on the GPU) + ray pixel choice "
                                          "inside the timed step" if args.raw_points else
                                          "voxelised batches resident in HBM; ray pixel choice inside the step"),
                       "parallelism": f"dp{world}"},
            "first_loss": first_loss, "final_loss": loss, "loss_sane": loss_sane,
            "backward_side_stream": side_state,
            "sparse_backbone": _backbone_route(),
            "optimizer": "%s%s, %d group updates on cached lists (utils/optimizer.py)" % (
                type(opt).__name__, " (fused)" if opt.defaults.get("fused") else "",
                getattr(opt, "_pv2_lean_steps", 0)),
            "render_head": (("fused ray-march kernels (csrc/raymarch_fused.hip)"
                             + (", UNet3D's final 1x1x1 convolution folded in per sample"
                                if fused_head.FOLD_ENABLED else ""))
                            if fused_head.ENABLED and not outdoor else "modular (torch ops + kernels)"),
        }
        if kernels:
            dom = kernels[0]
            if dom["alg_flops_per_launch"] > 0:
                result["roofline"] = {"kernel": dom["kernel"], "bound": "mfma",
                                      "achieved": dom["tflops"], "peak": dom["mfma_peak_tflops"],
                                      "unit": "TFLOP/s", "frac": dom["tflops"] / dom["mfma_peak_tflops"],
                                      "peak_note": (
                                          "algorithmic (fp32-equivalent) TFLOP/s; this kernel computes every fp32 "
                                          "product as six v_mfma_f32_32x32x16_bf16 over three exact bf16 pieces per "
                                          "operand (csrc/mfma_split.h): peak = dense bf16 MFMA peak / 6; against the "
                                          "fp32 MFMA peak (%.1f) the same rate is %.3f" % (
                                              F32_MFMA_PEAK_TFLOPS, dom["tflops"] / F32_MFMA_PEAK_TFLOPS)
                                          if dom["mfma_peak_tflops"] == SPLIT_MFMA_PEAK_TFLOPS else
                                          "dense MFMA peak of the kernel's operand type"),
                                      "frac_vs_f32_mfma_peak": dom["tflops"] / F32_MFMA_PEAK_TFLOPS,
                                      "hbm_frac_of_alg_bytes": dom.get("frac_of_hbm_peak"),
                                      "traffic": pmc_traffic(dom["kernel"].split(" (")[0].rstrip(">"))[0],
                                      "traffic_source": pmc_traffic(dom["kernel"].split(" (")[0].rstrip(">"))[1],
                                      "traffic_note": "HBM-side bytes per launch (2 x FETCH_SIZE + "
                                                      "WRITE_SIZE) of this kernel's instantiations in "
                                                      "the PMC passes; algorithmic bytes per launch = "
                                                      "alg_bytes_per_launch",
                                      "alg_bytes_per_launch": dom["alg_bytes_per_launch"],
                                      "alg_bytes_survey": dom.get("alg_bytes_survey_per_launch"),
                                      "alg_bytes_survey_note": "SURVEY 8(d)'s bytes of the conv per launch: "
                                      "4(N_in C_in + N_out C_out + K C_in C_out) + 8P; alg_bytes_per_launch "
                                      "adds the product rows this two-stage form writes (the ordered row reduce "
                                      "that reads them back is its own entry in `kernels`)",
                                      "alg_flops_per_launch": dom["alg_flops_per_launch"],
                                      "avg_launch_us": dom["avg_us"], "launches": dom["launches"]}
            else:
                gbs = dom.get("alg_gbs", dom.get("cache_resident_gbs", 0.0))
                result["roofline"] = {"kernel": dom["kernel"], "bound": "hbm",
                                      "achieved": gbs, "peak": HBM_PEAK_GBS,
                                      "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                      "traffic": None, "avg_launch_us": dom["avg_us"],
                                      "launches": dom["launches"]}
            result["kernels"] = [{k: (round(v, 4) if isinstance(v, float) else v)
                                  for k, v in r.items()} for r in kernels]
            #  what `kernels` / `roofline` / `handwritten_kernel_ms_per_step` describe
            result["kernels_note"] = (
                "PASS-2 figures: after the timed pass the same K steps run again with HIP events around every "
                "hand-written launch, weight gradients back on the training stream and the conv + BatchNorm "
                "units of the sparse backbone taken apart (modular walk instead of the native executor: the "
                "same kernels, one C call each) - clean per-launch durations, which agree with rocprofv3's "
                "single-stream kernel stats; NOT what those kernels cost inside the timed two-stream schedule "
                "of `ms_per_step`, where they share the GPU with the weight-gradient stream")
            result["handwritten_kernel_ms_per_step"] = sum(r["total_ms"] for r in kernels) / args.steps
            result["ms_per_step_with_event_instrumentation"] = 1e3 * elapsed_instr / args.steps
        if world == 1 and not args.no_cpu_baseline and args.workload == "indoor":
            result["cpu_baseline"] = cpu_baseline(args)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
