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

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s


def pmc_traffic_bytes(kernel_key="spconv_fwd_lds_kernel<4>"):
    """HBM-side bytes per launch of the dominant kernel from the rocprofv3 PMC passes of THIS command
    (profiles/r01_pmc_fetch_write_per_kernel.json: separate --pmc FETCH_SIZE / WRITE_SIZE runs, KB
    per launch).  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 16-B/lane
    streaming reads.  None when the file is absent."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_fetch_write_per_kernel.json")
    try:
        with open(path) as f:
            rec = json.load(f)[kernel_key]
        return 1024.0 * (2.0 * rec["fetch_kb_per_launch"] + rec["write_kb_per_launch"])
    except Exception:
        return None


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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--dense-dtype", default="float32", choices=["bfloat16", "float16", "float32"],
                    help="dtype of the dense UNet3D projection (MIOpen).  float32 (default) = the "
                         "parity configuration, everything on the path in fp32; bfloat16 runs only "
                         "those dense convs under autocast (the reference config trains with "
                         "enable_amp=True)")
    ap.add_argument("--print-losses", action="store_true")
    ap.add_argument("--kernel-table", default=None, help="write a per-layer-shape kernel table here")
    args = ap.parse_args()
    if args.scenes_per_gpu is None:
        args.scenes_per_gpu = 4 if args.workload == "outdoor" else 2
    return args


def model_cfg(rays_per_view, dense_dtype="float32"):
    import golden_cases as gc  # the ScanNet model section, restated (reference tree absent here)

    backbone = dict(type="SpUNet-v1m1", in_channels=6, num_classes=0,
                    channels=(32, 64, 128, 256, 256, 128, 96, 96), layers=(2, 3, 4, 6, 2, 2, 2, 2))
    cfg = gc.indoor_model_cfg(backbone, grid_shape=(128, 128, 32), ray_nsample=rays_per_view)
    cfg["proj_autocast"] = None if dense_dtype == "float32" else dense_dtype
    return cfg


PPT_CONDITIONS = ("Structured3D", "ScanNet", "S3DIS")
PPT_VALID = (tuple(range(25)), tuple(range(20)), tuple(range(13)))  # class counts of the reference
PPT_SCHEDULE = (0, 0, 0, 0, 1, 1, 2)                                 # sampling ratio 4:2:1


def ppt_model_cfg(rays_per_view, dense_dtype="float32"):
    """configs/scannet/pretrain-ponder-ppt-v1m1-0-sc-s3-st-spunet.py:22-228, restated."""
    cfg = model_cfg(rays_per_view, dense_dtype)
    cfg["backbone"] = dict(type="SpUNet-v1m3", in_channels=6, num_classes=0, base_channels=32,
                           context_channels=256, channels=(32, 64, 128, 256, 256, 128, 96, 96),
                           layers=(2, 3, 4, 6, 2, 2, 2, 2), cls_mode=False,
                           conditions=("ScanNet", "S3DIS", "Structured3D"), zero_init=False,
                           norm_decouple=True, norm_adaptive=True, norm_affine=True)
    names = tuple(f"class {i}" for i in range(36))
    cfg.update(conditions=PPT_CONDITIONS, class_name=names, valid_index=PPT_VALID)
    return cfg


def make_ppt_batches(rank, scenes, views, device):
    from ponderv2_amd.ponder.datasets import collate_fn, make_scene

    out = []
    for k, cond in enumerate(PPT_CONDITIONS):
        samples = [make_scene(1000 * rank + 100 * k + i, num_views=views, image_hw=(480, 640),
                              condition=cond, num_classes=len(PPT_VALID[k])) for i in range(scenes)]
        batch = collate_fn(samples)
        out.append({k2: (v.to(device) if torch.is_tensor(v) else v) for k2, v in batch.items()})
    return out


def outdoor_model_cfg(dense_dtype="float32"):
    import golden_cases as gc  # the nuScenes model section, restated

    backbone = dict(type="SpUNet-v1m1", in_channels=4, num_classes=0,
                    channels=(32, 64, 128, 256, 256, 128, 96, 96), layers=(2, 3, 4, 6, 2, 2, 2, 2))
    cfg = gc.outdoor_model_cfg(backbone)
    cfg["proj_autocast"] = None if dense_dtype == "float32" else dense_dtype
    return cfg


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
        self._orig = {}

    def _add(self, fam, s, e, flops, nbytes, shape=None):
        self.records.setdefault(fam, []).append((s, e, flops, nbytes, shape))

    def install(self):
        import ponderv2_amd.kernels as K

        timer = self

        def wrap(name, fam, cost):
            orig = getattr(K, name)
            self._orig[name] = orig

            def fn(*a, **k):
                s = torch.cuda.Event(enable_timing=True)
                e = torch.cuda.Event(enable_timing=True)
                s.record()
                out = orig(*a, **k)
                e.record()
                c = cost(*a, **k)
                timer._add(fam, s, e, c[0], c[1], c[2] if len(c) > 2 else None)
                return out

            setattr(K, name, fn)

        def conv_cost(feats, w, rb, out=None):
            c_out, kk, c_in = w.shape
            p = rb.n_pairs
            return (2.0 * p * c_in * c_out,
                    4.0 * (rb.n_in * c_in + rb.n_out * c_out + kk * c_in * c_out) + 8.0 * p,
                    (c_in, c_out, kk, p))

        def wgrad_cost(feats, gout, rb, c_out):
            c_in = feats.shape[1]
            p = rb.n_pairs
            return (2.0 * p * c_in * c_out,
                    4.0 * (rb.n_in * c_in + rb.n_out * c_out + rb.K * c_in * c_out) + 8.0 * p,
                    (c_in, c_out, rb.K, p))

        def tri_cost_factory(mult):
            def cost(*a, **k):
                grid = a[1] if mult == 1 else (a[2] if mult == 2 else a[3])
                inp = a[0] if mult == 1 else (a[1] if mult == 2 else a[2])
                pts = grid.numel() // 3
                c = inp.shape[1]
                return (0.0, pts * 8.0 * c * inp.element_size() * mult)
            return cost

        wrap("spconv_forward", "spconv_fwd_kernel (fwd+dgrad)", conv_cost)
        wrap("spconv_backward_weight", "spconv_wgrad_kernel", wgrad_cost)
        wrap("trilinear_forward", "tri_fwd_kernel", tri_cost_factory(1))
        wrap("trilinear_backward", "tri_bwd_kernel", tri_cost_factory(2))
        wrap("trilinear_backward_backward", "tri_bwdbwd_kernel", tri_cost_factory(3))

    def uninstall(self):
        import ponderv2_amd.kernels as K

        for name, orig in self._orig.items():
            setattr(K, name, orig)

    def reset(self):
        self.records = {}

    def summary(self):
        torch.cuda.synchronize()
        out = []
        for fam, recs in self.records.items():
            ms = sum(r[0].elapsed_time(r[1]) for r in recs)
            flops = sum(r[2] for r in recs)
            nbytes = sum(r[3] for r in recs)
            out.append(dict(kernel=fam, launches=len(recs), total_ms=ms,
                            avg_us=1e3 * ms / max(len(recs), 1),
                            tflops=flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
                            alg_gbs=nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0,
                            alg_flops_per_launch=flops / max(len(recs), 1),
                            alg_bytes_per_launch=nbytes / max(len(recs), 1)))
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
                acc[1] += r[0].elapsed_time(r[1])
                acc[2] += r[2]
        lines = ["%-32s %5s %5s %4s %9s %7s %9s %8s %8s" % ("kernel", "c_in", "c_out", "K", "pairs",
                 "n/step", "us/launch", "TFLOP/s", "ms/step")]
        for key, (n, ms, fl) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
            lines.append("%-32s %5d %5d %4d %9d %7.1f %9.1f %8.2f %8.3f" % (
                key[0][:32], key[1], key[2], key[3], key[4], n / steps, 1e3 * ms / n,
                fl / (ms * 1e-3) / 1e12 if ms > 0 else 0, ms / steps))
        return "\n".join(lines)


def make_batch(rank, scenes, views, device):
    from ponderv2_amd.ponder.datasets import collate_fn, make_scene

    samples = [make_scene(1000 * rank + i, num_views=views, image_hw=(480, 640))
               for i in range(scenes)]
    batch = collate_fn(samples)
    return {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}


def clone_batch(batch):
    return {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}


def cpu_baseline(args):
    """Same model code, oracle CPU kernels, host cores: one scene (20 000 voxels, 128 rays),
    one full training step (forward + backward + SGD), fp32."""
    from oracle import cpu_backend
    from ponderv2_amd.ponder.datasets import collate_fn, make_scene
    from ponderv2_amd.ponder.models import build_model
    from ponderv2_amd.ponder.utils.config import ConfigDict

    torch.manual_seed(0)
    with cpu_backend.installed():
        model = build_model(ConfigDict(model_cfg(64))).train()
        opt = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.9, nesterov=True,
                              weight_decay=1e-4)
        batch = collate_fn([make_scene(0, num_views=2, image_hw=(480, 640), n_voxels=20000)])
        tb = time.perf_counter()   # the SparseUNet forward on its own (north_star's CPU baseline item)
        with torch.no_grad():
            model.backbone(clone_batch(batch))
        t_backbone = time.perf_counter() - tb
        t0 = time.perf_counter()
        out = model(clone_batch(batch))
        t_fwd = time.perf_counter() - t0
        opt.zero_grad()
        out["loss"].backward()
        opt.step()
        t = time.perf_counter() - t0
    return dict(value=1.0 / t, unit="scenes/s", cores=torch.get_num_threads(), kind="port",
                sample="1 scene (20000 voxels, 2 views x 64 = 128 rays), ONE train step "
                       "fwd+bwd+SGD, fp32, product model code on oracle CPU kernels "
                       f"(SparseUNet forward alone {t_backbone:.2f} s, model forward {t_fwd:.2f} s, "
                       f"step {t:.2f} s), no warm-up",
                rays_per_s=128.0 / t, sparse_unet_forward_s=t_backbone, forward_s=t_fwd, step_s=t)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); "
                         "there is no CPU fallback for the measured path")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
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
    if outdoor:
        cfg = outdoor_model_cfg(args.dense_dtype)
    elif ppt:
        cfg = ppt_model_cfg(args.rays_per_view, args.dense_dtype)
    else:
        cfg = model_cfg(args.rays_per_view, args.dense_dtype)
    model = build_model(ConfigDict(cfg)).to(device).train()
    step_model = model
    if world > 1:
        step_model = torch.nn.parallel.DistributedDataParallel(
            model, device_ids=[local_rank], broadcast_buffers=False, find_unused_parameters=True)
    # optimiser + OneCycleLR exactly as the reference's run_step drives them every iteration
    # (engines/train.py:185-203): the schedule starts at max_lr / div_factor
    if outdoor:  # configs/nuscenes/pretrain-ponder-spunet-v1m1-0-base.py:95-104
        opt = torch.optim.AdamW(model.parameters(), lr=2e-4, weight_decay=0.01)
        sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=2e-4, total_steps=100000,
                                                    pct_start=0.4, anneal_strategy="cos",
                                                    div_factor=10.0, final_div_factor=100.0)
        batch = make_outdoor_batch(rank, args.scenes_per_gpu, args.rays_per_camera, device)
    else:       # configs/scannet/pretrain-ponder-spunet-v1m1-0-base.py:156-170 (ppt: lr 1e-4 * bs / 8)
        lr = (0.0001 if ppt else 0.0005) * (args.scenes_per_gpu * world) / 8
        opt = torch.optim.SGD(model.parameters(), lr=lr, momentum=0.9, weight_decay=1e-4,
                              nesterov=True)
        sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=lr, total_steps=100000,
                                                    pct_start=0.05, anneal_strategy="cos",
                                                    div_factor=10.0, final_div_factor=10000.0)
        batch = make_batch(rank, args.scenes_per_gpu, args.views, device)
    n_vox = int(batch["offset"][-1])
    batches, counter = [batch], [0]
    if ppt:  # one resident batch per condition, visited in the loader's 4:2:1 order
        per_cond = make_ppt_batches(rank, args.scenes_per_gpu, args.views, device)
        batches = [per_cond[k] for k in PPT_SCHEDULE]

    def step():
        out = step_model(clone_batch(batches[counter[0] % len(batches)]))
        counter[0] += 1
        opt.zero_grad(set_to_none=True)
        out["loss"].backward()
        opt.step()
        sched.step()
        if args.print_losses and rank == 0:
            print("loss", {k: round(float(v.detach()), 5) for k, v in out.items()}, flush=True)
        return out

    def timed_pass(n_steps):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t_cpu = 0.0
        for _ in range(n_steps):
            out = step()
        t_cpu = time.perf_counter() - t0       # host done enqueuing (device may still be busy)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
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
    elapsed, host_enqueue, out = timed_pass(args.steps)
    loss = float(out["loss"].detach())
    first_loss = float(first_loss_t) if first_loss_t is not None else loss
    # a training step that is fast but computes garbage is not a measurement: say so in the line
    loss_sane = bool(loss == loss and abs(loss) < 50.0 * max(abs(first_loss), 1.0))
    # pass 2: the SAME K steps again with HIP events around every hand-written kernel launch
    timer, elapsed_instr = None, None
    if not args.no_kernel_timing:
        timer = KernelTimer()
        timer.install()
        elapsed_instr, _, _ = timed_pass(args.steps)

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
            "ms_per_step": 1e3 * elapsed / args.steps,
            "host_enqueue_ms_per_step": 1e3 * host_enqueue / args.steps,
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32" if args.dense_dtype == "float32" else
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
                       "voxels_per_gpu": n_vox, "parallelism": f"dp{world}"},
            "first_loss": first_loss, "final_loss": loss, "loss_sane": loss_sane,
            "render_head": ("fused ray-march kernels (csrc/raymarch_fused.hip)"
                            if fused_head.ENABLED and not outdoor else "modular (torch ops + kernels)"),
        }
        if kernels:
            dom = kernels[0]
            if dom["alg_flops_per_launch"] > 0:
                result["roofline"] = {"kernel": dom["kernel"], "bound": "mfma",
                                      "achieved": dom["tflops"], "peak": F32_MFMA_PEAK_TFLOPS,
                                      "unit": "TFLOP/s", "frac": dom["tflops"] / F32_MFMA_PEAK_TFLOPS,
                                      "traffic": pmc_traffic_bytes(),
                                      "traffic_note": "bytes/launch of spconv_fwd_lds_kernel<4> "
                                                      "(72 of the family's 117 launches/step) from "
                                                      "separate rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE "
                                                      "passes (profiles/); algorithmic bytes/launch = "
                                                      "alg_bytes_per_launch",
                                      "alg_bytes_per_launch": dom["alg_bytes_per_launch"],
                                      "alg_flops_per_launch": dom["alg_flops_per_launch"],
                                      "avg_launch_us": dom["avg_us"], "launches": dom["launches"]}
            else:
                result["roofline"] = {"kernel": dom["kernel"], "bound": "hbm",
                                      "achieved": dom["alg_gbs"], "peak": HBM_PEAK_GBS,
                                      "unit": "GB/s", "frac": dom["alg_gbs"] / HBM_PEAK_GBS,
                                      "traffic": None, "avg_launch_us": dom["avg_us"],
                                      "launches": dom["launches"]}
            result["kernels"] = [{k: (round(v, 4) if isinstance(v, float) else v)
                                  for k, v in r.items()} for r in kernels]
            result["handwritten_kernel_ms_per_step"] = sum(r["total_ms"] for r in kernels) / args.steps
            result["ms_per_step_with_event_instrumentation"] = 1e3 * elapsed_instr / args.steps
        if world == 1 and not args.no_cpu_baseline and args.workload == "indoor":
            result["cpu_baseline"] = cpu_baseline(args)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
