"""Shared drivers for the golden-vector tests: build the PRODUCT model, give it the fixture's
weights / random draws, run it on `device`, and compare with what the reference produced
(fixtures written by oracle/make_golden.py)."""
import os

import numpy as np
import torch

from oracle.detweights import fill_deterministic, formula_tensor, real_init_modulation

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SMALL_BACKBONE = dict(type="SpUNet-v1m1", in_channels=6, num_classes=0, base_channels=16,
                      channels=(16, 32, 48, 64, 64, 48, 32, 32), layers=(1, 1, 1, 1, 1, 1, 1, 1))

# model section of configs/scannet/pretrain-ponder-spunet-v1m1-0-base.py (reference :16-152),
# restated here because the reference tree is not available on the GPU box.
RENDERER = dict(
    type="NeuSModel",
    field=dict(type="SDFField",
               sdf_decoder=dict(in_dim=64, out_dim=65, hidden_size=128, n_blocks=1, pos_enc=False,
                                points_factor=0.0),
               rgb_decoder=dict(in_dim=134, out_dim=3, hidden_size=128, n_blocks=0, pos_enc=False,
                                points_factor=0.0),
               semantic_decoder=dict(in_dim=131, out_dim=512, hidden_size=128, n_blocks=0,
                                     points_factor=0.0),
               beta_init=0.3, use_gradient=True, volume_type="default", padding_mode="zeros",
               share_volume=False, norm_pts=True, norm_padding=0.1),
    collider=dict(type="AABBBoxCollider", near_plane=0.01,
                  bbox=[-0.55, -0.55, -0.55, 0.55, 0.55, 0.55]),
    sampler=dict(type="NeuSSampler", initial_sampler="UniformSampler", num_samples=96,
                 num_samples_importance=36, num_upsample_steps=1, train_stratified=True,
                 single_jitter=False),
    loss=dict(sensor_depth_truncation=0.05, temperature=0.01,
              weights=dict(eikonal_loss=0.01, free_space_loss=1.0, sdf_loss=10.0, depth_loss=1.0,
                           rgb_loss=10.0, semantic_loss=0.1)))

CLASS_NAMES = ("wall", "floor", "cabinet", "bed", "chair", "sofa", "table", "door", "window",
               "bookshelf", "picture", "counter", "desk", "curtain", "refridgerator",
               "shower curtain", "toilet", "sink", "bathtub", "otherfurniture")


def indoor_model_cfg(backbone, grid_shape=(128, 128, 32), ray_nsample=256):
    return dict(type="PonderIndoor-v2", backbone=backbone,
                projection=dict(type="UNet3D-v1m2", in_channels=96, out_channels=128),
                renderer=RENDERER, mask=None, grid_shape=grid_shape, grid_size=0.02,
                val_ray_split=10240, ray_nsample=ray_nsample, padding=0.1, pool_type="mean",
                render_semantic=True, conditions=("ScanNet",), template="a photo of a [x]",
                clip_model="ViT-B/16", class_name=CLASS_NAMES, valid_index=(tuple(range(20)),),
                ppt_loss_weight=1.0,
                ppt_criteria=[dict(type="CrossEntropyLoss", loss_weight=1.0, ignore_index=-1)])


def cos_err(a, b):
    """1 - cosine similarity (direction error of a gradient tensor)."""
    a = torch.as_tensor(np.asarray(a.detach().cpu() if torch.is_tensor(a) else a)).double().flatten()
    b = torch.as_tensor(np.asarray(b)).double().flatten()
    return 1.0 - float((a @ b) / (a.norm() * b.norm() + 1e-300))


def rel_err(a, b):
    a = torch.as_tensor(np.asarray(a.detach().cpu() if torch.is_tensor(a) else a)).double()
    b = torch.as_tensor(np.asarray(b)).double()
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-30)


class ReplayRand:
    """Stands in for torch.rand inside the samplers: returns the reference's recorded draws.  The
    reference renders scene by scene (draw shapes (R,97),(R,37) per scene); a batched renderer asks
    once for (B*R,97) then (B*R,37): recorded draws with the same trailing size are concatenated
    in recording order, so both call patterns see exactly the reference's numbers."""

    def __init__(self, draws, device):
        self.queues, self.device = {}, device
        for d in draws:
            d = torch.as_tensor(d)
            self.queues.setdefault(d.shape[-1], []).append(d)

    def __call__(self, shape, dtype=None, device=None):
        rows, width = shape
        q, got = self.queues[width], []
        while sum(t.shape[0] for t in got) < rows:
            got.append(q.pop(0))
        out = torch.cat(got, 0)
        assert tuple(out.shape) == tuple(shape), (out.shape, shape)
        return out.to(self.device)


def _gradient_errors(params, g):
    """errs / cos per recorded gradient of a backbone fixture: full tensors where the fixture holds them,
    projections on the seeded probes for the large ones (oracle/make_golden.py _compact_gradients)."""
    from oracle.detweights import GRAD_PROBES, grad_probe

    errs, cos = {}, {}
    for i, name in enumerate(g["grad_names"]):
        name = str(name)
        grad = params[name].grad
        if f"grad_{i}" in g.files:
            errs[name] = rel_err(grad, g[f"grad_{i}"])
            cos[name] = cos_err(grad, g[f"grad_{i}"])
        else:
            flat = grad.detach().double().reshape(-1)
            ref = g[f"gproj_{i}"]
            e2 = np.mean([(float(flat @ grad_probe(name, j, grad.shape).to(grad.device).double().reshape(-1))
                           - float(ref[j])) ** 2 for j in range(GRAD_PROBES)])
            errs[name] = float(e2 ** 0.5 / (float(g[f"gnorm_{i}"]) + 1e-30))   # |error| / |gradient|
    return errs, cos


def run_spunet(device, dtype, real_init=False):
    from ponderv2_amd.ponder.models import build_model
    from ponderv2_amd.ponder.utils.config import ConfigDict

    g = np.load(os.path.join(GOLDEN, "spunet_small_real_init.npz" if real_init else "spunet_small.npz"))
    coords = g["coords"]
    counts = np.bincount(coords[:, 0])
    torch.manual_seed(0)     # (real_init: the constructor's draws are the fixture's weights)
    model = build_model(ConfigDict(SMALL_BACKBONE)).to(dtype)
    if not real_init:
        fill_deterministic(model)
    model = model.to(device).train()
    n = len(coords)
    feat = formula_tensor("spunet.feat", (n, 6), 1.0).to(dtype).to(device).requires_grad_(True)
    out = model(dict(grid_coord=torch.from_numpy(coords[:, 1:].astype(np.int64)).to(device),
                     feat=feat, offset=torch.from_numpy(np.cumsum(counts)).long().to(device)))
    probe = formula_tensor("spunet.probe", tuple(out.shape), 1.0).to(dtype).to(device)
    (out * probe).sum().backward()
    params = dict(model.named_parameters())
    errs = {"out": rel_err(out, g["out"]), "dfeat": rel_err(feat.grad, g["dfeat"])}
    cos = {"dfeat": cos_err(feat.grad, g["dfeat"])}
    ge, gcos = _gradient_errors(params, g)
    errs.update(ge)
    cos.update(gcos)
    if dtype == torch.float64:
        return errs
    return errs, cos


def run_neus(device, return_values=False):
    from ponderv2_amd import fused_head as fhd
    from ponderv2_amd.ponder.models.ponder.render_utils import RayBundle, build_renderer
    from ponderv2_amd.ponder.utils.config import ConfigDict

    g = np.load(os.path.join(GOLDEN, "neus_head.npz"))
    renderer = build_renderer(ConfigDict(RENDERER))
    fill_deterministic(renderer)
    renderer = renderer.to(device).train()
    replay = ReplayRand([g["rand0"], g["rand1"]], device)
    renderer.sampler.initial_sampler.rand = replay
    renderer.sampler.pdf_sampler.rand = replay
    volume = formula_tensor("neus.volume", (128, 8, 16, 16), 0.6).to(device)
    if device.type == "cuda":
        volume = volume.unsqueeze(0).contiguous(memory_format=torch.channels_last_3d).squeeze(0)
    volume.requires_grad_(True)
    o = torch.from_numpy(g["origins"]).to(device)
    d = torch.from_numpy(g["directions"]).to(device)
    targets = {k: torch.from_numpy(g[f"tgt_{k}"]).to(device) for k in ("depth", "rgb", "semantic")}
    fused_calls = {"n": 0}
    orig_render = fhd.field_render

    def counted(*a):
        fused_calls["n"] += 1
        return orig_render(*a)

    fhd.field_render = counted
    try:
        out = renderer(RayBundle(origins=o, directions=d), [volume])
    finally:
        fhd.field_render = orig_render
    losses = renderer.get_loss(out, targets)
    sum(v for k, v in losses.items() if "loss" in k).backward()
    if return_values:
        vals = {"out_" + k: out[k].detach().cpu() for k in ("rgb", "semantic", "depth", "normal", "weights",
                                                              "sdf", "gradients", "z_vals")}
        vals.update({"loss_" + k: v.detach().cpu().reshape(1) for k, v in losses.items()})
        vals["dvolume"] = volume.grad.detach().cpu()
        vals.update({"grad_" + k: p.grad.detach().cpu() for k, p in renderer.named_parameters()
                     if p.grad is not None})
        return dict(values=vals, fused_calls=fused_calls["n"])
    errs = {}
    for k in ("rgb", "semantic", "depth", "normal", "weights", "sdf", "gradients", "z_vals"):
        errs["out_" + k] = rel_err(out[k], g["out_" + k])
    for name, val in zip(g["loss_names"], g["loss_values"]):
        errs["loss_" + str(name)] = abs(float(losses[str(name)]) - val) / (abs(val) + 1e-12)
    errs["dvolume"] = rel_err(volume.grad, g["dvolume"])
    params = dict(renderer.named_parameters())
    for i, name in enumerate(g["grad_names"]):
        errs["grad_" + str(name)] = rel_err(params[str(name)].grad, g[f"grad_{i}"])
    return errs


def small_indoor(device):
    """The miniature PonderIndoor of the ``ponder_indoor_small`` fixture (deterministic weights) and
    its two-scene batch, without the fixture's recorded random draws."""
    from ponderv2_amd.ponder.datasets import collate_fn, make_scene
    from ponderv2_amd.ponder.models import build_model
    from ponderv2_amd.ponder.utils.config import ConfigDict

    cfg = indoor_model_cfg(dict(SMALL_BACKBONE, channels=(16, 32, 48, 64, 64, 48, 32, 96)),
                           grid_shape=(32, 32, 8), ray_nsample=20)
    cfg["template"] = ("a", "b")  # any template list: the stub embeddings do not depend on it
    model = build_model(ConfigDict(cfg))
    fill_deterministic(model)
    model = model.to(device).train()
    kw = dict(n_raw=16000, num_views=2, image_hw=(48, 64))
    batch = collate_fn([make_scene(100, **kw), make_scene(101, **kw)])
    batch = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
    return model, batch


def run_ponder_indoor(device):
    g = np.load(os.path.join(GOLDEN, "ponder_indoor_small.npz"))
    model, batch = small_indoor(device)
    replay = ReplayRand([g[f"rand_{i}"] for i in range(int(g["rands"]))], device)
    model.renderer.sampler.initial_sampler.rand = replay
    model.renderer.sampler.pdf_sampler.rand = replay
    batch["ray_pixels"] = torch.from_numpy(g["ray_pixels"])
    out = model(batch)
    out["loss"].backward()
    errs = {}
    for name, val in zip(g["out_names"], g["out_values"]):
        errs[str(name)] = abs(float(out[str(name)].detach()) - val) / (abs(val) + 1e-12)
    params = dict(model.named_parameters())
    for i, name in enumerate(g["grad_names"]):
        errs["grad_" + str(name)] = rel_err(params[str(name)].grad, g[f"grad_{i}"])
    return errs


FULL_BACKBONE = dict(type="SpUNet-v1m1", in_channels=6, num_classes=0,
                     channels=(32, 64, 128, 256, 256, 128, 96, 96), layers=(2, 3, 4, 6, 2, 2, 2, 2))


def run_ponder_indoor_cfg1(device, with_float64=True):
    """BASELINE.json configs[1] (the bench workload) at full size: 2 scenes, 512 rays each."""
    return run_ponder_indoor_cfg0(device, scenes=2, rays_per_view=256, n_voxels=None,
                                  name="ponder_indoor_cfg1", with_float64=with_float64)


def run_ponder_ppt_full(device, condition_index, with_float64=True, real_init=False):
    """BASELINE.json configs[3] at FULL size, one batch of one condition: the shipped multi-dataset
    model (SpUNet-v1m3 PDNorm at full width and depth, 128x128x32 grid, UNet3D-v1m2, NeuS head, 512
    rays per scene) against the reference's own step (oracle/make_golden.py::ponder_ppt_full_case)."""
    from ponderv2_amd.ponder.datasets import collate_fn, make_scene

    cond = PPT_CONDITIONS[condition_index]
    cfg = indoor_model_cfg(dict(PDNORM_BACKBONE, base_channels=32, context_channels=256,
                                channels=FULL_BACKBONE["channels"], layers=FULL_BACKBONE["layers"]),
                           grid_shape=(128, 128, 32), ray_nsample=256)
    cfg.update(conditions=PPT_CONDITIONS, class_name=tuple(f"class {i}" for i in range(36)),
               valid_index=PPT_VALID, template=("a", "b"))
    kw = dict(num_views=2, image_hw=(480, 640), condition=cond, num_classes=len(PPT_VALID[condition_index]))
    batch = collate_fn([make_scene(700 + 10 * condition_index + i, **kw) for i in range(2)])
    return _run_indoor_full(device, cfg, batch,
                            "ponder_ppt_full_" + cond.lower() + ("_real_init" if real_init else ""),
                            with_float64, real_init=real_init)


def run_ponder_indoor_cfg0(device, scenes=1, rays_per_view=64, n_voxels=20000,
                           name="ponder_indoor_cfg0", with_float64=True):
    """BASELINE.json configs[0] at full size against the reference's own run of it
    (oracle/make_golden.py::ponder_indoor_cfg0_case): one scene, 20 000 voxels, 128 rays, the
    shipped backbone / grid / head.  Returns relative errors of every loss term, of the rendered
    RGB / depth / normal per ray, of the gradient probes, and the number of importance-sampling bin
    indices that differ from the reference's ``searchsorted`` result."""
    from ponderv2_amd.ponder.datasets import collate_fn, make_scene

    cfg = indoor_model_cfg(FULL_BACKBONE, grid_shape=(128, 128, 32), ray_nsample=rays_per_view)
    batch = collate_fn([make_scene(i, num_views=2, image_hw=(480, 640), n_voxels=n_voxels)
                        for i in range(scenes)])
    return _run_indoor_full(device, cfg, batch, name, with_float64)


def run_ponder_indoor_cfg1_real_init(device):
    """configs[1] at full size with the reference's REAL initialisation: the model is built under
    ``torch.manual_seed(0)`` exactly as the reference's was when the fixture was made (same
    construction order -> same draws; oracle/make_golden.py::ponder_indoor_cfg1_real_init_case)."""
    from ponderv2_amd.ponder.datasets import collate_fn, make_scene

    cfg = indoor_model_cfg(FULL_BACKBONE, grid_shape=(128, 128, 32), ray_nsample=256)
    batch = collate_fn([make_scene(i, num_views=2, image_hw=(480, 640), n_voxels=None) for i in range(2)])
    return _run_indoor_full(device, cfg, batch, "ponder_indoor_cfg1_real_init", True, real_init=True)


def _run_indoor_full(device, cfg, batch, name, with_float64=True, real_init=False):
    """One training step of a full-size PonderIndoor model against tests/golden/<name>.npz."""
    from ponderv2_amd import fused_head as fhd
    from ponderv2_amd.ponder.models import build_model
    from ponderv2_amd.ponder.utils.config import ConfigDict

    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    if real_init:
        torch.manual_seed(0)     # the reference's own initialisation, drawn from the same seed
    model = build_model(ConfigDict(cfg))
    if not real_init:
        fill_deterministic(model)
    model = model.to(device).train()
    replay = ReplayRand([g[f"rand_{i}"] for i in range(int(g["rands"]))], device)
    model.renderer.sampler.initial_sampler.rand = replay
    model.renderer.sampler.pdf_sampler.rand = replay
    assert int(batch["offset"][-1]) == int(g["n_voxels"])
    batch = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
    batch["ray_pixels"] = torch.from_numpy(g["ray_pixels"])
    rendered, capture = {"rgb": [], "depth": [], "normal": []}, {}
    orig = model.renderer.forward

    def render(*a, **k):
        out = orig(*a, **k)
        for key in rendered:   # one call for the batched render, one per scene otherwise
            rendered[key].append(out[key].detach())
        return out

    model.renderer.forward = render
    fhd.CAPTURE = capture
    try:
        out = model(batch)
    finally:
        fhd.CAPTURE = None
    out["loss"].backward()
    errs = {}
    for name, val in zip(g["out_names"], g["out_values"]):
        errs[str(name)] = abs(float(out[str(name)].detach()) - val) / (abs(val) + 1e-12)
    for key in ("rgb", "depth", "normal"):
        errs["render_" + key] = rel_err(torch.cat(rendered[key]), g["render_" + key])
    for key in ("rgb", "depth"):
        if "render64_" + key in g.files:   # the reference's FLOAT64 render: ours / the reference's fp32 one
            errs["render64_" + key] = rel_err(torch.cat(rendered[key]), g["render64_" + key])
            errs["ref32_render64_" + key] = rel_err(torch.from_numpy(g["render_" + key]), g["render64_" + key])
    params = dict(model.named_parameters())
    for i, name in enumerate(g["grad_names"]):
        errs["grad_" + str(name)] = rel_err(params[str(name)].grad, g[f"grad_{i}"])
    flips = -1
    if "idx" in capture:
        flips = int((capture["idx"].cpu().numpy() != g["pdf_bins"]).sum())
    if "g64_names" in g.files and with_float64:
        errs["float64"] = float64_gradient_errors(params, g)
    return errs, flips


def float64_gradient_errors(params, g):
    """Every parameter gradient of the step against the reference's FLOAT64 pass
    (oracle/make_golden.py::float64_gradient_record): the fixture holds each gradient's norm and
    GRAD_PROBES random projections; for e = g - g_ref and iid standard-normal probes
    E[(probe . e)^2] = |e|^2, so |e| is estimated by the root mean square of the projection
    differences (8 probes: +-25 % on a single tensor, far tighter on the global figures).

    Returns the global relative error ||g - g_ref|| / ||g_ref|| over all parameters and over the
    backbone alone, each next to the SAME figure for the reference's own fp32 pass (``ref32_*``:
    what fp32 arithmetic does to these gradients on the reference's side - the yardstick), and the
    worst single tensor among those whose norm is not negligible (parameters in front of a
    BatchNorm have an exactly-zero true gradient: pure rounding noise on either side)."""
    from oracle.detweights import GRAD_PROBES, grad_probe

    keys = ("all", "backbone")
    num, num32, den = ({k: 0.0 for k in keys} for _ in range(3))
    per_tensor = []
    have32 = "g32_proj" in g.files
    scale = float(np.sqrt((g["g64_norm"] ** 2).mean()))
    for t, (name, ref_norm, ref_proj) in enumerate(zip(g["g64_names"], g["g64_norm"], g["g64_proj"])):
        name = str(name)
        grad = params[name].grad
        assert grad is not None, name
        flat = grad.detach().double().reshape(-1)
        e2 = 0.0
        for i in range(GRAD_PROBES):
            probe = grad_probe(name, i, grad.shape).to(grad.device).double().reshape(-1)
            e2 += (float(flat @ probe) - float(ref_proj[i])) ** 2
        e2 /= GRAD_PROBES
        e2_32 = float(((g["g32_proj"][t] - ref_proj) ** 2).mean()) if have32 else 0.0
        for key in keys:
            if key == "all" or name.startswith("backbone."):
                num[key] += e2
                num32[key] += e2_32
                den[key] += float(ref_norm) ** 2
        if ref_norm > 1e-3 * scale:
            per_tensor.append((e2 ** 0.5 / float(ref_norm), e2_32 ** 0.5 / float(ref_norm), name))
    worst = max(per_tensor)
    return dict(global_rel=(num["all"] / den["all"]) ** 0.5,
                backbone_rel=(num["backbone"] / den["backbone"]) ** 0.5,
                ref32_global_rel=(num32["all"] / den["all"]) ** 0.5 if have32 else None,
                ref32_backbone_rel=(num32["backbone"] / den["backbone"]) ** 0.5 if have32 else None,
                worst_tensor_rel=worst[0], worst_tensor_ref32_rel=worst[1], worst_tensor=worst[2],
                tensors=len(g["g64_names"]), top=sorted(per_tensor, reverse=True)[:8])


def check_float64_gradients_tight(f64, tensor_floor=5e-3):
    """On the well-conditioned fixtures (the reference's real initialisation) the GPU's fp32 gradients
    must sit within 1e-3 of the float64 ones globally and over the backbone - or within TWICE what
    the reference's own fp32 pass manages on that batch, whichever is larger - and no single tensor may
    be off by more than twice its reference-fp32 figure (or ``tensor_floor``: 5e-3, the round-4 bar; a
    single tensor's figure is an estimate from 8 projections, +-25 %).  Measured on MI355X (round 6, ours /
    the reference's fp32, global | backbone): configs[1] 1.13e-3 / 1.15e-3 | 1.24e-3 / 1.25e-3; configs[3]
    Structured3D 2.1e-4 / 5.4e-4 | 2.4e-4 / 2.9e-4, ScanNet 1.8e-3 / 2.1e-3 | 2.0e-3 / 1.9e-3, S3DIS
    9.4e-4 / 6.2e-4 | 1.24e-3 / 6.5e-4; configs[4] 7.4e-5 / 2.8e-4 | 1.5e-4 / 5.4e-4.
    S3DIS is the one batch where this program lands further out than the reference's fp32 (one first-level
    BatchNorm bias at 6.4e-3: that fixture alone gets ``tensor_floor=1e-2``).  Round 6 looked for a cause in
    the column sums of the BatchNorm backward and found the figure to be a property of
    the rounding pattern, not of one kernel: with every statistic accumulated in DOUBLE per block (what
    ships) the five fixtures read as above; with fp32 accumulation S3DIS reads 4.0e-4 (below the reference)
    and configs[1] 2.0e-3 (1.75x above it) - the same code, the order of a few fp32 additions apart
    (profiles/r06_gradient_distance.txt)."""
    assert f64["tensors"] > 200, f64
    for key in ("global_rel", "backbone_rel"):
        ref = f64["ref32_" + key]
        assert ref is not None, "fixture without the reference's fp32 record"
        assert f64[key] <= max(2.0 * ref, 1e-3), f64
    assert f64["worst_tensor_rel"] <= max(2.0 * f64["worst_tensor_ref32_rel"], tensor_floor), f64


def check_float64_gradients(f64, closed_form=True):
    """The GPU's fp32 gradients are as close to the reference's float64 gradients as fp32 arithmetic
    lets ANY implementation be on these fixtures.  The yardstick is the reference's own fp32 pass
    against its own float64 pass, recorded in the fixtures: global relative error 1.2e-3 ... 1.1e-1
    depending on the batch (configs[0] 3.2e-2, configs[1] 8.3e-2, configs[3] 7.5e-2 / 7.2e-3 / 1.2e-3,
    configs[4] 1.1e-1): with the fixtures' closed-form weights the ~60 BatchNorm layers amplify a
    rounding error of the forward pass (a ReLU or max-pool decision at the last bit) into percents
    of the gradient, on either side, so the size of the error is a property of the batch, not of the
    implementation.  Hence: within 3x of the reference's own fp32 error, or below 15 % where the
    reference happened to land close (measured on MI355X: 7.8e-3 ... 1.1e-1)."""
    # (closed_form=False - any fixture made with the reference's REAL initialisation - has no business
    # here: those are held by check_float64_gradients_tight; the 15 % floor exists for the closed-form
    # weights only)
    assert closed_form, "real-initialisation fixtures use check_float64_gradients_tight"
    assert f64["tensors"] > 200, f64
    for key in ("global_rel", "backbone_rel"):
        ref = f64["ref32_" + key]
        assert ref is not None, "fixture without the reference's fp32 record"
        assert f64[key] <= max(3.0 * ref, 0.15), f64


# model section of configs/nuscenes/pretrain-ponder-spunet-v1m1-0-base.py (reference :20-92)
OUTDOOR_RENDERER = dict(
    type="NeuSModel",
    field=dict(type="SDFField", sdf_decoder=dict(in_dim=32, out_dim=17, hidden_size=16, n_blocks=5),
               beta_init=0.3, use_gradient=True, volume_type="default", padding_mode="zeros",
               share_volume=True),
    collider=dict(type="AABBBoxCollider", near_plane=0.01, bbox=[0.0, 0.0, 0.0, 1.0, 1.0, 1.0]),
    sampler=dict(type="NeuSSampler", initial_sampler="UniformSampler", num_samples=72,
                 num_samples_importance=24, num_upsample_steps=1, train_stratified=True,
                 single_jitter=False),
    loss=dict(sensor_depth_truncation=0.01, weights=dict(depth_loss=10.0)))

NUSCENES_CLASSES = ("barrier", "bicycle", "bus", "car", "construction vehicle", "motorcycle",
                    "pedestrian", "traffic cone", "trailer", "truck",
                    "path suitable or safe for driving", "other flat", "sidewalk", "terrain",
                    "man made", "vegetation")

OUTDOOR_SMALL = dict(scene_bbox=((-27.0, -27.0, -5.0, 27.0, 27.0, 3.0),), grid_shape=((45, 45, 5),),
                     grid_size=((1.2, 1.2, 1.6),))
OUTDOOR_SCENE_KW = dict(grid_size=0.1, point_nsample=24, n_azimuth=200,
                        point_cloud_range=(-27.0, -27.0, -5.0, 27.0, 27.0, 3.0))


def outdoor_model_cfg(backbone, **geometry):
    cfg = dict(type="PonderOutdoor-v2", mask=dict(ratio=0.8, size=8, channel=4), backbone=backbone,
               projection=dict(type="SimpleConv3D-v1m1", in_channels=96, out_channels=32),
               renderer=OUTDOOR_RENDERER,
               scene_bbox=((-54.0, -54.0, -5.0, 54.0, 54.0, 3.0),), grid_shape=((180, 180, 5),),
               grid_size=((0.6, 0.6, 1.6),), val_ray_split=8192, pool_type="mean",
               share_volume=True, render_semantic=False, conditions=("nuScenes",),
               template="[x]", clip_model="ViT-B/16", class_name=NUSCENES_CLASSES,
               valid_index=(tuple(range(16)),))
    cfg.update(geometry)
    return cfg


def run_ponder_outdoor(device):
    from ponderv2_amd.ponder.datasets import lidar_collate_fn, make_lidar_scene
    from ponderv2_amd.ponder.models import build_model
    from ponderv2_amd.ponder.utils.config import ConfigDict

    g = np.load(os.path.join(GOLDEN, "ponder_outdoor_small.npz"))
    cfg = outdoor_model_cfg(dict(SMALL_BACKBONE, in_channels=4,
                                 channels=(16, 32, 48, 64, 64, 48, 32, 96)), **OUTDOOR_SMALL)
    model = build_model(ConfigDict(cfg))
    fill_deterministic(model)
    model = model.to(device).train()
    replay = ReplayRand([g[f"rand_{i}"] for i in range(int(g["rands"]))], device)
    model.renderer.sampler.initial_sampler.rand = replay
    model.renderer.sampler.pdf_sampler.rand = replay
    batch = lidar_collate_fn([make_lidar_scene(200, **OUTDOOR_SCENE_KW),
                              make_lidar_scene(201, **OUTDOOR_SCENE_KW)])
    batch = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
    batch["mask_rand"] = torch.from_numpy(g["mask_rand"]).to(device)
    out = model(batch)
    out["loss"].backward()
    errs = {}
    for name, val in zip(g["out_names"], g["out_values"]):
        errs[str(name)] = abs(float(out[str(name)].detach()) - val) / (abs(val) + 1e-12)
    params = dict(model.named_parameters())
    for i, name in enumerate(g["grad_names"]):
        errs["grad_" + str(name)] = rel_err(params[str(name)].grad, g[f"grad_{i}"])
    return errs


def run_ponder_outdoor_full(device, with_float64=True, real_init=False):
    """BASELINE.json configs[4] at FULL size, one sweep: the reference's nuScenes model section
    unchanged (SpUNet-v1m1 32..256 over a 1080 x 1080 x 80 voxel range, 180 x 180 x 5 dense grid,
    SimpleConv3D, 16-wide five-block SDF MLP, 72 + 24 samples, 6 x 512 rays, mask ratio 0.8) against
    the reference's own step (oracle/make_golden.py::ponder_outdoor_full_case)."""
    from ponderv2_amd.ponder.datasets import lidar_collate_fn, make_lidar_scene
    from ponderv2_amd.ponder.models import build_model
    from ponderv2_amd.ponder.utils.config import ConfigDict

    g = np.load(os.path.join(GOLDEN, "ponder_outdoor_full%s.npz" % ("_real_init" if real_init else "")))
    cfg = outdoor_model_cfg(dict(FULL_BACKBONE, in_channels=4))
    if real_init:
        torch.manual_seed(0)     # the reference's own initialisation, drawn from the same seed
    model = build_model(ConfigDict(cfg))
    if not real_init:
        fill_deterministic(model)
    model = model.to(device).train()
    replay = ReplayRand([g[f"rand_{i}"] for i in range(int(g["rands"]))], device)
    model.renderer.sampler.initial_sampler.rand = replay
    model.renderer.sampler.pdf_sampler.rand = replay
    batch = lidar_collate_fn([make_lidar_scene(900, point_nsample=512)])
    assert int(batch["offset"][-1]) == int(g["n_voxels"]) and int(batch["ray_offset"][-1]) == int(g["n_rays"])
    batch = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
    batch["mask_rand"] = torch.from_numpy(g["mask_rand"]).to(device)
    out = model(batch)
    out["loss"].backward()
    errs = {}
    for name, val in zip(g["out_names"], g["out_values"]):
        errs[str(name)] = abs(float(out[str(name)].detach()) - val) / (abs(val) + 1e-12)
    # the reference's own float64 pass of the same step: how far ITS fp32 loss is from the exact
    # one (1.1e-4 here: a depth loss over 3 072 rays of 96 samples), and how far ours is
    ref64 = dict(zip((str(n) for n in g["out64_names"]), g["out64_values"]))
    ref32 = dict(zip((str(n) for n in g["out_names"]), g["out_values"]))
    for name, val in ref64.items():
        errs["f64_" + name] = abs(float(out[name].detach()) - val) / abs(val)
        errs["ref32_f64_" + name] = abs(ref32[name] - val) / abs(val)
    params = dict(model.named_parameters())
    for i, name in enumerate(g["grad_names"]):
        errs["grad_" + str(name)] = rel_err(params[str(name)].grad, g[f"grad_{i}"])
    if with_float64:
        errs["float64"] = float64_gradient_errors(params, g)
    return errs


PDNORM_BACKBONE = dict(type="SpUNet-v1m3", in_channels=6, num_classes=0, base_channels=16,
                       context_channels=32, channels=(16, 32, 48, 64, 64, 48, 32, 32),
                       layers=(1, 1, 1, 1, 1, 1, 1, 1), conditions=("ScanNet", "S3DIS", "Structured3D"),
                       zero_init=False, norm_decouple=True, norm_adaptive=True, norm_affine=True)


def run_spunet_pdnorm(device, dtype, real_init=False):
    from ponderv2_amd.ponder.models import build_model
    from ponderv2_amd.ponder.utils.config import ConfigDict

    g = np.load(os.path.join(GOLDEN, "spunet_pdnorm_small_real_init.npz" if real_init
                             else "spunet_pdnorm_small.npz"))
    coords = g["coords"]
    counts = np.bincount(coords[:, 0])
    torch.manual_seed(0)
    model = build_model(ConfigDict(PDNORM_BACKBONE))
    if real_init:
        real_init_modulation(model)
    else:
        fill_deterministic(model)
    model = model.to(dtype).to(device).train()
    n = len(coords)
    feat = formula_tensor("pdnorm.feat", (n, 6), 1.0).to(dtype).to(device).requires_grad_(True)
    context = formula_tensor("pdnorm.context", (1, 32), 1.0).to(dtype).to(device).requires_grad_(True)
    out = model(dict(grid_coord=torch.from_numpy(coords[:, 1:].astype(np.int64)).to(device),
                     feat=feat, offset=torch.from_numpy(np.cumsum(counts)).long().to(device),
                     condition=["S3DIS"], context=context))
    probe = formula_tensor("pdnorm.probe", tuple(out.shape), 1.0).to(dtype).to(device)
    (out * probe).sum().backward()
    params = dict(model.named_parameters())
    errs = {"out": rel_err(out, g["out"]), "dfeat": rel_err(feat.grad, g["dfeat"]),
            "dcontext": rel_err(context.grad, g["dcontext"])}
    cos = {"dfeat": cos_err(feat.grad, g["dfeat"]), "dcontext": cos_err(context.grad, g["dcontext"])}
    ge, gcos = _gradient_errors(params, g)
    errs.update(ge)
    cos.update(gcos)
    untouched = [k for k, p in params.items() if ".bns.0." in k or ".bns.2." in k]
    assert untouched and all(params[k].grad is None for k in untouched)
    if dtype == torch.float64:
        return errs
    return errs, cos


PPT_CONDITIONS = ("Structured3D", "ScanNet", "S3DIS")
PPT_VALID = (tuple(range(0, 13)), tuple(range(5, 25)), tuple(range(20, 36)))


def run_ponder_ppt(device):
    """PonderIndoor over SpUNet-v1m3, batch of condition "ScanNet" (BASELINE config 4 in miniature)."""
    from ponderv2_amd.ponder.datasets import collate_fn, make_scene
    from ponderv2_amd.ponder.models import build_model
    from ponderv2_amd.ponder.utils.config import ConfigDict

    g = np.load(os.path.join(GOLDEN, "ponder_ppt_small.npz"))
    cfg = indoor_model_cfg(dict(PDNORM_BACKBONE, context_channels=256,
                                channels=(16, 32, 48, 64, 64, 48, 32, 96)),
                           grid_shape=(32, 32, 8), ray_nsample=20)
    cfg.update(conditions=PPT_CONDITIONS, class_name=tuple(f"class {i}" for i in range(36)),
               valid_index=PPT_VALID, template=("a", "b"))
    model = build_model(ConfigDict(cfg))
    fill_deterministic(model)
    model = model.to(device).train()
    replay = ReplayRand([g[f"rand_{i}"] for i in range(int(g["rands"]))], device)
    model.renderer.sampler.initial_sampler.rand = replay
    model.renderer.sampler.pdf_sampler.rand = replay
    kw = dict(n_raw=16000, num_views=2, image_hw=(48, 64), condition="ScanNet", num_classes=20)
    batch = collate_fn([make_scene(300, **kw), make_scene(301, **kw)])
    batch = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
    batch["ray_pixels"] = torch.from_numpy(g["ray_pixels"])
    out = model(batch)
    out["loss"].backward()
    errs = {}
    for name, val in zip(g["out_names"], g["out_values"]):
        errs[str(name)] = abs(float(out[str(name)].detach()) - val) / (abs(val) + 1e-12)
    params = dict(model.named_parameters())
    for i, name in enumerate(g["grad_names"]):
        errs["grad_" + str(name)] = rel_err(params[str(name)].grad, g[f"grad_{i}"])
    return errs


def check_model_errors(errs, loss_tol=1e-4, rest_tol=5e-3, deep_tol=5e-2):
    """Shared assertion of the end-to-end golden tests.  Every loss term to ``loss_tol`` (the north
    star's 1e-4 - no exception path).  Gradients that have crossed the sparse backbone's ~60
    BatchNorm layers (backbone parameters, the mask token, the context embedding) to ``deep_tol``:
    in these miniature scenes a different fp32 summation order alone (1 thread instead of 128 on
    the host) moves them by up to ~3e-2 while the loss moves by 1e-7..1e-4 (measured with 128 host
    threads: <= 4.4e-4; the GPU tests pass 5e-3); every other gradient to ``rest_tol`` (measured
    <= 7.1e-4 on the host).

    The importance sampler inverts a CDF with ``searchsorted`` - integer work the full-size fixtures
    pin bit-exactly (tests/golden/ponder_indoor_cfg*.npz: ``pdf_bins``).  The GPU tests run the
    kernels in their DETERMINISTIC mode (output-stationary convs: identical forward bits on every
    run), so a sample cannot land in a different bin from one run to the next."""
    deep_keys = [k for k in errs if k.startswith("grad_backbone.")
                 or k in ("grad_mtoken", "grad_embedding_table.weight")]
    losses = {k: v for k, v in errs.items() if not k.startswith("grad_")}
    rest = {k: v for k, v in errs.items() if k.startswith("grad_") and k not in deep_keys}
    assert max(losses.values()) < loss_tol, errs
    assert not deep_keys or max(errs[k] for k in deep_keys) < deep_tol, errs
    assert not rest or max(rest.values()) < rest_tol, errs
