"""Generate tests/golden/*.npz by running the REFERENCE's own model files on the host.

Runs only in the build container (needs /root/reference; the GPU box never sees it):
    python -m oracle.make_golden
The reference's Python is imported unmodified through oracle/ref_shims.py; its out-of-tree
dependencies (spconv, smooth_sampler, torch_scatter) are served by the oracle restatements, so
these vectors pin (a) our re-written model/render code against the reference's control flow and
arithmetic and (b) the HIP kernels end to end.  Random draws made by the reference
(torch.rand / torch.randperm) are recorded into the fixtures so tests replay them exactly.

Weights are filled by a closed-form rule of the parameter NAME and element index
(oracle/detweights.py), so neither side needs a checkpoint or a matching RNG stream.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import ref_shims  # noqa: E402
from oracle.detweights import (GRAD_PROBES, fill_deterministic, formula_tensor, grad_probe,  # noqa: E402
                               real_init_modulation)

GOLDEN = os.path.join(ROOT, "tests", "golden")

SMALL_BACKBONE = dict(type="SpUNet-v1m1", in_channels=6, num_classes=0, base_channels=16,
                      channels=(16, 32, 48, 64, 64, 48, 32, 32), layers=(1, 1, 1, 1, 1, 1, 1, 1))


class Recorder:
    """Records every torch.rand / torch.randperm result made while active."""

    def __enter__(self):
        self.rand, self.perm = [], []
        self._rand, self._perm = torch.rand, torch.randperm

        def rand(*a, **k):
            t = self._rand(*a, **k)
            self.rand.append(t.detach().cpu().clone())
            return t

        def randperm(*a, **k):
            t = self._perm(*a, **k)
            self.perm.append(t.detach().cpu().clone())
            return t

        self.search, self._search = [], torch.searchsorted

        def searchsorted(*a, **k):
            t = self._search(*a, **k)
            self.search.append(t.detach().cpu().clone())
            return t

        torch.rand, torch.randperm, torch.searchsorted = rand, randperm, searchsorted
        return self

    def __exit__(self, *exc):
        torch.rand, torch.randperm, torch.searchsorted = self._rand, self._perm, self._search


class Replay:
    """Feeds the draws a ``Recorder`` captured back to torch.rand / torch.randperm, in order, cast to
    whatever dtype the caller asks for (the float64 reference pass re-runs the fp32 pass's draws)."""

    def __init__(self, rec):
        self.rand, self.perm = list(rec.rand), list(rec.perm)

    def __enter__(self):
        self._rand, self._perm = torch.rand, torch.randperm
        ri, pi = iter(self.rand), iter(self.perm)

        def rand(*a, **k):
            t = next(ri)
            return t.to(k.get("dtype") or torch.get_default_dtype()) if k.get("dtype") else t.clone()

        torch.rand, torch.randperm = rand, lambda *a, **k: next(pi).clone()
        return self

    def __exit__(self, *exc):
        torch.rand, torch.randperm = self._rand, self._perm


class _EverythingDouble:
    """The reference's model code pins float32 in a few places (``torch.FloatTensor``, ``.float()``,
    default-dtype factories).  For the float64 reference pass those are widened for the duration of
    the forward: default dtype float64, ``torch.FloatTensor`` builds doubles, ``Tensor.float()`` is
    the identity on doubles.  The reference's source is untouched."""

    def __enter__(self):
        self._default = torch.get_default_dtype()
        self._ft, self._float = torch.FloatTensor, torch.Tensor.float
        torch.set_default_dtype(torch.float64)
        torch.FloatTensor = lambda *a: torch.tensor(*a, dtype=torch.float64)
        keep = self._float
        torch.Tensor.float = lambda t, *a, **k: t if t.dtype == torch.float64 else keep(t, *a, **k).double()
        return self

    def __exit__(self, *exc):
        torch.set_default_dtype(self._default)
        torch.FloatTensor, torch.Tensor.float = self._ft, self._float


def float64_gradient_record(model, batch, rec, post=None):
    """Re-run the reference step in FLOAT64 (same weights, same inputs, the fp32 pass's random draws
    replayed) and record, for EVERY parameter that gets a gradient, its L2 norm and GRAD_PROBES
    random projections - what tests/ compare the GPU's fp32 gradients with (all ~240 tensors of the
    step, not eight probes; errors of the fp32 host reference itself no longer sit in the bound)."""
    import copy
    import time

    m64 = copy.deepcopy(model).double()
    for p in m64.parameters():
        p.grad = None
    if post is not None:
        post(m64)
    inp = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else
               v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
    t0 = time.perf_counter()
    with Replay(rec), _EverythingDouble():
        out = m64(inp)
    out["loss"].backward()
    names, norms, projs = _gradient_projections(m64)
    # the SAME record of the reference's own fp32 pass (``model`` still holds its gradients): how far
    # fp32 arithmetic alone moves these gradients - the yardstick for the GPU's fp32 gradients
    names32, norms32, projs32 = _gradient_projections(model)
    assert names32 == names
    e2 = ((np.array(projs32) - np.array(projs)) ** 2).mean(1)
    print("   float64 reference pass: %.1f s, %d parameters with gradients, loss %.9f; the reference's "
          "fp32 gradients differ from it by %.3e (global relative, estimated from %d projections)"
          % (time.perf_counter() - t0, len(names), float(out["loss"]),
             (e2.sum() / (np.array(norms) ** 2).sum()) ** 0.5, GRAD_PROBES))
    return dict(g64_names=np.array(names), g64_norm=np.array(norms), g64_proj=np.array(projs),
                g32_norm=np.array(norms32), g32_proj=np.array(projs32),
                out64_names=np.array(list(out.keys())),
                out64_values=np.array([float(v) for v in out.values()]))


def _gradient_projections(model):
    names, norms, projs = [], [], []
    for name, p in model.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.double()
        names.append(name)
        norms.append(float(g.norm()))
        projs.append([float((g * grad_probe(name, i, g.shape).double()).sum()) for i in range(GRAD_PROBES)])
    return names, norms, projs


def _compact_gradients(params, names, small=4096):
    """Every gradient of ``names`` for a fixture file: in full (float64) up to ``small`` elements, as its
    projections on GRAD_PROBES seeded random probes plus its norm beyond (oracle/detweights.grad_probe)."""
    out = {}
    for i, k in enumerate(names):
        g = params[k].grad.detach().double()
        if g.numel() <= small:
            out[f"grad_{i}"] = g.numpy()
        else:
            flat = g.reshape(-1)
            out[f"gproj_{i}"] = np.array([float(flat @ grad_probe(k, j, g.shape).double().reshape(-1))
                                          for j in range(GRAD_PROBES)])
            out[f"gnorm_{i}"] = np.array(float(flat.norm()))
    return out


def spunet_case(real_init=False):
    """Reference SpUNetBase (spconv_unet_v1m1_base.py:86-278) on the oracle sparse-conv runtime,
    float64, two small scenes.  ``real_init`` (round 6): the constructor's own initialisation under
    ``torch.manual_seed(0)`` (spconv_unet_v1m1_base.py:225-240) instead of the closed-form weights - a
    well-conditioned twin of the fixture, whose gradients can be held to fp32 bounds; every parameter's
    gradient is recorded."""
    from helpers import random_voxels
    from ponder.models.builder import MODELS

    coords = random_voxels(21, batch=2, extent=(48, 40, 24), n_per_batch=900)
    counts = np.bincount(coords[:, 0])
    grid_coord = torch.from_numpy(coords[:, 1:].astype(np.int64))
    n = len(coords)
    feat = formula_tensor("spunet.feat", (n, 6), 1.0).double()
    cfg = dict(SMALL_BACKBONE)
    torch.manual_seed(0)
    model = MODELS.build(cfg).double()
    if not real_init:
        fill_deterministic(model)
    model.train()
    feat.requires_grad_(True)
    out = model(dict(grid_coord=grid_coord, feat=feat,
                     offset=torch.from_numpy(np.cumsum(counts)).long()))
    probe = formula_tensor("spunet.probe", tuple(out.shape), 1.0).double()
    (out * probe).sum().backward()
    names = ["conv_input.0.weight", "enc.1.block0.conv1.weight", "down.2.0.weight",
             "up.1.0.weight", "dec.0.block0.proj.0.weight", "dec.3.block0.bn2.bias"]
    params = dict(model.named_parameters())
    if real_init:
        names = [k for k, p in params.items() if p.grad is not None]
    np.savez_compressed(
        os.path.join(GOLDEN, "spunet_small_real_init.npz" if real_init else "spunet_small.npz"),
        coords=coords, out=out.detach().numpy(),
        dfeat=feat.grad.numpy(), grad_names=np.array(names),
        **(_compact_gradients(params, names) if real_init else
           {f"grad_{i}": params[k].grad.numpy() for i, k in enumerate(names)}))
    print("spunet_small: out", tuple(out.shape), "mean |out|", out.abs().mean().item())


def _render_cfg():
    from ponderv2_amd.ponder.utils.config import Config

    cfg = Config.fromfile(os.path.join(ref_shims.REFERENCE_ROOT,
                                       "configs/scannet/pretrain-ponder-spunet-v1m1-0-base.py"))
    return cfg


def neus_case_impl(ConfigDict):
    from ponder.models.ponder.render_utils import RayBundle, build_renderer

    cfg = _render_cfg()
    rcfg = ConfigDict(cfg.model.renderer.to_dict())
    torch.manual_seed(0)
    renderer = build_renderer(rcfg)
    fill_deterministic(renderer)
    renderer.train()
    C, D, H, W, R = 128, 8, 16, 16, 40
    volume = (formula_tensor("neus.volume", (C, D, H, W), 0.6)).requires_grad_(True)
    # rays start outside/inside the padded unit cube and point towards its centre
    o = formula_tensor("neus.origins", (R, 3), 0.9)
    tgt = formula_tensor("neus.targets", (R, 3), 0.25)
    d = torch.nn.functional.normalize(tgt - o, dim=-1)
    targets = dict(depth=(formula_tensor("neus.depth", (R, 1), 0.5) + 0.45),
                   rgb=formula_tensor("neus.rgb", (R, 3), 0.5) + 0.5,
                   semantic=formula_tensor("neus.sem", (R, 512), 1.0))
    targets["depth"][::7] = -0.001          # invalid-depth rays
    targets["semantic"][3::5] = 0.0         # rays without a language target
    torch.manual_seed(123)
    with Recorder() as rec:
        out = renderer(RayBundle(origins=o, directions=d), [volume])
        losses = renderer.get_loss(out, targets)
    loss = sum(v for k, v in losses.items() if "loss" in k)
    loss.backward()
    params = dict(renderer.named_parameters())
    gnames = ["field.sdf_decoder.lin0.weight", "field.sdf_decoder.fc_c.1.weight",
              "field.rgb_decoder.fc_c.0.weight", "field.semantic_decoder.lin0.bias",
              "field.deviation_network.variance"]
    keys = ["rgb", "semantic", "depth", "normal", "weights", "sdf", "gradients", "z_vals"]
    np.savez_compressed(
        os.path.join(GOLDEN, "neus_head.npz"),
        origins=o.numpy(), directions=d.numpy(),
        rand0=rec.rand[0].numpy(), rand1=rec.rand[1].numpy(), n_rand=len(rec.rand),
        loss_names=np.array(list(losses.keys())),
        loss_values=np.array([float(v) for v in losses.values()]),
        dvolume=volume.grad.numpy().astype(np.float32), grad_names=np.array(gnames),
        **{f"out_{k}": out[k].detach().numpy() for k in keys},
        **{f"tgt_{k}": v.numpy() for k, v in targets.items()},
        **{f"grad_{i}": params[k].grad.numpy() for i, k in enumerate(gnames)})
    print("neus_head:", {k: round(float(v), 6) for k, v in losses.items()}, "n_rand", len(rec.rand))


def ponder_indoor_case(ConfigDict):
    """Reference PonderIndoor.forward (ponder_indoor_base.py:694-706) end to end on a synthetic
    two-scene batch with a reduced backbone / grid, fp32, training mode."""
    from ponder.models.builder import MODELS
    from ponderv2_amd.ponder.datasets import collate_fn, make_scene

    cfg = _render_cfg()
    mcfg = cfg.model.to_dict()
    mcfg["backbone"] = dict(SMALL_BACKBONE, channels=(16, 32, 48, 64, 64, 48, 32, 96))
    mcfg.update(grid_shape=(32, 32, 8), ray_nsample=20)
    torch.manual_seed(0)
    model = MODELS.build(ConfigDict(mcfg))
    fill_deterministic(model)
    model.train()
    scene_kw = dict(n_raw=16000, num_views=2, image_hw=(48, 64))
    batch = collate_fn([make_scene(100, **scene_kw), make_scene(101, **scene_kw)])
    inp = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
    torch.manual_seed(321)
    with Recorder() as rec:
        out = model(inp)
    out["loss"].backward()
    # pixels the reference picked: where(mask>0) order + randperm[:n]  (:546-551)
    B, V, H, W = batch["depth"].shape
    pix = np.zeros((B, V, 20, 2), dtype=np.int64)
    it = iter(rec.perm)
    for b in range(B):
        for v in range(V):
            ys, xs = torch.where(batch["depth"][b, v] > 0)
            sel = next(it)[:20]
            pix[b, v, :, 0], pix[b, v, :, 1] = ys[sel].numpy(), xs[sel].numpy()
    params = dict(model.named_parameters())
    gnames = ["backbone.conv_input.0.weight", "backbone.dec.0.block0.conv2.weight",
              "proj_net.encoders.0.basic_module.conv.weight", "proj_net.final_conv.bias",
              "renderer.field.sdf_decoder.lin1.weight"]
    np.savez_compressed(
        os.path.join(GOLDEN, "ponder_indoor_small.npz"), ray_pixels=pix,
        rands=np.array(len(rec.rand)),
        **{f"rand_{i}": r.numpy() for i, r in enumerate(rec.rand)},
        out_names=np.array(list(out.keys())), out_values=np.array([float(v) for v in out.values()]),
        grad_names=np.array(gnames),
        **{f"grad_{i}": params[k].grad.numpy() for i, k in enumerate(gnames)})
    print("ponder_indoor_small:", {k: round(float(v), 6) for k, v in out.items()},
          "rand draws", [tuple(r.shape) for r in rec.rand])


def ponder_indoor_cfg1_case(ConfigDict):
    """BASELINE.json configs[1] at FULL size (the bench workload: 2 scenes, 2 views x 256 = 512 rays
    per scene) - same recipe and fixture layout as configs[0] below."""
    ponder_indoor_cfg0_case(ConfigDict, scenes=2, rays_per_view=256, n_voxels=None,
                            name="ponder_indoor_cfg1")


def ponder_indoor_cfg1_real_init_case(ConfigDict):
    """configs[1] at full size with the reference's REAL initialisation (``torch.manual_seed(0)`` + the
    constructors' own ``_init_weights``, spconv_unet_v1m1_base.py:225-240: truncated-normal conv / linear
    weights, BatchNorm weight 1) instead of the closed-form weights of the other fixtures.  Those are
    convenient (no weight file) but make ~60 badly conditioned BatchNorm layers in a row, whose fp32
    gradients scatter by several per cent around the float64 ones even in the reference itself; on
    this well-conditioned net the fp32 / float64 distance is small, so the GPU's gradients can be held
    to a tight bound.  The product model draws the same weights from the same seed (same construction
    order: tests/test_golden_cpu.py checks state_dict equality), so no weights are stored."""
    ponder_indoor_cfg0_case(ConfigDict, scenes=2, rays_per_view=256, n_voxels=None,
                            name="ponder_indoor_cfg1_real_init", real_init=True)


def ponder_indoor_cfg0_case(ConfigDict, scenes=1, rays_per_view=64, n_voxels=20000,
                            name="ponder_indoor_cfg0", real_init=False):
    """BASELINE.json configs[0] at FULL size: the reference's PonderIndoor.forward with the shipped
    model section (SpUNet-v1m1 32..256 channels, (2,3,4,6,2,2,2,2) blocks, 128x128x32 grid,
    UNet3D-v1m2, NeuS head 96+36 samples) on one synthetic ScanNet-shaped scene of 20 000 voxels and
    2 views x 64 = 128 rays, fp32, training mode.  Besides the losses and gradient probes the
    fixture holds what the renderer returned per ray (RGB, depth), the importance sampler's
    searchsorted bin indices (integer work: asserted bit-exact by the tests) and the FLOAT64 record
    of every parameter gradient (float64_gradient_record)."""
    from ponderv2_amd.ponder.datasets import collate_fn, make_scene

    cfg = _render_cfg()
    mcfg = cfg.model.to_dict()
    mcfg.update(ray_nsample=rays_per_view, template="a photo of a [x]")
    batch = collate_fn([make_scene(i, num_views=2, image_hw=(480, 640), n_voxels=n_voxels)
                        for i in range(scenes)])
    gnames = ["backbone.conv_input.1.bias", "backbone.enc.3.block5.bn2.weight",
              "backbone.dec.0.block1.bn2.bias", "proj_net.final_conv.bias",
              "renderer.field.sdf_decoder.lin1.bias", "renderer.field.rgb_decoder.lin0.weight",
              "renderer.field.semantic_decoder.lin0.bias", "renderer.field.deviation_network.variance"]
    _indoor_full_step(ConfigDict, mcfg, batch, name, rays_per_view, gnames, real_init=real_init)


def _indoor_full_step(ConfigDict, mcfg, batch, name, rays_per_view, gnames, real_init=False):
    """One reference training step of a PonderIndoor model at full size -> tests/golden/<name>.npz."""
    import time

    from ponder.models.builder import MODELS

    scenes = len(batch["offset"])
    torch.manual_seed(0)
    model = MODELS.build(ConfigDict(mcfg))
    if not real_init:
        fill_deterministic(model)
    model.train()
    inp = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()
           if not k.endswith("_host")}
    captured = {"rgb": [], "depth": [], "normal": []}
    orig_render = model.renderer.forward

    def render(*a, **k):
        out = orig_render(*a, **k)
        for key in captured:   # the reference renders scene by scene
            captured[key].append(out[key].detach().clone())
        return out

    model.renderer.forward = render
    torch.manual_seed(321)
    t0 = time.perf_counter()
    with Recorder() as rec:
        out = model(inp)
    out["loss"].backward()
    print("%s reference step on %d host threads: %.1f s" % (name, torch.get_num_threads(),
                                                           time.perf_counter() - t0))
    B, V, H, W = batch["depth"].shape
    pix = np.zeros((B, V, rays_per_view, 2), dtype=np.int64)
    it = iter(rec.perm)
    for b in range(B):
        for v in range(V):
            ys, xs = torch.where(batch["depth"][b, v] > 0)
            sel = next(it)[:rays_per_view]
            pix[b, v, :, 0], pix[b, v, :, 1] = ys[sel].numpy(), xs[sel].numpy()
    params = dict(model.named_parameters())
    assert len(rec.search) == scenes
    model.renderer.forward = orig_render
    # the float64 pass also keeps what ITS renderer returned per ray: the exact RGB-D the fp32 renders of
    # either side are measured against (round 5: the real-initialisation fixtures' flat alphas put the
    # reference's own fp32 render ~1e-3 from it, so "within 1e-4 of the fp32 fixture" is the wrong bar)
    captured64 = {"rgb": [], "depth": []}

    def wrap64(m64):
        orig64 = m64.renderer.forward

        def render64(*a, **k):
            out64 = orig64(*a, **k)
            for key in captured64:
                captured64[key].append(out64[key].detach().clone())
            return out64

        m64.renderer.forward = render64

    g64 = float64_gradient_record(model, {k: v for k, v in batch.items() if not k.endswith("_host")}, rec,
                                  post=wrap64)
    np.savez_compressed(
        os.path.join(GOLDEN, name + ".npz"), ray_pixels=pix, **g64,
        render64_rgb=torch.cat(captured64["rgb"]).numpy(), render64_depth=torch.cat(captured64["depth"]).numpy(),
        rands=np.array(len(rec.rand)),
        **{f"rand_{i}": r.numpy() for i, r in enumerate(rec.rand)},
        pdf_bins=torch.cat(rec.search).numpy().astype(np.int32),
        render_rgb=torch.cat(captured["rgb"]).numpy(), render_depth=torch.cat(captured["depth"]).numpy(),
        render_normal=torch.cat(captured["normal"]).numpy(),
        n_voxels=np.array(int(batch["offset"][-1])),
        out_names=np.array(list(out.keys())), out_values=np.array([float(v) for v in out.values()]),
        grad_names=np.array(gnames),
        **{f"grad_{i}": params[k].grad.numpy() for i, k in enumerate(gnames)})
    print(name + ":", {k: round(float(v), 6) for k, v in out.items()},
          "rand draws", [tuple(r.shape) for r in rec.rand])


PPT_CONDITIONS = ("Structured3D", "ScanNet", "S3DIS")
PPT_VALID = (tuple(range(0, 13)), tuple(range(5, 25)), tuple(range(20, 36)))


def ponder_ppt_full_case(ConfigDict, rays_per_view=256, real_init=False):
    """BASELINE.json configs[3] at FULL size: the model section of the reference's shipped
    multi-dataset config (configs/scannet/pretrain-ponder-ppt-v1m1-0-sc-s3-st-spunet.py:22-90:
    SpUNet-v1m3 PDNorm 32..256 channels, (2,3,4,6,2,2,2,2) blocks, decoupled + adaptive + affine
    norms, 128x128x32 grid, UNet3D-v1m2, NeuS head, ray_nsample 256) - only the class tables are
    the stub text encoder's - run for ONE BATCH PER CONDITION (2 scenes, 512 rays each):
    tests/golden/ponder_ppt_full_<condition>.npz."""
    from ponderv2_amd.ponder.datasets import collate_fn, make_scene
    from ponderv2_amd.ponder.utils.config import Config

    cfg = Config.fromfile(os.path.join(
        ref_shims.REFERENCE_ROOT, "configs/scannet/pretrain-ponder-ppt-v1m1-0-sc-s3-st-spunet.py"))
    ref_shims.install(num_classes=36)
    for k, cond in enumerate(PPT_CONDITIONS):
        mcfg = cfg.model.to_dict()
        mcfg.update(ray_nsample=rays_per_view, conditions=PPT_CONDITIONS,
                    class_name=tuple(f"class {i}" for i in range(36)), valid_index=PPT_VALID,
                    template=("a", "b"))
        kw = dict(num_views=2, image_hw=(480, 640), condition=cond, num_classes=len(PPT_VALID[k]))
        batch = collate_fn([make_scene(700 + 10 * k + i, **kw) for i in range(2)])
        bns = PPT_CONDITIONS_BACKBONE.index(cond)
        gnames = ["embedding_table.weight", "backbone.conv_input.bn.modulation.1.weight",
                  f"backbone.enc.3.block5.bn2.bns.{bns}.weight",
                  f"backbone.dec.0.block1.bn2.bns.{bns}.bias", "proj_net.final_conv.bias",
                  "renderer.field.sdf_decoder.lin1.bias", "renderer.field.semantic_decoder.lin0.bias",
                  "renderer.field.deviation_network.variance"]
        _indoor_full_step(ConfigDict, mcfg, batch,
                          "ponder_ppt_full_" + cond.lower() + ("_real_init" if real_init else ""),
                          rays_per_view, gnames, real_init=real_init)
    ref_shims.install()  # back to the default 20-class text table


PPT_CONDITIONS_BACKBONE = ("ScanNet", "S3DIS", "Structured3D")   # the shipped backbone's order


# reduced nuScenes geometry shared with tests/golden_cases.py: a 54 m box, 1.2 m dense cells
OUTDOOR_SMALL = dict(scene_bbox=((-27.0, -27.0, -5.0, 27.0, 27.0, 3.0),), grid_shape=((45, 45, 5),),
                     grid_size=((1.2, 1.2, 1.6),))
OUTDOOR_SCENE_KW = dict(grid_size=0.1, point_nsample=24, n_azimuth=200,
                        point_cloud_range=(-27.0, -27.0, -5.0, 27.0, 27.0, 3.0))


def lidar_transform_case():
    """Reference PointRangeFilter / GridSample(ravel) / ProjectOnImage / RaySample
    (datasets/transform.py:232-378,1078-1213) on one synthetic sweep, numpy seed fixed."""
    from ponderv2_amd.ponder.datasets import make_sweep

    T = ref_shims.load_reference_file("ponder/datasets/transform.py")
    data = make_sweep(7, n_azimuth=200)
    np.random.seed(7)
    data = T.PointRangeFilter(point_cloud_range=OUTDOOR_SCENE_KW["point_cloud_range"], padding=0.1)(data)
    data = T.GridSample(grid_size=0.1, hash_type="ravel", mode="train",
                        keys=("coord", "strength", "segment"), return_grid_coord=True)(data)
    data = T.ProjectOnImage(filter_overlap=True, close_radius=3.0)(data)
    n_proj = np.array([int(m.sum()) for m in data["img_proj_mask"]])
    data = T.RaySample(point_nsample=24, fetch_color=False, fetch_segment=True)(data)
    np.savez_compressed(os.path.join(GOLDEN, "lidar_transforms.npz"),
                        grid_coord=data["grid_coord"].astype(np.int32), n_proj=n_proj,
                        ray_start=data["ray_start"], ray_end=data["ray_end"],
                        ray_segment=data["ray_segment"])
    print("lidar_transforms: voxels", len(data["grid_coord"]), "projected", n_proj.tolist(),
          "rays", len(data["ray_end"]))


def ponder_outdoor_case(ConfigDict):
    """Reference PonderOutdoor.forward (ponder_outdoor_base.py:258-265) end to end on a synthetic
    two-sweep batch: block masking with mtoken, reduced backbone / grid, fp32, training mode."""
    from ponder.models.builder import MODELS
    from ponderv2_amd.ponder.datasets import lidar_collate_fn, make_lidar_scene
    from ponderv2_amd.ponder.utils.config import Config

    cfg = Config.fromfile(os.path.join(ref_shims.REFERENCE_ROOT,
                                       "configs/nuscenes/pretrain-ponder-spunet-v1m1-0-base.py"))
    mcfg = cfg.model.to_dict()
    mcfg["backbone"] = dict(SMALL_BACKBONE, in_channels=4, channels=(16, 32, 48, 64, 64, 48, 32, 96))
    mcfg.update(OUTDOOR_SMALL)
    torch.manual_seed(0)
    model = MODELS.build(ConfigDict(mcfg))
    fill_deterministic(model)
    model.train()
    batch = lidar_collate_fn([make_lidar_scene(200, **OUTDOOR_SCENE_KW),
                              make_lidar_scene(201, **OUTDOOR_SCENE_KW)])
    inp = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()
           if not k.endswith("_host")}
    torch.manual_seed(99)
    with Recorder() as rec:
        out = model(inp)
    out["loss"].backward()
    B = len(batch["offset"])
    mask_rand = np.concatenate([r.numpy().reshape(-1) for r in rec.rand[:B]])
    draws = rec.rand[B:]
    params = dict(model.named_parameters())
    gnames = ["mtoken", "backbone.conv_input.0.weight", "backbone.dec.0.block0.conv2.weight",
              "proj_net.conv.0.weight", "proj_net.conv.1.bias",
              "renderer.field.sdf_decoder.lin1.weight", "renderer.field.deviation_network.variance"]
    np.savez_compressed(
        os.path.join(GOLDEN, "ponder_outdoor_small.npz"), mask_rand=mask_rand,
        rands=np.array(len(draws)), **{f"rand_{i}": r.numpy() for i, r in enumerate(draws)},
        out_names=np.array(list(out.keys())), out_values=np.array([float(v) for v in out.values()]),
        grad_names=np.array(gnames),
        **{f"grad_{i}": params[k].grad.numpy() for i, k in enumerate(gnames)})
    print("ponder_outdoor_small:", {k: round(float(v), 6) for k, v in out.items()},
          "mask draws", [tuple(r.shape) for r in rec.rand[:B]],
          "sampler draws", [tuple(r.shape) for r in draws],
          "rays", batch["ray_offset"].tolist(), "voxels", batch["offset"].tolist())


def ponder_outdoor_full_case(ConfigDict, real_init=False):
    """BASELINE.json configs[4] at FULL size: the reference's PonderOutdoor.forward with the model
    section of configs/nuscenes/pretrain-ponder-spunet-v1m1-0-base.py UNCHANGED (SpUNet-v1m1
    32..256 channels over a 1080 x 1080 x 80 voxel range, 180 x 180 x 5 dense grid, SimpleConv3D
    projection, 16-wide five-block SDF MLP, 72 + 24 samples, mask ratio 0.8) on ONE synthetic lidar
    sweep with 6 x 512 rays; fp32 step + the float64 record of every parameter gradient."""
    import time

    from ponder.models.builder import MODELS
    from ponderv2_amd.ponder.datasets import lidar_collate_fn, make_lidar_scene
    from ponderv2_amd.ponder.utils.config import Config

    cfg = Config.fromfile(os.path.join(ref_shims.REFERENCE_ROOT,
                                       "configs/nuscenes/pretrain-ponder-spunet-v1m1-0-base.py"))
    mcfg = cfg.model.to_dict()
    ref_shims.install(num_classes=16)
    torch.manual_seed(0)
    model = MODELS.build(ConfigDict(mcfg))
    if not real_init:   # (real_init: the constructors' own initialisation under torch.manual_seed(0))
        fill_deterministic(model)
    model.train()
    batch = lidar_collate_fn([make_lidar_scene(900, point_nsample=512)])
    inp = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()
           if not k.endswith("_host")}
    torch.manual_seed(99)
    t0 = time.perf_counter()
    with Recorder() as rec:
        out = model(inp)
    out["loss"].backward()
    print("ponder_outdoor_full reference step: %.1f s" % (time.perf_counter() - t0))
    B = len(batch["offset"])
    mask_rand = np.concatenate([r.numpy().reshape(-1) for r in rec.rand[:B]])
    draws = rec.rand[B:]
    params = dict(model.named_parameters())
    gnames = ["mtoken", "backbone.conv_input.1.bias", "backbone.enc.3.block5.bn2.weight",
              "backbone.dec.0.block1.bn2.bias", "proj_net.conv.1.bias",
              "renderer.field.sdf_decoder.lin1.bias", "renderer.field.deviation_network.variance"]
    g64 = float64_gradient_record(model, {k: v for k, v in batch.items() if not k.endswith("_host")}, rec)
    np.savez_compressed(
        os.path.join(GOLDEN, "ponder_outdoor_full%s.npz" % ("_real_init" if real_init else "")),
        mask_rand=mask_rand, **g64,
        rands=np.array(len(draws)), **{f"rand_{i}": r.numpy() for i, r in enumerate(draws)},
        n_voxels=np.array(int(batch["offset"][-1])), n_rays=np.array(int(batch["ray_offset"][-1])),
        out_names=np.array(list(out.keys())), out_values=np.array([float(v) for v in out.values()]),
        grad_names=np.array(gnames),
        **{f"grad_{i}": params[k].grad.numpy() for i, k in enumerate(gnames)})
    ref_shims.install()
    print("ponder_outdoor_full:", {k: round(float(v), 6) for k, v in out.items()},
          "rays", batch["ray_offset"].tolist(), "voxels", batch["offset"].tolist())


PDNORM_BACKBONE = dict(type="SpUNet-v1m3", in_channels=6, num_classes=0, base_channels=16,
                       context_channels=32, channels=(16, 32, 48, 64, 64, 48, 32, 32),
                       layers=(1, 1, 1, 1, 1, 1, 1, 1), conditions=("ScanNet", "S3DIS", "Structured3D"),
                       zero_init=False, norm_decouple=True, norm_adaptive=True, norm_affine=True)


def spunet_pdnorm_case(real_init=False):
    """Reference SpUNet-v1m3 (spconv_unet_v1m3_pdnorm.py:236-427: per-condition BatchNorm +
    context modulation) on the oracle sparse-conv runtime, float64, condition "S3DIS".  ``real_init``
    (round 6): the constructor's own initialisation under ``torch.manual_seed(0)`` - with the modulation
    layers' weights redrawn (the reference zero-initialises them, spconv_unet_v1m3_pdnorm.py:389-404, which
    would leave the modulation path untested) - and every parameter's gradient recorded."""
    from helpers import random_voxels
    from ponder.models.builder import MODELS

    coords = random_voxels(33, batch=2, extent=(48, 40, 24), n_per_batch=900)
    counts = np.bincount(coords[:, 0])
    n = len(coords)
    feat = formula_tensor("pdnorm.feat", (n, 6), 1.0).double().requires_grad_(True)
    context = formula_tensor("pdnorm.context", (1, 32), 1.0).double().requires_grad_(True)
    torch.manual_seed(0)
    model = MODELS.build(dict(PDNORM_BACKBONE))
    if real_init:
        real_init_modulation(model)
    else:
        fill_deterministic(model)
    model = model.double().train()
    out = model(dict(grid_coord=torch.from_numpy(coords[:, 1:].astype(np.int64)), feat=feat,
                     offset=torch.from_numpy(np.cumsum(counts)).long(), condition=["S3DIS"],
                     context=context))
    probe = formula_tensor("pdnorm.probe", tuple(out.shape), 1.0).double()
    (out * probe).sum().backward()
    params = dict(model.named_parameters())
    names = ["conv_input.conv.weight", "conv_input.bn.bns.1.weight",
             "conv_input.bn.modulation.1.weight", "enc.1.block0.bn1.modulation.1.bias",
             "down.2.bn.bns.1.bias", "dec.0.block0.proj_norm.modulation.1.weight",
             "dec.3.block0.bn2.bns.1.weight", "up.1.conv.weight"]
    unused = [k for k, p in params.items() if ".bns.0." in k or ".bns.2." in k]
    assert unused and all(params[k].grad is None for k in unused)  # other datasets' BN untouched
    if real_init:
        names = [k for k, p in params.items() if p.grad is not None]
    np.savez_compressed(
        os.path.join(GOLDEN, "spunet_pdnorm_small_real_init.npz" if real_init else "spunet_pdnorm_small.npz"),
        coords=coords, out=out.detach().numpy(),
        dfeat=feat.grad.numpy(), dcontext=context.grad.numpy(), grad_names=np.array(names),
        **(_compact_gradients(params, names) if real_init else
           {f"grad_{i}": params[k].grad.numpy() for i, k in enumerate(names)}))
    print("spunet_pdnorm_small: out", tuple(out.shape), "mean |out|", out.abs().mean().item(),
          "|dcontext|", context.grad.abs().max().item())


def transform_chain_case():
    """The reference's own transform classes (datasets/transform.py) on the ScanNet pre-training
    chain, fixed python / numpy seeds."""
    import test_transforms as tt

    T = ref_shims.load_reference_file("ponder/datasets/transform.py")
    seed = 7
    out = tt.flatten(tt.run_chain(T.Compose, tt.SCANNET_CHAIN, tt.raw_scene(), seed))
    np.savez_compressed(os.path.join(GOLDEN, "transform_chain.npz"), seed=np.array(seed), **out)
    print("transform_chain:", {k: v.shape for k, v in out.items()})




def ponder_ppt_case(ConfigDict):
    """Reference PonderIndoor.forward over SpUNet-v1m3 (BASELINE config 4 in miniature): context
    embedding per condition, PDNorm backbone, the condition's valid class subset for the language
    targets and the ppt loss; one batch of condition "ScanNet" (index 1 of the model's conditions,
    index 0 of the backbone's)."""
    from ponder.models.builder import MODELS
    from ponderv2_amd.ponder.datasets import collate_fn, make_scene

    cfg = _render_cfg()
    mcfg = cfg.model.to_dict()
    mcfg["backbone"] = dict(PDNORM_BACKBONE, context_channels=256,
                            channels=(16, 32, 48, 64, 64, 48, 32, 96))
    mcfg.update(grid_shape=(32, 32, 8), ray_nsample=20, conditions=PPT_CONDITIONS,
                class_name=tuple(f"class {i}" for i in range(36)), valid_index=PPT_VALID,
                template=("a", "b"))
    ref_shims.install(num_classes=36)
    torch.manual_seed(0)
    model = MODELS.build(ConfigDict(mcfg))
    fill_deterministic(model)
    model.train()
    scene_kw = dict(n_raw=16000, num_views=2, image_hw=(48, 64), condition="ScanNet", num_classes=20)
    batch = collate_fn([make_scene(300, **scene_kw), make_scene(301, **scene_kw)])
    inp = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()
           if not k.endswith("_host")}
    torch.manual_seed(77)
    with Recorder() as rec:
        out = model(inp)
    out["loss"].backward()
    B, V, H, W = batch["depth"].shape
    pix = np.zeros((B, V, 20, 2), dtype=np.int64)
    it = iter(rec.perm)
    for b in range(B):
        for v in range(V):
            ys, xs = torch.where(batch["depth"][b, v] > 0)
            sel = next(it)[:20]
            pix[b, v, :, 0], pix[b, v, :, 1] = ys[sel].numpy(), xs[sel].numpy()
    params = dict(model.named_parameters())
    gnames = ["embedding_table.weight", "backbone.conv_input.conv.weight",
              "backbone.conv_input.bn.modulation.1.weight", "backbone.dec.0.block0.bn2.bns.0.weight",
              "proj_net.final_conv.bias", "renderer.field.semantic_decoder.lin0.weight"]
    np.savez_compressed(
        os.path.join(GOLDEN, "ponder_ppt_small.npz"), ray_pixels=pix, rands=np.array(len(rec.rand)),
        **{f"rand_{i}": r.numpy() for i, r in enumerate(rec.rand)},
        out_names=np.array(list(out.keys())), out_values=np.array([float(v) for v in out.values()]),
        grad_names=np.array(gnames),
        **{f"grad_{i}": params[k].grad.numpy() for i, k in enumerate(gnames)})
    ref_shims.install()  # back to the default 20-class text table
    print("ponder_ppt_small:", {k: round(float(v), 6) for k, v in out.items()})


def narrow_decoder_case():
    """The nuScenes head's SDF decoder in isolation (tests/golden/narrow_decoder.npz): the REFERENCE's
    own ``SDFDecoder(in_dim=32, out_dim=17, hidden_size=16, n_blocks=5)``
    (ponder/models/ponder/render_utils/decoders.py:6-36) in float64 on random points / features, with
    the gradient of its SDF output with respect to the points through a LINEAR feature field
    feat = A p + a (so that d feat / d p = A is known exactly) - what sdf_field.py:211-250 takes by
    autograd.  Pins oracle/narrow_head.py's value / tangent recursion and the parameter layout of
    ponderv2_amd/narrow_head.py::pack_theta."""
    D = ref_shims.load_reference_file("ponder/models/ponder/render_utils/decoders.py")
    torch.manual_seed(7)
    dec = D.SDFDecoder(in_dim=32, out_dim=17, hidden_size=16, n_blocks=5).double()
    with torch.no_grad():
        for q in dec.parameters():          # well away from the softplus threshold on both sides
            q.mul_(1.5)
    n = 64
    pts = torch.rand(n, 3, dtype=torch.float64, requires_grad=True)
    A = torch.randn(32, 3, dtype=torch.float64) * 0.4
    a = torch.randn(32, dtype=torch.float64) * 0.2
    feat = pts @ A.t() + a
    out = dec(pts, feat)
    (g,) = torch.autograd.grad(out[:, 0].sum(), pts)
    names = [k for k, _ in dec.named_parameters()]
    np.savez(os.path.join(GOLDEN, "narrow_decoder.npz"), points=pts.detach().numpy(), A=A.numpy(),
             a=a.numpy(), out=out.detach().numpy(), grad_sdf=g.numpy(), param_names=np.array(names),
             **{f"param_{i}": q.detach().numpy() for i, (_, q) in enumerate(dec.named_parameters())})
    print("narrow_decoder:", out.shape, float(out[:, 0].abs().max()), float(g.abs().max()))


def main():
    ref_shims.install()
    os.makedirs(GOLDEN, exist_ok=True)
    from ponderv2_amd.ponder.utils.config import ConfigDict  # attribute dict, like addict's

    only = sys.argv[1:]
    cases = dict(spunet=spunet_case, neus=lambda: neus_case_impl(ConfigDict),
                 indoor=lambda: ponder_indoor_case(ConfigDict), lidar=lidar_transform_case, pdnorm=spunet_pdnorm_case, transforms=transform_chain_case, ppt=lambda: ponder_ppt_case(ConfigDict),
                 outdoor=lambda: ponder_outdoor_case(ConfigDict),
                 cfg0=lambda: ponder_indoor_cfg0_case(ConfigDict),
                 cfg1=lambda: ponder_indoor_cfg1_case(ConfigDict),
                 cfg1_real=lambda: ponder_indoor_cfg1_real_init_case(ConfigDict),
                 ppt_full=lambda: ponder_ppt_full_case(ConfigDict),
                 outdoor_full=lambda: ponder_outdoor_full_case(ConfigDict),
                 # round 5: the REAL initialisation for configs[3] (one fixture per condition) and configs[4]
                 ppt_full_real=lambda: ponder_ppt_full_case(ConfigDict, real_init=True),
                 outdoor_full_real=lambda: ponder_outdoor_full_case(ConfigDict, real_init=True),
                 # round 6: real-initialisation twins of the two small backbone fixtures
                 spunet_real=lambda: spunet_case(real_init=True),
                 pdnorm_real=lambda: spunet_pdnorm_case(real_init=True),
                 narrow_decoder=narrow_decoder_case)
    for name, fn in cases.items():
        if not only or name in only:
            fn()


if __name__ == "__main__":
    main()
