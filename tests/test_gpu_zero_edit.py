"""Zero-edit drop-in: the reference's UNMODIFIED model files
run over the PRODUCT mirrors (``ponderv2_amd.dropin.install()``: spconv.pytorch, smooth_sampler,
torch_scatter -> libponderv2_hip.so) on the MI355X and reproduce the golden vectors the reference
itself produced on the host oracle (oracle/make_golden.py).

Needs a checkout of the reference: ``PONDERV2_REFERENCE=<path>`` (the GPU box has none by default -
tools/zero_edit_trip.sh ships a scratch copy for one gpurun call; it is never committed).  Skipped
otherwise.  The packages the reference imports that are NOT part of the hot path and are absent
from this image (timm's trunc_normal_, clip's text encoder, torch_geometric's scatter helper used
outside the path) come from oracle/ref_shims.py as in the CPU parity tests; the three native
boundaries are then re-pointed at the product.
"""
import os
import sys

import numpy as np
import pytest
import torch

import golden_cases as gc
from oracle import ref_shims
from oracle.detweights import fill_deterministic, formula_tensor

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not ref_shims.reference_available(),
                                 reason="no reference checkout (set PONDERV2_REFERENCE)")]


@pytest.fixture(scope="module")
def reference_over_product():
    """sys.modules as an unmodified reference process would see them with the drop-in installed."""
    from ponderv2_amd import dropin

    saved = {k: sys.modules.get(k) for k in list(sys.modules)
             if k == "ponder" or k.startswith("ponder.") or k in dropin._NAMES}
    for k in [k for k in sys.modules if k == "ponder" or k.startswith("ponder.")]:
        del sys.modules[k]          # (an earlier test may have imported it over the oracle)
    ref_shims.install()             # timm / clip / torch_geometric stand-ins + sys.path
    dropin.install(force=True)      # the three native boundaries -> libponderv2_hip.so
    assert dropin.installed()
    import ponder.models  # noqa: F401  the reference's package, unmodified

    assert os.path.realpath(sys.modules["ponder"].__file__).startswith(
        os.path.realpath(ref_shims.REFERENCE_ROOT))
    import spconv.pytorch as sp

    assert sp.__name__ == "ponderv2_amd.spconv.pytorch"
    yield
    for k in [k for k in sys.modules if k == "ponder" or k.startswith("ponder.")]:
        del sys.modules[k]
    for k, v in saved.items():
        if v is not None:
            sys.modules[k] = v


class _TorchDraws:
    """Replays recorded ``torch.rand`` draws (by trailing size, like golden_cases.ReplayRand) and
    hands ``torch.randperm`` the permutations that select the reference's recorded pixels."""

    def __init__(self, rands, perms, device):
        self.replay = gc.ReplayRand(rands, device)
        self.perms = list(perms)
        self.device = device

    def __enter__(self):
        self._rand, self._perm = torch.rand, torch.randperm

        def rand(*shape, **kw):
            shape = shape[0] if len(shape) == 1 and not isinstance(shape[0], int) else shape
            return self.replay(tuple(shape))

        def randperm(n, **kw):
            p = self.perms.pop(0)
            assert p.numel() == n, (p.numel(), n)
            return p.to(kw.get("device") or "cpu")

        torch.rand, torch.randperm = rand, randperm
        return self

    def __exit__(self, *exc):
        torch.rand, torch.randperm = self._rand, self._perm


def test_reference_spunet_runs_unmodified_on_the_product_spconv(device, reference_over_product):
    """spconv_unet_v1m1_base.py:86-278 (the reference's class, from the reference's registry) over
    ponderv2_amd.spconv.pytorch, fp32 on the GPU, against the float64 golden."""
    from ponder.models.builder import MODELS

    g = np.load(os.path.join(gc.GOLDEN, "spunet_small.npz"))
    coords = g["coords"]
    counts = np.bincount(coords[:, 0])
    model = MODELS.build(dict(gc.SMALL_BACKBONE))
    assert type(model).__module__.startswith("ponder.models.sparse_unet")
    fill_deterministic(model)
    model = model.to(device).train()
    feat = formula_tensor("spunet.feat", (len(coords), 6), 1.0).to(device).requires_grad_(True)
    out = model(dict(grid_coord=torch.from_numpy(coords[:, 1:].astype(np.int64)).to(device),
                     feat=feat, offset=torch.from_numpy(np.cumsum(counts)).long().to(device)))
    probe = formula_tensor("spunet.probe", tuple(out.shape), 1.0).to(device)
    (out * probe).sum().backward()
    params = dict(model.named_parameters())
    errs = {"out": gc.rel_err(out, g["out"]), "dfeat": gc.rel_err(feat.grad, g["dfeat"])}
    for i, name in enumerate(g["grad_names"]):
        errs[str(name)] = gc.rel_err(params[str(name)].grad, g[f"grad_{i}"])
    print("zero-edit SpUNet:", {k: "%.2e" % v for k, v in errs.items()})
    assert errs["out"] < 1e-4, errs
    # same bounds as the product model's own test (test_gpu_golden.py): deep closed-form-weight
    # BatchNorm stacks amplify fp32 rounding in the gradients
    assert max(errs.values()) < 6e-2, errs


def test_reference_render_head_runs_unmodified_on_the_product_sampler(device, reference_over_product):
    """The reference's NeuSModel / SDFField (sdf_field.py:148-183 feature_sampling ->
    SmoothSampler.apply, :211-284 with autograd.grad(create_graph=True) through it) over
    ponderv2_amd.smooth_sampler on the GPU, against neus_head.npz."""
    from ponder.models.ponder.render_utils import RayBundle, build_renderer
    from ponderv2_amd.ponder.utils.config import ConfigDict

    g = np.load(os.path.join(gc.GOLDEN, "neus_head.npz"))
    renderer = build_renderer(ConfigDict(gc.RENDERER))
    assert type(renderer).__module__.startswith("ponder.models.ponder.render_utils")
    fill_deterministic(renderer)
    renderer = renderer.to(device).train()
    volume = formula_tensor("neus.volume", (128, 8, 16, 16), 0.6).to(device).requires_grad_(True)
    o = torch.from_numpy(g["origins"]).to(device)
    d = torch.from_numpy(g["directions"]).to(device)
    targets = {k: torch.from_numpy(g[f"tgt_{k}"]).to(device) for k in ("depth", "rgb", "semantic")}
    with _TorchDraws([g["rand0"], g["rand1"]], [], device):
        out = renderer(RayBundle(origins=o, directions=d), [volume])
        losses = renderer.get_loss(out, targets)
    sum(v for k, v in losses.items() if "loss" in k).backward()
    errs = {}
    for k in ("rgb", "semantic", "depth", "normal", "weights", "sdf", "gradients", "z_vals"):
        errs["out_" + k] = gc.rel_err(out[k], g["out_" + k])
    for name, val in zip(g["loss_names"], g["loss_values"]):
        errs["loss_" + str(name)] = abs(float(losses[str(name)].detach()) - val) / (abs(val) + 1e-12)
    errs["dvolume"] = gc.rel_err(volume.grad, g["dvolume"])
    params = dict(renderer.named_parameters())
    for i, name in enumerate(g["grad_names"]):
        errs["grad_" + str(name)] = gc.rel_err(params[str(name)].grad, g[f"grad_{i}"])
    print("zero-edit NeuS head:", {k: "%.2e" % v for k, v in errs.items()})
    losses_only = {k: v for k, v in errs.items() if k.startswith("loss_")}
    assert max(losses_only.values()) < 1e-4, errs      # the north star's bound
    assert max(errs.values()) < 2e-3, errs             # (the product head's own bound on this fixture)


def test_reference_ponder_indoor_runs_unmodified_end_to_end(device, reference_over_product):
    """ponder_indoor_base.py:694-706 - the reference's PonderIndoor.forward, every line of it, with
    all three boundaries on the product: SpUNet over spconv.pytorch, to_dense over
    torch_scatter.scatter (:214), the NeuS head over smooth_sampler - one training step on the GPU
    against the reference's own host run (ponder_indoor_small.npz)."""
    from ponder.models.builder import MODELS
    from ponderv2_amd.ponder.datasets import collate_fn, make_scene
    from ponderv2_amd.ponder.utils.config import ConfigDict

    g = np.load(os.path.join(gc.GOLDEN, "ponder_indoor_small.npz"))
    cfg = gc.indoor_model_cfg(dict(gc.SMALL_BACKBONE, channels=(16, 32, 48, 64, 64, 48, 32, 96)),
                              grid_shape=(32, 32, 8), ray_nsample=20)
    cfg["template"] = ("a", "b")
    model = MODELS.build(ConfigDict(cfg))
    assert type(model).__module__ == "ponder.models.ponder.ponder_indoor_base"
    fill_deterministic(model)
    model = model.to(device).train()
    kw = dict(n_raw=16000, num_views=2, image_hw=(48, 64))
    batch = collate_fn([make_scene(100, **kw), make_scene(101, **kw)])
    # the permutations that pick the recorded pixels: position of each pixel in where(depth > 0)
    # order first, the rest of the indices behind them (ponder_indoor_base.py:546-551)
    perms = []
    B, V, H, W = batch["depth"].shape
    for b in range(B):
        for v in range(V):
            mask = batch["depth"][b, v] > 0
            rank = torch.cumsum(mask.flatten().long(), 0) - 1
            pix = torch.from_numpy(g["ray_pixels"][b, v])
            sel = rank[pix[:, 0] * W + pix[:, 1]]
            rest = torch.ones(int(mask.sum()), dtype=torch.bool)
            rest[sel] = False
            perms.append(torch.cat([sel, torch.nonzero(rest).flatten()]))
    batch = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
    rands = [g[f"rand_{i}"] for i in range(int(g["rands"]))]
    with _TorchDraws(rands, perms, device):
        out = model(batch)
    out["loss"].backward()
    errs = {}
    for name, val in zip(g["out_names"], g["out_values"]):
        errs[str(name)] = abs(float(out[str(name)].detach()) - val) / (abs(val) + 1e-12)
    params = dict(model.named_parameters())
    for i, name in enumerate(g["grad_names"]):
        errs["grad_" + str(name)] = gc.rel_err(params[str(name)].grad, g[f"grad_{i}"])
    print("zero-edit PonderIndoor:", {k: "%.2e" % v for k, v in errs.items()})
    losses = {k: v for k, v in errs.items() if not k.startswith("grad_")}
    assert max(losses.values()) < 1e-4, errs
    assert max(errs.values()) < 6e-2, errs
