"""GPU parity: the HIP-backed product models against the reference-generated golden vectors."""
import pytest
import torch

import golden_cases as gc

pytestmark = pytest.mark.gpu


def test_spunet_gpu_vs_reference_golden(device):
    """Forward to 1e-4.  Gradients: on the GPU the scatter atomics make the fp32 forward vary by
    ~1e-7 from run to run, which occasionally flips one or two ReLU decisions whose pre-activation
    sits at zero; through SpUNet's ~60 BatchNorm layers (some over 34 voxels in this fixture) that
    moves the gradients by up to ~4e-2 (tools/flaky_trace.py: fresh processes land in two clusters,
    1e-6 and 3.8e-2, that first differ in the ReLU mask of one BN-backward call).  The kernels'
    own gradients are held to 1e-5 in test_gpu_kernels.py; here the bound reflects the chain's
    conditioning and the direction of every gradient is checked as well."""
    errs, cos = gc.run_spunet(device, torch.float32)
    print(errs, cos)
    assert errs["out"] < 1e-4, errs
    assert max(errs.values()) < 0.15, errs
    assert max(cos.values()) < 5e-3, cos


def test_neus_head_gpu_vs_reference_golden(device):
    errs = gc.run_neus(device)
    print(errs)
    outs = {k: v for k, v in errs.items() if k.startswith(("out_", "loss_"))}
    assert max(outs.values()) < 1e-4, errs
    assert max(errs.values()) < 2e-3, errs


def test_ponder_indoor_gpu_vs_reference_golden(device):
    errs = gc.run_ponder_indoor(device)
    print(errs)
    # the north-star bound: loss within 1e-4 relative (measured 4e-6).  The stem weight gradient
    # sits at the end of a backward chain through ~60 BatchNorm layers, some over a few dozen
    # voxels: on the CPU oracle a 1e-7 relative perturbation of the input features moves it by
    # 3e-2 (and the dec.0 gradient by 1e-4) while the loss moves by 2e-7 - hence the separate
    # bound for backbone-chain gradients; every other probe is held to 1e-3.  flip_tol: see
    # golden_cases.check_model_errors (one importance sample landing in the neighbouring bin).
    gc.check_model_errors(errs, rest_tol=1e-3, flip_tol=5e-3)


def test_graphed_render_head_equals_eager(device):
    """hipGraph replay of the render head (forward + backward) == the eager path, with the samplers'
    jitter switched off so both are deterministic; three steps to cover capture, replay, replay with
    updated inputs."""
    import copy

    from oracle.detweights import fill_deterministic
    from ponderv2_amd.ponder.datasets import collate_fn, make_scene
    from ponderv2_amd.ponder.models import build_model
    from ponderv2_amd.ponder.utils.config import ConfigDict

    cfg = gc.indoor_model_cfg(dict(gc.SMALL_BACKBONE, channels=(16, 32, 48, 64, 64, 48, 32, 96)),
                              grid_shape=(32, 32, 8), ray_nsample=24)
    cfg["renderer"] = copy.deepcopy(cfg["renderer"])
    cfg["renderer"]["sampler"]["train_stratified"] = False
    kw = dict(n_raw=16000, num_views=2, image_hw=(48, 64))
    batches = [collate_fn([make_scene(200 + 2 * i, **kw), make_scene(201 + 2 * i, **kw)])
               for i in range(3)]
    results = {}
    for graphed in (False, True):
        cfg["graph_render_head"] = graphed
        model = build_model(ConfigDict(cfg))
        fill_deterministic(model)
        model = model.to(device).train()
        rows = []
        for i, b in enumerate(batches):
            torch.manual_seed(100 + i)  # same pixel choice in both runs
            b = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in b.items()}
            model.zero_grad(set_to_none=True)
            out = model(b)
            out["loss"].backward()
            grads = [p.grad.clone() for p in model.renderer.parameters() if p.grad is not None]
            rows.append((float(out["loss"]), float(out["eikonal_loss"]),
                         model.proj_net.final_conv.weight.grad.clone(), *grads))
        if graphed:
            assert model._graphed is not None and not model._graphed.failed, "capture was refused"
        results[graphed] = rows
    for e, g in zip(results[False], results[True]):
        assert abs(e[0] - g[0]) < 1e-5 * abs(e[0]) and abs(e[1] - g[1]) < 1e-4 * abs(e[1]) + 1e-9
        assert len(e) == len(g) > 20  # every renderer parameter (weights AND biases) is compared
        for a, b in zip(e[2:], g[2:]):
            assert (a - b).abs().max() <= 2e-4 * a.abs().max() + 1e-12


def test_ponder_outdoor_gpu_vs_reference_golden(device):
    """PonderOutdoor-v2 (reference ponder_outdoor_base.py run on the host with the same weights,
    mask draws and sampler jitter): depth loss to 1e-4, gradients from the mask token to the
    variance network."""
    errs = gc.run_ponder_outdoor(device)
    print(errs)
    # measured on MI355X: loss 3e-6, backbone-chain gradients 2-3e-3, the rest <= 2e-4
    gc.check_model_errors(errs, rest_tol=1e-3, flip_tol=2e-3)


def test_outdoor_graphed_render_head_equals_eager(device):
    """Same check as the indoor one for the outdoor head (flat ray arrays with ray_offset, depth
    loss only, 5-block 16-wide SDF MLP): graph replay == eager over three different batches; a
    ragged batch must fall back to the per-scene path and still train."""
    import copy

    from oracle.detweights import fill_deterministic
    from ponderv2_amd.ponder.datasets import lidar_collate_fn, make_lidar_scene
    from ponderv2_amd.ponder.models import build_model
    from ponderv2_amd.ponder.utils.config import ConfigDict

    cfg = gc.outdoor_model_cfg(dict(gc.SMALL_BACKBONE, in_channels=4,
                                    channels=(16, 32, 48, 64, 64, 48, 32, 96)), **gc.OUTDOOR_SMALL)
    cfg["renderer"] = copy.deepcopy(cfg["renderer"])
    cfg["renderer"]["sampler"]["train_stratified"] = False
    batches = [lidar_collate_fn([make_lidar_scene(400 + 2 * i, **gc.OUTDOOR_SCENE_KW),
                                 make_lidar_scene(401 + 2 * i, **gc.OUTDOOR_SCENE_KW)])
               for i in range(3)]
    results = {}
    for graphed in (False, True):
        cfg["graph_render_head"] = graphed
        model = build_model(ConfigDict(cfg))
        fill_deterministic(model)
        model = model.to(device).train()
        rows = []
        for i, b in enumerate(batches):
            torch.manual_seed(100 + i)  # same block mask in both runs
            b = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in b.items()}
            model.zero_grad(set_to_none=True)
            out = model(b)
            out["loss"].backward()
            grads = [p.grad.clone() for p in model.renderer.parameters() if p.grad is not None]
            rows.append((float(out["loss"]), model.proj_net.conv[0].weight.grad.clone(),
                         model.mtoken.grad.clone(), *grads))
        if graphed:
            assert model._graphed is not None and not model._graphed.failed, "capture was refused"
        results[graphed] = rows
    for e, g in zip(results[False], results[True]):
        assert abs(e[0] - g[0]) < 1e-5 * abs(e[0])
        assert len(e) == len(g) > 10
        # both runs go through the backbone's float atomics, so the volume the two heads see
        # already differs at the 1e-6 level; measured head-gradient differences: <= 2.2e-4
        for a, b in zip([e[1]] + list(e[3:]), [g[1]] + list(g[3:])):
            assert (a - b).abs().max() <= 2e-3 * a.abs().max() + 1e-12
        # e[2] / g[2] (the mask token's gradient) are not compared: it is the far end of the
        # backbone's backward chain, where the run-to-run atomic noise of EITHER mode is amplified
        # to tens of percent in this 80 %-masked miniature scene (conditioning note in
        # test_spunet_gpu_vs_reference_golden); the projection-conv gradient above already shows
        # that both modes hand the same volume gradient to everything upstream
        assert torch.isfinite(g[2]).all()
    # ragged batch: drop 5 rays of the second sweep -> per-scene rendering, no graph
    b = dict(batches[0])
    n = int(b["ray_offset"][-1]) - 5
    b["ray_start"], b["ray_end"] = b["ray_start"][:n], b["ray_end"][:n]
    b["ray_offset"] = torch.tensor([int(b["ray_offset"][0]), n])
    b["ray_offset_host"] = [int(v) for v in b["ray_offset"]]
    b = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in b.items()}
    model.zero_grad(set_to_none=True)
    out = model(b)
    out["loss"].backward()
    assert torch.isfinite(out["loss"]) and torch.isfinite(model.mtoken.grad).all()


def test_spunet_pdnorm_gpu_vs_reference_golden(device):
    """SpUNet-v1m3 on the GPU: the per-condition modulation rides in the fused BatchNorm kernel's
    affine epilogue.  Same bounds (and conditioning caveat) as the v1m1 backbone test."""
    errs, cos = gc.run_spunet_pdnorm(device, torch.float32)
    print(errs, cos)
    assert errs["out"] < 1e-4, errs
    assert max(errs.values()) < 0.15, errs
    assert max(cos.values()) < 5e-3, cos


@pytest.mark.xfail(strict=False, reason="open issue (DESIGN.md section 6): with the full-size model and an "
                   "optimizer step between replays, the graphed head's sdf / free-space / eikonal loss "
                   "terms were observed to turn into garbage on some steps while the eager head trains "
                   "smoothly (profiles/r01_graph_head_loss_trace.txt); the head is eager by default")
def test_graphed_head_training_trajectory_equals_eager(device):
    """Six SGD steps of the full-size indoor model on one batch, eager head vs graph-replayed head,
    sampler jitter off: every logged loss term must follow the same trajectory."""
    import copy

    import bench
    from ponderv2_amd.ponder.models import build_model
    from ponderv2_amd.ponder.utils.config import ConfigDict

    cfg = bench.model_cfg(256)
    cfg["renderer"] = copy.deepcopy(cfg["renderer"])
    cfg["renderer"]["sampler"]["train_stratified"] = False
    batch = bench.make_batch(0, 2, 2, device)
    traces = {}
    for graphed in (False, True):
        cfg["graph_render_head"] = graphed
        torch.manual_seed(0)
        model = build_model(ConfigDict(cfg)).to(device).train()
        opt = torch.optim.SGD(model.parameters(), lr=1.25e-4, momentum=0.9, weight_decay=1e-4,
                              nesterov=True)
        rows = []
        for step in range(6):
            torch.manual_seed(100 + step)  # same pixel choice in both runs
            out = model(bench.clone_batch(batch))
            opt.zero_grad(set_to_none=True)
            out["loss"].backward()
            opt.step()
            rows.append({k: float(v.detach()) for k, v in out.items()})
        traces[graphed] = rows
    for e, g in zip(traces[False], traces[True]):
        for k in e:
            assert abs(e[k] - g[k]) <= 2e-2 * abs(e[k]) + 1e-3, (k, traces)


@pytest.mark.xfail(strict=False, reason="added after the round's GPU budget was spent; first hardware run is "
                   "the round-end driver's (the CPU twin passes at 3e-7 on the loss)")
def test_ponder_ppt_gpu_vs_reference_golden(device):
    """Multi-condition PonderIndoor over the PDNorm backbone on the GPU (batched modulation GEMV,
    fused BatchNorm epilogue) against the reference run on the host."""
    errs = gc.run_ponder_ppt(device)
    print(errs)
    gc.check_model_errors(errs, rest_tol=2e-3, flip_tol=5e-3)
