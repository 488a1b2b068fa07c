"""The reduced-precision training mode on MI355X: 16-bit (bf16 / fp16) sparse-conv and BatchNorm
kernels against the fp64 oracle evaluated on the SAME 16-bit-rounded operands, so that what is
measured is the kernels' own error (fp32 accumulation order + one rounding of the result), not the
quantisation of the inputs.  Reference mode: enable_amp=True,
configs/scannet/pretrain-ponder-spunet-v1m1-0-base.py:12, ponder/engines/train.py:183-196."""
import numpy as np
import pytest
import torch

from helpers import random_voxels

pytestmark = pytest.mark.gpu

# one rounding of the result: half an ulp relative to the element, measured relative to the
# tensor's maximum (plus fp32 accumulation noise)
OUT_TOL = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}


def _oracle_conv(feats, w, pin, pout, ks, n_out):
    from oracle.sparse_ops import sparse_conv

    return sparse_conv(feats, w, torch.from_numpy(pin.astype(np.int64)),
                       torch.from_numpy(pout.astype(np.int64)), ks, n_out)


def _rel(a, b):
    return (a.double().cpu() - b).abs().max().item() / (b.abs().max().item() + 1e-12)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("c_in,c_out,ksize", [(32, 32, 3), (32, 64, 3), (96, 96, 3), (128, 96, 1),
                                               (384, 256, 3), (256, 256, 3), (64, 40, 3), (96, 128, 1),
                                               (192, 128, 3)])
def test_spconv16_forward_backward_vs_oracle(device, dtype, c_in, c_out, ksize):
    from oracle import rulebook as orb
    from ponderv2_amd import kernels as K

    torch.manual_seed(c_in * 1000 + c_out)
    coords = random_voxels(5, batch=2, n_per_batch=700)
    n = len(coords)
    feats = torch.randn(n, c_in).to(dtype)
    w = (torch.randn(c_out, ksize ** 3, c_in) * 0.1).to(dtype).float()  # exactly representable
    bias = torch.randn(c_out) * 0.1
    gout = torch.randn(n, c_out).to(dtype)
    pin, pout, ks = orb.subm_rulebook(coords, ksize)

    f_ref = feats.double().requires_grad_(True)
    w_ref = w.double().requires_grad_(True)
    ref = _oracle_conv(f_ref, w_ref, pin, pout, ks, n) + bias.double()
    ref.backward(gout.double())

    rb = K.build_subm_rulebook(torch.from_numpy(coords).to(device), ksize)
    f_dev = feats.to(device).requires_grad_(True)
    w_dev = w.to(device).requires_grad_(True)
    b_dev = bias.to(device).requires_grad_(True)
    assert K.spconv16_supported(f_dev, w_dev, rb)
    out = K.SparseConv16Function.apply(f_dev, w_dev, rb, b_dev, {})
    assert out.dtype == dtype
    out.backward(gout.to(device))
    assert f_dev.grad.dtype == dtype and w_dev.grad.dtype == torch.float32

    assert _rel(out, ref.detach()) < OUT_TOL[dtype]
    assert _rel(f_dev.grad, f_ref.grad) < OUT_TOL[dtype]
    assert _rel(w_dev.grad, w_ref.grad) < 2e-5     # fp32 result: accumulation order only
    assert _rel(b_dev.grad, gout.double().sum(0)) < 2e-5


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_spconv16_strided_and_inverse_vs_oracle(device, dtype):
    from oracle import rulebook as orb
    from ponderv2_amd import kernels as K

    torch.manual_seed(7)
    coords = random_voxels(6, batch=2, n_per_batch=2500)
    n = len(coords)
    shape = [68, 66, 58]
    ooc, pin, pout, ks = orb.downsample_rulebook(coords, 2, shape)
    m = len(ooc)
    c_in, c_out = 32, 64
    feats = torch.randn(n, c_in).to(dtype)
    w = (torch.randn(c_out, 8, c_in) * 0.1).to(dtype).float()
    w_inv = (torch.randn(c_in, 8, c_out) * 0.1).to(dtype).float()
    g_up = torch.randn(n, c_in).to(dtype)
    rb, _ = K.build_downsample_rulebook(torch.from_numpy(coords).to(device), 2, shape)

    f_dev = feats.to(device).requires_grad_(True)
    w_dev, wi_dev = w.to(device).requires_grad_(True), w_inv.to(device).requires_grad_(True)
    down = K.SparseConv16Function.apply(f_dev, w_dev, rb, None, {})
    up = K.SparseConv16Function.apply(down, wi_dev, rb.transposed(), None, {})
    up.backward(g_up.to(device))

    f_ref = feats.double().requires_grad_(True)
    w_ref, wi_ref = w.double().requires_grad_(True), w_inv.double().requires_grad_(True)
    ref_down = _oracle_conv(f_ref, w_ref, pin, pout, ks, m)
    assert _rel(down, ref_down.detach()) < OUT_TOL[dtype]
    # second stage on the kernel's own (rounded) intermediate, as the kernel saw it
    d16 = down.detach().double().cpu().requires_grad_(True)
    ref_up = _oracle_conv(d16, wi_ref, pout, pin, ks, n)
    ref_up.backward(g_up.double())
    assert _rel(up, ref_up.detach()) < OUT_TOL[dtype]
    assert _rel(wi_dev.grad, wi_ref.grad) < 2e-5
    # gradient reaching `down` is rounded to 16 bits before it travels on: compare one stage
    g_down = d16.grad.to(dtype).double()
    ref_down.backward(g_down)
    assert _rel(w_dev.grad, w_ref.grad) < 3 * OUT_TOL[dtype]  # through the rounded g_down
    assert _rel(f_dev.grad, f_ref.grad) < 3 * OUT_TOL[dtype]


def test_packed_weights_follow_the_parameter_version(device):
    from ponderv2_amd import kernels as K

    w = torch.randn(32, 27, 32, device=device)
    cache = {}
    a, _ = K.packed_weights(w, torch.bfloat16, cache)
    b, _ = K.packed_weights(w, torch.bfloat16, cache)
    assert a.data_ptr() == b.data_ptr()
    w.mul_(2.0)  # an optimiser step moves the version counter
    c, _ = K.packed_weights(w, torch.bfloat16, cache)
    assert torch.equal(c.float(), a.float() * 2)


@pytest.mark.parametrize("x_dtype,y_dtype", [(torch.float32, torch.bfloat16), (torch.bfloat16, torch.bfloat16),
                                             (torch.float32, torch.float16), (torch.float16, torch.float16)])
@pytest.mark.parametrize("n,c", [(37, 48), (5000, 32), (46842, 96), (989, 256)])
@pytest.mark.parametrize("relu,with_res", [(False, False), (True, True)])
def test_fused_bn_16bit_matches_torch(device, x_dtype, y_dtype, n, c, relu, with_res):
    import torch.nn as nn

    from ponderv2_amd import precision
    from ponderv2_amd.rownorm import fused_bn

    torch.manual_seed(n + c)
    x = (torch.randn(n, c) * 2 + 0.5).to(x_dtype)
    res = torch.randn(n, c).to(y_dtype) if with_res else None
    gout = torch.randn(n, c).to(y_dtype)

    def make(dev, dtype):
        bn = nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(dev).to(dtype).train()
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 1.5, c))
            bn.bias.copy_(torch.linspace(-0.2, 0.2, c))
        return bn

    bn_ref = make("cpu", torch.float64)
    xr = x.double().requires_grad_(True)
    rr = res.double().requires_grad_(True) if with_res else None
    y = bn_ref(xr)
    if with_res:
        y = y + rr
    if relu:
        y = torch.relu(y)
    y.backward(gout.double())

    bn = make(device, torch.float32)
    xi = x.to(device).requires_grad_(True)
    ri = res.to(device).requires_grad_(True) if with_res else None
    with precision.sparse_activations(y_dtype):
        out = fused_bn(bn, xi, residual=ri, relu=relu)
    assert out.dtype == y_dtype
    out.backward(gout.to(device))
    assert xi.grad.dtype == x_dtype

    tol = OUT_TOL[y_dtype]
    assert _rel(out, y.detach()) < tol
    # the ReLU mask is taken from the ROUNDED output: elements whose exact value is within rounding
    # of zero may differ - compare where the reference output is clear of zero
    clear = (y.detach().abs() > 4 * tol * y.detach().abs().max()) | (not relu)
    gx = xi.grad.double().cpu()
    scale = xr.grad.abs().max().item() + 1e-12
    frac_bad = (((gx - xr.grad).abs() > 6 * max(tol, OUT_TOL[x_dtype] if x_dtype != torch.float32 else 0) * scale)
                & clear).double().mean().item()
    assert frac_bad < 2e-3, frac_bad
    assert _rel(bn.weight.grad, bn_ref.weight.grad) < 2e-2
    assert _rel(bn.bias.grad, bn_ref.bias.grad) < 2e-2
    assert _rel(bn.running_mean, bn_ref.running_mean) < 1e-5
    assert _rel(bn.running_var, bn_ref.running_var) < 1e-5
    if with_res:
        assert ri.grad.dtype == y_dtype


def test_spunet_16bit_mode_tracks_fp32(device, monkeypatch):
    """The whole sparse U-Net with 16-bit activations against its fp32 self: outputs within the
    mode's noise, every parameter gets a finite gradient, forward bitwise reproducible."""
    from golden_cases import FULL_BACKBONE  # noqa: F401  (the BASELINE backbone config)
    from ponderv2_amd import kernels as K
    from ponderv2_amd import precision
    from ponderv2_amd.ponder.models import build_model


    torch.manual_seed(0)
    cfg = dict(FULL_BACKBONE)
    model = build_model(cfg).to(device).train()
    coords = random_voxels(11, batch=2, n_per_batch=6000)
    grid = torch.from_numpy(coords[:, 1:]).to(device)
    counts = np.bincount(coords[:, 0], minlength=2)
    data = dict(grid_coord=grid, feat=torch.randn(len(coords), cfg["in_channels"], device=device),
                offset=torch.from_numpy(np.cumsum(counts)).to(device))

    ref = model(dict(data))
    with precision.sparse_activations(torch.bfloat16):
        out = model(dict(data))
        out2 = model(dict(data))
    assert out.dtype == torch.bfloat16 and torch.equal(out, out2)
    err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 0.1, err
    cos = torch.nn.functional.cosine_similarity(out.float().flatten(), ref.flatten(), dim=0).item()
    assert cos > 0.995, cos
    # a random linear functional of the output: its gradient is not aligned with what the
    # BatchNorm backward projects out (a loss like sum(out^2) is, and then measures nothing but
    # the rounding of that cancellation)
    probe = torch.randn_like(ref)
    model.zero_grad()
    (model(dict(data)) * probe).sum().backward()
    g32 = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    assert len(g32) > 100
    medians = {}
    for dtype in (torch.float16, torch.bfloat16):
        model.zero_grad()
        with precision.sparse_activations(dtype):
            (model(dict(data)).float() * probe).sum().backward()
        cosines = []
        for k, p in model.named_parameters():
            if p.grad is None:
                continue
            assert torch.isfinite(p.grad).all(), k
            if p.grad.numel() > 64 and g32[k].abs().max() > 0:
                cosines.append(torch.nn.functional.cosine_similarity(
                    p.grad.flatten(), g32[k].flatten(), dim=0).item())
        cosines = np.sort(np.array(cosines))
        medians[dtype] = cosines[len(cosines) // 2]
        print({str(dtype) + " grad cosine min / p10 / median": (cosines[0], cosines[len(cosines) // 10],
                                                                medians[dtype])})
    # Rounding the activations of ~45 randomly initialised conv-BN-ReLU layers perturbs the
    # gradient direction in proportion to the rounding step: fp16 (11 bits) must sit close to the
    # fp32 gradient, bf16 (8 bits) further out but still pointing the same way.
    assert medians[torch.float16] > 0.97 and medians[torch.bfloat16] > 0.8


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_native_executor_runs_the_16bit_mode(device, dtype):
    """Round 6: under ``precision.sparse_activations`` the whole sparse U-Net is ONE
    native call per direction (``UNET_CONV_BN16`` records, csrc/spunet_exec.hip) - the same kernels in
    the same order as the module-by-module walk: forward bits identical, parameter gradients equal up to
    the order in which 16-bit gradients of an activation with several consumers are added and the float
    atomics of the 16-bit weight gradient."""
    from golden_cases import FULL_BACKBONE
    from ponderv2_amd import precision, spunet_native
    from ponderv2_amd.ponder.models import build_model

    torch.manual_seed(0)
    cfg = dict(FULL_BACKBONE)
    model = build_model(cfg).to(device).train()
    coords = random_voxels(11, batch=2, n_per_batch=6000)
    grid = torch.from_numpy(coords[:, 1:]).to(device)
    counts = np.bincount(coords[:, 0], minlength=2)
    data = dict(grid_coord=grid, feat=torch.randn(len(coords), cfg["in_channels"], device=device),
                offset=torch.from_numpy(np.cumsum(counts)).to(device))
    probe = None
    res = {}
    for native in (True, False):
        spunet_native.NATIVE16 = native
        try:
            model.zero_grad()
            calls = spunet_native.CALLS
            with precision.sparse_activations(dtype):
                out = model(dict(data))
            assert (spunet_native.CALLS > calls) == native
            assert out.dtype == dtype
            if probe is None:
                probe = torch.randn(out.shape, device=device)
            (out.float() * probe).sum().backward()
            res[native] = (out.detach().clone(),
                           {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
        finally:
            spunet_native.NATIVE16 = True
    assert torch.equal(res[True][0], res[False][0])        # same kernels, same order: same bits
    assert res[True][1].keys() == res[False][1].keys() and len(res[True][1]) > 100
    worst = 1.0
    for k, g in res[True][1].items():
        h = res[False][1][k]
        assert torch.isfinite(g).all(), k
        if g.numel() > 64 and h.abs().max() > 0:
            worst = min(worst, torch.nn.functional.cosine_similarity(g.flatten(), h.flatten(), dim=0).item())
    assert worst > 0.99, worst


def test_fp16_gradscaler_step_through_an_overflow(device):
    """The reference's training step (ponder/engines/train.py:183-196: autocast + GradScaler) in fp16
    THROUGH a real overflow: with a loss scale that pushes the backbone's 16-bit
    gradients past fp16's range the native 16-bit executor must hand the scaler non-finite parameter
    gradients (Inf or NaN - an isfinite check cannot tell, csrc/mfma_split.h), the scaler skips the step
    and halves the scale, parameters stay untouched; once the scale fits, steps are taken and every
    parameter stays finite."""
    import golden_cases as gc
    import ddp_worker
    from ponderv2_amd import spunet_native
    from ponderv2_amd.ponder.datasets import collate_fn
    from ponderv2_amd.ponder.models import build_model
    from ponderv2_amd.ponder.utils.config import ConfigDict

    torch.manual_seed(5)
    cfg = gc.indoor_model_cfg(dict(gc.SMALL_BACKBONE, base_channels=32, channels=(32, 32, 64, 64, 64, 64, 32, 96)),
                              grid_shape=(32, 32, 8), ray_nsample=6)
    model = build_model(ConfigDict(cfg)).to(device).train()
    batch = collate_fn([ddp_worker.tiny_scene(60), ddp_worker.tiny_scene(61)])
    batch = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
    opt = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.9)
    scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 30, growth_interval=1000)
    backbone = [p for n, p in model.named_parameters() if n.startswith("backbone.") and p.requires_grad]
    skipped = taken = 0
    for step in range(40):
        before = [p.detach().clone() for p in backbone[:8]]
        opt.zero_grad(set_to_none=True)
        calls = spunet_native.CALLS
        with torch.autocast("cuda", dtype=torch.float16):
            out = model({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
        assert spunet_native.CALLS == calls + 1       # the 16-bit units ran in the native executor
        scale = scaler.get_scale()
        scaler.scale(out["loss"]).backward()
        scaler.step(opt)
        scaler.update()
        moved = any(not torch.equal(a, p.detach()) for a, p in zip(before, backbone[:8]))
        if scaler.get_scale() < scale:                # an overflow: the step was skipped
            skipped += 1
            assert not moved
            assert taken == 0                         # (overflows come first, while the scale is too large)
        else:
            taken += 1
            assert moved
        assert all(torch.isfinite(p).all() for p in model.parameters())
        if taken >= 3:
            break
    assert skipped >= 1 and taken >= 3, (skipped, taken, scaler.get_scale())
