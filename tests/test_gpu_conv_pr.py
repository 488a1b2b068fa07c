"""GPU tests of the product-row conv path (csrc/sparse_conv_pr.hip), the deterministic weight gradient
and the fused conv + BatchNorm units (ponderv2_amd/convbn.py): against the fp64 oracle, against the
scatter-add kernels, and bit for bit against themselves.  Run with -m gpu on an MI355X."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from helpers import random_voxels

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


def _rel(a, b):
    return (a.double().cpu() - b.double().cpu()).abs().max().item() / (b.abs().max().item() + 1e-12)


def _positions_reference(pair_row, kstart, K, n_rows):
    pos = -np.ones((K, n_rows), np.int64)
    for k in range(K):
        p = np.arange(kstart[k], kstart[k + 1])
        pos[k, pair_row[p]] = p
    return pos


# ------------------------------------------------------------------ position tables
@pytest.mark.parametrize("seed,batch,n", [(0, 2, 1500), (1, 1, 4000), (2, 3, 3)])
def test_pair_position_tables(device, seed, batch, n):
    from ponderv2_amd import kernels as K

    coords = random_voxels(seed, batch=batch, n_per_batch=n)
    dev_coords = torch.from_numpy(coords).to(device)
    shape = [(s - 2) // 2 + 1 for s in (40 + 96, 36 + 96, 20 + 96)]
    rb3 = K.build_subm_rulebook(dev_coords, 3)
    rbd, _ = K.build_downsample_rulebook(dev_coords, 2, shape)
    for rb in (rb3, rbd, rbd.transposed()):
        pos_out, so, pos_in, si = rb.positions()
        pin, pout, ks = _np(rb.pair_in), _np(rb.pair_out), rb.kstart_host
        assert np.array_equal(_np(pos_out).reshape(rb.K, so)[:, :rb.n_out],
                              _positions_reference(pout, ks, rb.K, rb.n_out))
        assert np.array_equal(_np(pos_in).reshape(rb.K, si)[:, :rb.n_in],
                              _positions_reference(pin, ks, rb.K, rb.n_in))


# ------------------------------------------------------------------ reproducibility
@pytest.mark.parametrize("c_in,c_out", [(32, 32), (64, 64), (96, 96), (128, 128), (256, 192), (384, 256)])
def test_product_row_conv_is_bitwise_reproducible_and_equals_the_scatter_kernels(
        device, c_in, c_out, monkeypatch):
    """Forward, grad-input and weight gradient of a submanifold conv on the default path: identical
    bits on every call, and the scatter-add kernels' numbers up to fp32 re-association."""
    from ponderv2_amd import kernels as K

    assert K.USE_PR == "all" and K.USE_WGRAD_DET
    torch.manual_seed(c_in + c_out)
    coords = random_voxels(8, batch=2, n_per_batch=2500)
    n = len(coords)
    rb = K.build_subm_rulebook(torch.from_numpy(coords).to(device), 3)
    x = torch.randn(n, c_in, device=device)
    w = torch.randn(c_out, 27, c_in, device=device) * 0.1
    g = torch.randn(n, c_out, device=device)
    bias = torch.randn(c_out, device=device)

    def run():
        return (K.spconv_forward(x, w, rb, bias=bias), K.spconv_grad_input(g, w, rb),
                K.spconv_backward_weight(x, g, rb, c_out))

    runs = [run() for _ in range(3)]
    for other in runs[1:]:
        for a, b in zip(runs[0], other):
            assert torch.equal(a, b)
    monkeypatch.setattr(K, "USE_PR", False)
    monkeypatch.setattr(K, "USE_OS", False)
    monkeypatch.setattr(K, "USE_WGRAD_DET", False)
    ref = (K.spconv_forward(x, w, rb) + bias, K.spconv_grad_input(g, w, rb),
           K.spconv_backward_weight(x, g, rb, c_out))
    for a, b in zip(runs[0], ref):
        assert _rel(a, b) < 2e-5


def test_product_row_strided_and_inverse_conv(device, monkeypatch):
    """Strided conv, the inverse conv over the same pairs and their gradients on the product-row
    path: equal the scatter-add kernels, bitwise repeatable."""
    from ponderv2_amd import kernels as K

    torch.manual_seed(11)
    coords = random_voxels(9, batch=2, n_per_batch=2500)
    shape = [68, 66, 58]
    rb, _ = K.build_downsample_rulebook(torch.from_numpy(coords).to(device), 2, shape)
    n, m = len(coords), rb.n_out
    x = torch.randn(n, 32, device=device)
    w = torch.randn(64, 8, 32, device=device) * 0.1
    w_inv = torch.randn(96, 8, 64, device=device) * 0.1
    g_up = torch.randn(n, 96, device=device)

    def run():
        down = K.spconv_forward(x, w, rb)
        up = K.spconv_forward(down, w_inv, rb.transposed())
        return (down, up, K.spconv_grad_input(g_up, w_inv, rb.transposed()),
                K.spconv_backward_weight(down, g_up, rb.transposed(), 96))

    first, second = run(), run()
    assert first[0].shape == (m, 64) and first[1].shape == (n, 96) and first[2].shape == (m, 64)
    for a, b in zip(first, second):
        assert torch.equal(a, b)
    monkeypatch.setattr(K, "USE_PR", False)
    monkeypatch.setattr(K, "USE_OS", False)
    monkeypatch.setattr(K, "USE_WGRAD_DET", False)
    for a, b in zip(first, run()):
        assert _rel(a, b) < 2e-5


# ------------------------------------------------------------------ fused conv + BatchNorm units
class _Unit(nn.Module):
    def __init__(self, kind, c_in, c_out):
        super().__init__()
        from ponderv2_amd.spconv import pytorch as spconv

        if kind == "subm":
            self.conv = spconv.SubMConv3d(c_in, c_out, 3, padding=1, bias=False, indice_key="s")
        elif kind == "subm1":
            self.conv = spconv.SubMConv3d(c_in, c_out, 1, bias=False)
        elif kind == "down":
            self.conv = spconv.SparseConv3d(c_in, c_out, 2, stride=2, bias=False, indice_key="d")
        else:
            self.pre = spconv.SparseConv3d(c_in, c_in, 2, stride=2, bias=False, indice_key="d")
            self.conv = spconv.SparseInverseConv3d(c_in, c_out, 2, bias=False, indice_key="d")
        self.kind = kind
        self.bn = nn.BatchNorm1d(c_out, eps=1e-3, momentum=0.01)
        nn.init.uniform_(self.bn.weight, 0.5, 1.5)
        nn.init.uniform_(self.bn.bias, -0.5, 0.5)

    def forward(self, x, residual=None, relu=True):
        if self.kind == "inverse":
            x = self.pre(x)
        return self.conv.forward_bn(x, self.bn, residual=residual, relu=relu)


def _reference_unit(unit, rb_arrays, feats, residual, relu, gout, n_out):
    """fp64: oracle sparse conv -> training-mode batch norm -> (+res) -> relu, with autograd."""
    from oracle.sparse_ops import sparse_conv

    pin, pout, ks = rb_arrays
    w = unit.conv.weight.detach().double().cpu().reshape(unit.conv.out_channels, -1,
                                                          unit.conv.in_channels).requires_grad_(True)
    x = feats.detach().double().cpu().requires_grad_(True)
    gamma = unit.bn.weight.detach().double().cpu().requires_grad_(True)
    beta = unit.bn.bias.detach().double().cpu().requires_grad_(True)
    res = residual.detach().double().cpu().requires_grad_(True) if residual is not None else None
    y = sparse_conv(x, w, torch.from_numpy(pin.astype(np.int64)), torch.from_numpy(pout.astype(np.int64)),
                    ks, n_out)
    out = F.batch_norm(y, None, None, gamma, beta, True, 0.0, unit.bn.eps)
    if res is not None:
        out = out + res
    if relu:
        out = F.relu(out)
    out.backward(gout.double().cpu())
    return out.detach(), x.grad, w.grad, gamma.grad, beta.grad, (res.grad if res is not None else None), y


@pytest.mark.parametrize("kind,c_in,c_out,with_res,relu", [
    ("subm", 32, 32, False, True), ("subm", 64, 64, True, True), ("subm", 128, 128, True, True),
    ("subm", 384, 256, False, True), ("subm", 96, 96, True, False), ("subm1", 192, 128, False, False),
    ("down", 32, 64, False, True), ("down", 128, 256, False, True)])
def test_fused_conv_bn_unit_vs_fp64_reference_and_the_modular_path(device, kind, c_in, c_out,
                                                                   with_res, relu, monkeypatch):
    from oracle import rulebook as orb
    from ponderv2_amd import convbn, kernels as K
    from ponderv2_amd.spconv import pytorch as spconv

    torch.manual_seed(c_in * 3 + c_out)
    coords = random_voxels(4, batch=2, n_per_batch=1800)
    n = len(coords)
    shape = [68, 66, 58]
    if kind == "down":
        ooc, pin, pout, ks = orb.downsample_rulebook(coords, 2, [(s - 2) // 2 + 1 for s in shape])
        n_out = len(ooc)
    else:
        pin, pout, ks = orb.subm_rulebook(coords, 1 if kind == "subm1" else 3)
        n_out = n
    unit = _Unit(kind, c_in, c_out).to(device).train()
    feats = torch.randn(n, c_in, device=device)
    residual = torch.randn(n_out, c_out, device=device) if with_res else None
    gout = torch.randn(n_out, c_out, device=device)

    def run(fused):
        monkeypatch.setattr(K, "USE_CONVBN", fused)
        calls = []
        orig = convbn.ConvBNFunction.apply
        monkeypatch.setattr(convbn.ConvBNFunction, "apply",
                            staticmethod(lambda *a: calls.append(1) or orig(*a)))
        for p in unit.parameters():
            p.grad = None
        unit.bn.running_mean.zero_()
        unit.bn.running_var.fill_(1.0)
        f = feats.clone().requires_grad_(True)
        r = residual.clone().requires_grad_(True) if with_res else None
        x = spconv.SparseConvTensor(f, torch.from_numpy(coords).to(device), shape, 2)
        out = unit(x, residual=r, relu=relu).features
        out.backward(gout)
        assert bool(calls) == fused
        return (out.detach(), f.grad, unit.conv.weight.grad.reshape(c_out, -1, c_in).clone(),
                unit.bn.weight.grad.clone(), unit.bn.bias.grad.clone(),
                r.grad if with_res else None, unit.bn.running_mean.clone(), unit.bn.running_var.clone())

    fused_a, fused_b, modular = run(True), run(True), run(False)
    for a, b in zip(fused_a, fused_b):   # bitwise reproducible, gradients included
        assert a is None or torch.equal(a, b)
    ref = _reference_unit(unit, (pin, pout, ks), feats, residual, relu, gout, n_out)
    names = ("out", "dx", "dw", "dgamma", "dbeta", "dres")
    for name, got, mod, r in zip(names, fused_a[:6], modular[:6], ref[:6]):
        if got is None:
            continue
        assert _rel(got, r) < 3e-5, (name, _rel(got, r))
        assert _rel(got, mod) < 3e-5, (name, _rel(got, mod))
    # running statistics: (1 - momentum) * init + momentum * batch statistic (unbiased variance)
    y = ref[6].detach()
    assert _rel(fused_a[6], 0.01 * y.mean(0)) < 1e-5
    assert _rel(fused_a[7], 0.99 + 0.01 * y.var(0, unbiased=True)) < 1e-5


def test_fused_inverse_conv_unit(device, monkeypatch):
    """SparseInverseConv3d + BatchNorm + ReLU through the fused unit == the modular path."""
    from ponderv2_amd import kernels as K
    from ponderv2_amd.spconv import pytorch as spconv

    torch.manual_seed(5)
    coords = random_voxels(14, batch=2, n_per_batch=2000)
    n = len(coords)
    unit = _Unit("inverse", 64, 96).to(device).train()
    feats = torch.randn(n, 64, device=device)
    gout = torch.randn(n, 96, device=device)

    def run(fused):
        monkeypatch.setattr(K, "USE_CONVBN", fused)
        for p in unit.parameters():
            p.grad = None
        f = feats.clone().requires_grad_(True)
        x = spconv.SparseConvTensor(f, torch.from_numpy(coords).to(device), [68, 66, 58], 2)
        out = unit(x).features
        out.backward(gout)
        return out.detach(), f.grad, unit.conv.weight.grad.clone(), unit.pre.weight.grad.clone()

    fused, modular = run(True), run(False)
    for a, b in zip(fused, modular):
        assert _rel(a, b) < 3e-5


def test_tied_conv_weights_with_the_side_stream(device):
    """One conv module applied twice in a graph: the engine sums its two weight gradients on the
    main stream, so only the first may be in flight on the side stream (sidestream.safe_leaf)."""
    from ponderv2_amd import sidestream
    from ponderv2_amd.spconv import pytorch as spconv

    torch.manual_seed(3)
    coords = random_voxels(2, batch=1, n_per_batch=3000)
    n = len(coords)
    conv = spconv.SubMConv3d(64, 64, 3, padding=1, bias=False, indice_key="s").to(device)
    bn = nn.BatchNorm1d(64).to(device).train()
    feats = torch.randn(n, 64, device=device)

    def run(enabled):
        was = sidestream.ENABLED
        sidestream.ENABLED = enabled
        try:
            conv.weight.grad = None
            x = spconv.SparseConvTensor(feats, torch.from_numpy(coords).to(device), [68, 66, 58], 1)
            y = conv.forward_bn(conv.forward_bn(x, bn, relu=True), bn, relu=True).features
            y.square().sum().backward()
            torch.cuda.synchronize()
            return conv.weight.grad.clone()
        finally:
            sidestream.ENABLED = was

    on, off = run(True), run(False)
    assert torch.equal(on, off)   # deterministic kernels: the stream placement must not change a bit


# ------------------------------------------------------------------ the U-Net as one native call
def test_native_unet_equals_the_modular_walk(device, monkeypatch):
    """SpUNet-v1m1 at full width / depth: the natively executed plan (ponderv2_amd/spunet_native.py,
    csrc/spunet_exec.hip) against the module-by-module walk with the same kernels.  Forward: the
    same kernels in the same order => identical bits.  Gradients: equal up to the order in which an
    activation's gradients are added (inside the grad-input reduce here, by autograd there), which
    the ~60 BatchNorm layers amplify - compared by global relative error.  Also: bitwise repeatable,
    and bitwise independent of the side stream."""
    from golden_cases import FULL_BACKBONE
    from ponderv2_amd import sidestream, spunet_native
    from ponderv2_amd.ponder.models import build_model

    torch.manual_seed(0)
    model = build_model(dict(FULL_BACKBONE)).to(device).train()
    coords = random_voxels(11, batch=2, n_per_batch=6000)
    counts = np.bincount(coords[:, 0], minlength=2)
    data = dict(grid_coord=torch.from_numpy(coords[:, 1:]).to(device),
                feat=torch.randn(len(coords), 6, device=device),
                offset=torch.from_numpy(np.cumsum(counts)).to(device))
    probe = None

    def step(native, side=True):
        nonlocal probe
        monkeypatch.setattr(spunet_native, "ENABLED", native)
        monkeypatch.setattr(sidestream, "ENABLED", side)
        before = spunet_native.CALLS
        model.zero_grad(set_to_none=True)
        for m in model.modules():
            if isinstance(m, nn.BatchNorm1d):
                m.reset_running_stats()
        out = model(dict(data))
        assert (spunet_native.CALLS > before) == native
        if probe is None:
            probe = torch.randn_like(out)
        (out * probe).sum().backward()
        torch.cuda.synchronize()
        stats = {n: b.clone() for n, b in model.named_buffers() if "running" in n}
        return out.detach().clone(), {n: p.grad.clone() for n, p in model.named_parameters()}, stats

    nat_out, nat_g, nat_s = step(True)
    again_out, again_g, _ = step(True)
    off_out, off_g, _ = step(True, side=False)
    mod_out, mod_g, mod_s = step(False)
    assert torch.equal(nat_out, again_out) and torch.equal(nat_out, off_out)
    for n in nat_g:
        assert torch.equal(nat_g[n], again_g[n]) and torch.equal(nat_g[n], off_g[n]), n
    assert torch.equal(nat_out, mod_out)                      # forward: the same kernels, the same bits
    for n in nat_s:
        assert torch.equal(nat_s[n], mod_s[n]), n             # running statistics
    assert nat_g.keys() == mod_g.keys()
    num = sum(float((nat_g[n].double() - mod_g[n].double()).square().sum()) for n in nat_g)
    den = sum(float(mod_g[n].double().square().sum()) for n in nat_g)
    worst = max(_rel(nat_g[n], mod_g[n]) for n in nat_g if float(mod_g[n].abs().max()) > 1e-6)
    print("native vs modular gradients: global", (num / den) ** 0.5, "worst tensor", worst)
    assert (num / den) ** 0.5 < 1e-3 and worst < 5e-2


def test_native_unet_returns_the_gradient_of_its_input_features(device, monkeypatch):
    """Block masking writes a LEARNABLE token into the input features (ponder_outdoor_base.py:93-137 /
    masking.py), so the backbone's input carries a gradient: the native executor's stem computes it
    (the gather table walked with mirrored offsets and the transposed weight) - equal to the modular
    walk's, and the 4-channel lidar input (zero-padded to 8 for the kernels) gets exactly its own 4
    columns back."""
    from golden_cases import FULL_BACKBONE
    from ponderv2_amd import spunet_native
    from ponderv2_amd.ponder.models import build_model

    torch.manual_seed(0)
    model = build_model(dict(FULL_BACKBONE, in_channels=4)).to(device).train()
    coords = random_voxels(12, batch=2, n_per_batch=5000)
    counts = np.bincount(coords[:, 0], minlength=2)
    feat0 = torch.randn(len(coords), 4, device=device)
    token = torch.zeros(1, 4, device=device)
    masked = torch.rand(len(coords), device=device) < 0.6
    probe = None
    res = {}
    for native in (True, False):
        monkeypatch.setattr(spunet_native, "ENABLED", native)
        before = spunet_native.CALLS
        model.zero_grad(set_to_none=True)
        for m in model.modules():
            if isinstance(m, nn.BatchNorm1d):
                m.reset_running_stats()
        tok = token.clone().requires_grad_(True)
        feat = torch.where(masked[:, None], tok, feat0)
        feat.retain_grad()
        out = model(dict(grid_coord=torch.from_numpy(coords[:, 1:]).to(device), feat=feat,
                         offset=torch.from_numpy(np.cumsum(counts)).to(device)))
        assert (spunet_native.CALLS > before) == native
        if probe is None:
            probe = torch.randn_like(out)
        (out * probe).sum().backward()
        res[native] = (out.detach().clone(), feat.grad.clone(), tok.grad.clone(),
                       model.conv_input[0].weight.grad.clone())
    a, b = res[True], res[False]
    assert torch.equal(a[0], b[0])
    assert a[1].shape == feat0.shape
    for i, name in ((1, "d features"), (2, "d token"), (3, "d stem weight")):
        assert _rel(a[i], b[i]) < 2e-3, (name, _rel(a[i], b[i]))
