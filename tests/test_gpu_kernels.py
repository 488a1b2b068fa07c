"""GPU parity tests of the C-ABI kernels against oracle/ (run with -m gpu on an MI355X)."""
import numpy as np
import pytest
import torch

from helpers import away_from_kinks, random_voxels

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------ rulebooks (bit exact)
@pytest.mark.parametrize("ksize", [3, 5])
@pytest.mark.parametrize("seed,batch,n", [(0, 2, 1500), (1, 1, 5000), (2, 3, 37)])
def test_subm_rulebook_bit_exact(device, ksize, seed, batch, n):
    from oracle import rulebook as orb
    from ponderv2_amd import kernels as K

    coords = random_voxels(seed, batch=batch, n_per_batch=n)
    rb = K.build_subm_rulebook(torch.from_numpy(coords).to(device), ksize)
    pin, pout, ks = orb.subm_rulebook(coords, ksize)
    assert np.array_equal(rb.kstart_host, ks)
    assert np.array_equal(_np(rb.pair_in), pin)
    assert np.array_equal(_np(rb.pair_out), pout)


def test_subm_rulebook_edge_cases(device):
    from oracle import rulebook as orb
    from ponderv2_amd import kernels as K

    for coords in (np.zeros((0, 4), np.int32), np.array([[0, 0, 0, 0]], np.int32),
                   np.array([[0, 0, 0, 0], [0, 1, 0, 0], [1, 0, 0, 0]], np.int32)):
        rb = K.build_subm_rulebook(torch.from_numpy(coords).to(device), 3)
        pin, pout, ks = orb.subm_rulebook(coords, 3)
        assert np.array_equal(rb.kstart_host, ks)
        assert np.array_equal(_np(rb.pair_in), pin) and np.array_equal(_np(rb.pair_out), pout)


@pytest.mark.parametrize("seed,batch,n", [(0, 2, 1500), (3, 1, 6000), (4, 2, 5)])
def test_downsample_rulebook_bit_exact(device, seed, batch, n):
    from oracle import rulebook as orb
    from ponderv2_amd import kernels as K

    coords = random_voxels(seed, batch=batch, n_per_batch=n)
    shape = [(s - 2) // 2 + 1 for s in (40 + 96, 36 + 96, 20 + 96)]
    rb, oc = K.build_downsample_rulebook(torch.from_numpy(coords).to(device), 2, shape)
    ooc, pin, pout, ks = orb.downsample_rulebook(coords, 2, shape)
    assert np.array_equal(_np(oc), ooc)
    assert np.array_equal(rb.kstart_host, ks)
    assert np.array_equal(_np(rb.pair_in), pin)
    assert np.array_equal(_np(rb.pair_out), pout)
    # chained levels: the next level is built from device-produced coordinates
    rb2, oc2 = K.build_downsample_rulebook(oc, 2, [(s - 2) // 2 + 1 for s in shape])
    ooc2, pin2, pout2, ks2 = orb.downsample_rulebook(ooc, 2, [(s - 2) // 2 + 1 for s in shape])
    assert np.array_equal(_np(oc2), ooc2) and np.array_equal(_np(rb2.pair_in), pin2)


def test_downsample_drops_out_of_shape(device):
    from oracle import rulebook as orb
    from ponderv2_amd import kernels as K

    coords = np.array([[0, 0, 0, 0], [0, 4, 0, 0], [0, 1, 1, 1], [0, 3, 3, 3]], np.int32)
    shape = [2, 2, 2]  # spatial 5 -> (5-2)//2+1 = 2 : the voxel at x=4 has no output
    rb, oc = K.build_downsample_rulebook(torch.from_numpy(coords).to(device), 2, shape)
    ooc, pin, pout, ks = orb.downsample_rulebook(coords, 2, shape)
    assert np.array_equal(_np(oc), ooc) and rb.n_pairs == 3
    assert np.array_equal(_np(rb.pair_in), pin) and np.array_equal(_np(rb.pair_out), pout)


def test_rulebooks_bit_exact_at_the_bench_size(device):
    """BASELINE configs[1] geometry (2 synthetic ScanNet-shaped scenes, ~46.8 k voxels in the
    dataset's hash order): the k5 stem and k3 rulebooks of level 0 and the whole chain of four
    strided levels with their k3 rulebooks - voxel coordinates and (k, in, out) pair lists equal
    the numpy oracle's, bit for bit."""
    from oracle import rulebook as orb
    from ponderv2_amd import kernels as K
    from ponderv2_amd.ponder.datasets import collate_fn, make_scene

    b = collate_fn([make_scene(i, num_views=1, image_hw=(12, 16)) for i in range(2)])
    batch = torch.repeat_interleave(torch.arange(2), torch.diff(b["offset"], prepend=torch.zeros(1, dtype=torch.long)))
    coords = torch.cat([batch[:, None], b["grid_coord"]], 1).int()
    assert len(coords) > 45000
    c_np, c_dev = coords.numpy(), coords.to(device)
    shape = [int(v) + 96 for v in b["grid_coord"].max(0).values]
    for ksize in (5, 3):
        rb = K.build_subm_rulebook(c_dev, ksize)
        pin, pout, ks = orb.subm_rulebook(c_np, ksize)
        assert np.array_equal(rb.kstart_host, ks) and np.array_equal(_np(rb.pair_in), pin)
        assert np.array_equal(_np(rb.pair_out), pout)
    for level in range(4):
        shape = [(s - 2) // 2 + 1 for s in shape]
        rb, oc = K.build_downsample_rulebook(c_dev, 2, shape)
        ooc, pin, pout, ks = orb.downsample_rulebook(c_np, 2, shape)
        assert np.array_equal(_np(oc), ooc) and np.array_equal(rb.kstart_host, ks)
        assert np.array_equal(_np(rb.pair_in), pin) and np.array_equal(_np(rb.pair_out), pout)
        c_np, c_dev = ooc, oc
        rb3 = K.build_subm_rulebook(c_dev, 3)
        pin, pout, ks = orb.subm_rulebook(c_np, 3)
        assert np.array_equal(rb3.kstart_host, ks) and np.array_equal(_np(rb3.pair_in), pin)
        assert np.array_equal(_np(rb3.pair_out), pout)


# ------------------------------------------------------------------ conv arithmetic
def _oracle_conv(feats, w, pin, pout, ks, n_out):
    from oracle.sparse_ops import sparse_conv

    return sparse_conv(feats, w, torch.from_numpy(pin.astype(np.int64)),
                       torch.from_numpy(pout.astype(np.int64)), ks, n_out)


@pytest.mark.parametrize("mode", ["default", "atomics"])
@pytest.mark.parametrize("c_in,c_out,ksize", [(6, 32, 5), (32, 32, 3), (32, 64, 3), (96, 96, 3),
                                               (128, 96, 1), (384, 256, 3), (256, 256, 3), (64, 7, 3)])
def test_spconv_forward_backward_vs_oracle(device, c_in, c_out, ksize, mode, monkeypatch):
    """default: the product-row path + deterministic weight gradient where the shape allows (every
    case but the 6-channel stem and the 7-channel output); atomics: the scatter-add kernels."""
    from oracle import rulebook as orb
    from ponderv2_amd import kernels as K

    if mode == "atomics":
        monkeypatch.setattr(K, "USE_PR", False)
        monkeypatch.setattr(K, "USE_OS", False)
        monkeypatch.setattr(K, "USE_WGRAD_DET", False)
    # odd c_out cases also exercise the optional centre-offset store pass
    monkeypatch.setattr(K, "USE_CENTER_STORE", bool(c_out % 2) or c_out == 96)

    torch.manual_seed(c_in * 1000 + c_out)
    coords = random_voxels(5, batch=2, n_per_batch=700)
    n = len(coords)
    feats = torch.randn(n, c_in)
    w = torch.randn(c_out, ksize ** 3, c_in) * 0.1
    gout = torch.randn(n, c_out)
    pin, pout, ks = orb.subm_rulebook(coords, ksize)

    f_ref = feats.double().requires_grad_(True)
    w_ref = w.double().requires_grad_(True)
    ref = _oracle_conv(f_ref, w_ref, pin, pout, ks, n)
    ref.backward(gout.double())

    rb = K.build_subm_rulebook(torch.from_numpy(coords).to(device), ksize)
    f_dev = feats.to(device).requires_grad_(True)
    w_dev = w.to(device).requires_grad_(True)
    out = K.SparseConvFunction.apply(f_dev, w_dev, rb)
    out.backward(gout.to(device))

    def rel(a, b):
        return (a.double().cpu() - b).abs().max().item() / (b.abs().max().item() + 1e-12)

    assert rel(out, ref.detach()) < 1e-5
    assert rel(f_dev.grad, f_ref.grad) < 1e-5
    assert rel(w_dev.grad, w_ref.grad) < 1e-5


def test_spconv_down_and_inverse_vs_oracle(device):
    from oracle import rulebook as orb
    from ponderv2_amd import kernels as K

    torch.manual_seed(7)
    coords = random_voxels(6, batch=2, n_per_batch=2500)
    n = len(coords)
    shape = [68, 66, 58]
    ooc, pin, pout, ks = orb.downsample_rulebook(coords, 2, shape)
    m = len(ooc)
    c_in, c_out = 32, 64
    feats, w = torch.randn(n, c_in), torch.randn(c_out, 8, c_in) * 0.1
    w_inv = torch.randn(c_in, 8, c_out) * 0.1
    rb, oc = K.build_downsample_rulebook(torch.from_numpy(coords).to(device), 2, shape)
    down = K.spconv_forward(feats.to(device), w.to(device), rb)
    ref_down = _oracle_conv(feats.double(), w.double(), pin, pout, ks, m)
    assert (down.double().cpu() - ref_down).abs().max() < 1e-4 * ref_down.abs().max()
    up = K.spconv_forward(down, w_inv.to(device), rb.transposed())
    ref_up = _oracle_conv(ref_down, w_inv.double(), pout, pin, ks, n)
    assert (up.double().cpu() - ref_up).abs().max() < 1e-4 * ref_up.abs().max()


@pytest.mark.parametrize("tile", [512, 2048])
@pytest.mark.parametrize("c_in,c_out", [(64, 32), (128, 256), (192, 128), (32, 96)])
def test_spconv_wgrad_chunk_sizes(device, tile, c_in, c_out):
    """Both chunk lengths of the LDS-staged weight-gradient kernel, on enough pairs that chunks
    are full, partly filled and (for some offsets) empty."""
    from oracle import rulebook as orb
    from ponderv2_amd import kernels as K

    torch.manual_seed(1)
    coords = random_voxels(8, batch=2, n_per_batch=9000, extent=(64, 60, 24))
    n = len(coords)
    feats, gout = torch.randn(n, c_in), torch.randn(n, c_out)
    pin, pout, ks = orb.subm_rulebook(coords, 3)
    w = torch.zeros(c_out, 27, c_in, dtype=torch.double, requires_grad=True)
    _oracle_conv(feats.double(), w, pin, pout, ks, n).backward(gout.double())
    rb = K.build_subm_rulebook(torch.from_numpy(coords).to(device), 3)
    dw = K.spconv_backward_weight(feats.to(device), gout.to(device), rb, c_out, tile=tile)
    err = (dw.double().cpu() - w.grad).abs().max().item() / w.grad.abs().max().item()
    assert err < 1e-5, err


# ------------------------------------------------------------------ MLP-head GEMMs
@pytest.mark.parametrize("k,n", [(64, 128), (128, 128), (128, 65), (134, 128), (131, 128), (128, 3),
                                 (3, 128), (128, 512), (32, 16), (16, 16), (16, 17), (35, 16)])
def test_mfma_linear_matches_torch_to_second_order(device, k, n):
    """ponderv2_amd.linear.linear == F.linear (fp64 reference) for value, first-order gradients and
    a second-order term (gradient of a function of d(out)/d(x)), incl. the zero-padded odd sizes."""
    import torch.nn.functional as F

    from ponderv2_amd.linear import linear

    torch.manual_seed(k * 1000 + n)
    m = 4500
    x, w, b = torch.randn(m, k), torch.randn(n, k) * 0.2, torch.randn(n)
    probe = torch.randn(m, n)

    def run(fn, x_, w_, b_, probe_):
        x_, w_, b_ = (t.requires_grad_(True) for t in (x_, w_, b_))
        y = torch.tanh(fn(x_, w_, b_))
        (gx,) = torch.autograd.grad((y * probe_).sum(), x_, create_graph=True)
        loss = (gx ** 2).mean() + (y ** 2).mean()
        return (y.detach(), gx.detach()) + torch.autograd.grad(loss, [x_, w_, b_])

    ref = run(F.linear, x.double(), w.double(), b.double(), probe.double())
    got = run(linear, x.to(device), w.to(device), b.to(device), probe.to(device))
    for name, a, r in zip(("y", "dx", "g_x", "g_w", "g_b"), got, ref):
        err = (a.double().cpu() - r).abs().max().item() / (r.abs().max().item() + 1e-12)
        assert err < 2e-5, (name, err)


@pytest.mark.parametrize("m", [1, 2, 31, 4097, 300001])
@pytest.mark.parametrize("i,j", [(16, 16), (32, 16), (20, 32), (4, 8), (32, 32)])
def test_skinny_gemm_tn(device, m, i, j):
    """A^T B for narrow operands (the streaming-reduction kernel): odd row counts, one-row inputs,
    runs that end inside an unrolled group."""
    from ponderv2_amd.linear import reduce_gemm_tn

    torch.manual_seed(m + i + j)
    a, b = torch.randn(m, i), torch.randn(m, j)
    ref = a.double().t() @ b.double()
    got = reduce_gemm_tn(a.to(device), b.to(device)).double().cpu()
    scale = (a.double().abs().t() @ b.double().abs()).max().item()
    assert (got - ref).abs().max().item() < 2e-6 * scale + 1e-12


# ------------------------------------------------------------------ fused BatchNorm (+add+ReLU)
@pytest.mark.parametrize("n,c", [(2, 16), (37, 48), (5000, 32), (46842, 96), (989, 256), (3001, 300)])
@pytest.mark.parametrize("relu,with_res", [(False, False), (True, False), (True, True)])
def test_fused_bn_matches_torch(device, n, c, relu, with_res):
    import torch.nn as nn

    from ponderv2_amd.rownorm import fused_bn

    torch.manual_seed(n + c)
    x = torch.randn(n, c) * 2 + 0.5
    res = torch.randn(n, c) if with_res else None
    gout = torch.randn(n, c)

    def run(dev, dtype):
        bn = nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(dev).to(dtype).train()
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 1.5, c))
            bn.bias.copy_(torch.linspace(-0.2, 0.2, c))
        xi = x.to(dev).to(dtype).requires_grad_(True)
        ri = res.to(dev).to(dtype).requires_grad_(True) if with_res else None
        if dtype == torch.float32:
            y = fused_bn(bn, xi, residual=ri, relu=relu)
        else:  # reference composition
            y = bn(xi)
            if ri is not None:
                y = y + ri
            if relu:
                y = torch.relu(y)
        y.backward(gout.to(dev).to(dtype))
        outs = [y.detach(), xi.grad, bn.weight.grad, bn.bias.grad, bn.running_mean, bn.running_var]
        if with_res:
            outs.append(ri.grad)
        # (the fused path counts batches on the host and writes them back when the state is read)
        return [o.double().cpu() for o in outs], int(bn.state_dict()["num_batches_tracked"])

    ref, _ = run(torch.device("cpu"), torch.float64)
    got, nbt = run(device, torch.float32)
    assert nbt == 1
    for name, a, b in zip(("y", "dx", "dw", "db", "rmean", "rvar", "dres"), got, ref):
        # relative to the tensor's scale, floored: with n = 2 the exact dx is ~eps-sized (BN of two
        # rows cancels almost completely) and a purely relative test would measure rounding noise
        # dx is a cancellation (g - mean g - xhat mean(g xhat)): its error scales with |g| * w * invstd
        floor = float(gout.abs().max()) * 1.5 / (float(x.std()) * 0.2 + 1e-3) if name == "dx" else 0.05
        err = (a - b).abs().max().item() / max(b.abs().max().item(), floor)
        assert err < 2e-5, (name, err)


def test_fused_bn_large_mean_to_std_ratio(device):
    """Columns whose mean is 1e3..1e4 times their spread: the shifted accumulation keeps the batch
    variance (E[(x-s)^2] - (E[x-s])^2 with s = the first row) accurate where E[x^2] - mean^2 in fp32
    would lose every digit."""
    import torch.nn as nn
    from ponderv2_amd.rownorm import fused_bn

    torch.manual_seed(3)
    n, c = 5000, 48
    base = torch.linspace(50.0, 4000.0, c)
    x = base + 0.3 * torch.randn(n, c)
    bn = nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(device).train()
    y = fused_bn(bn, x.to(device))
    xd = x.double()
    ref = (xd - xd.mean(0)) / torch.sqrt(xd.var(0, unbiased=False) + 1e-3)
    # fp32 inputs at |x| ~ 4000 carry ~2.4e-4 absolute noise, i.e. ~1e-3 of the 0.3 spread
    assert (y.double().cpu() - ref).abs().max() < 5e-3
    sd = bn.state_dict()
    assert (sd["running_var"].double().cpu() - (0.99 + 0.01 * xd.var(0))).abs().max() < 1e-4


def test_col_sum(device):
    from ponderv2_amd.rownorm import col_sum

    torch.manual_seed(0)
    for m, n in ((135168, 128), (5000, 65), (70000, 3), (4096, 512), (200000, 1)):
        x = torch.randn(m, n)
        got = col_sum(x.to(device)).double().cpu()
        ref = x.double().sum(0)
        assert (got - ref).abs().max() < 1e-4 * (ref.abs().max() + 1), (m, n)


# ------------------------------------------------------------------ scatter mean
def test_scatter_mean_vs_oracle(device):
    from oracle.scatter import scatter as oscatter
    from ponderv2_amd.torch_scatter import scatter

    torch.manual_seed(0)
    m, c, g = 5000, 96, 700
    src = torch.randn(m, c)
    idx = torch.randint(0, g, (m, 1))
    gout = torch.randn(g, c)
    s_ref = src.double().requires_grad_(True)
    ref = oscatter(s_ref, idx, dim=0, reduce="mean", out=torch.zeros(g, c, dtype=torch.double))
    ref.backward(gout.double())
    s_dev = src.to(device).requires_grad_(True)
    out = scatter(s_dev, idx.to(device), dim=0, reduce="mean", out=torch.zeros(g, c, device=device))
    out.backward(gout.to(device))
    assert (out.double().cpu() - ref.detach()).abs().max() < 1e-5
    assert (s_dev.grad.double().cpu() - s_ref.grad).abs().max() < 1e-5


def test_scatter_into_a_view_like_the_reference_to_dense(device):
    """The reference's call pattern (ponder_indoor_base.py:214): ``fea_grid[i] = scatter(feat, idx,
    dim=0, reduce="mean", out=fea_grid[i])`` - ``out`` is a VIEW of the batch grid, modified in
    place; the gradient of every scene's rows must come back through the base tensor."""
    from oracle.scatter import scatter as oscatter
    from ponderv2_amd.torch_scatter import scatter

    torch.manual_seed(1)
    m, c, g, scenes = 900, 24, 64, 3
    srcs = [torch.randn(m + 7 * i, c) for i in range(scenes)]
    idxs = [torch.randint(0, g, (m + 7 * i, 1)) for i in range(scenes)]
    probe = torch.randn(scenes, g, c)

    def run(fn, dev, dtype):
        leaves = [s.to(dev, dtype).requires_grad_(True) for s in srcs]
        grid = torch.zeros(scenes, g, c, device=dev, dtype=dtype)
        for i in range(scenes):
            grid[i] = fn(leaves[i] * (i + 1.0), idxs[i].to(dev), dim=0, reduce="mean", out=grid[i])
        (grid * probe.to(dev, dtype)).sum().backward()
        return grid.detach(), [l.grad for l in leaves]

    ref_grid, ref_grads = run(oscatter, "cpu", torch.float64)
    got_grid, got_grads = run(scatter, device, torch.float32)
    assert (got_grid.double().cpu() - ref_grid).abs().max() < 1e-5
    for a, b in zip(got_grads, ref_grads):
        assert a is not None and (a.double().cpu() - b).abs().max() < 1e-5


# ------------------------------------------------------------------ trilinear sampler
@pytest.mark.parametrize("padding_mode", ["zeros", "border", "reflection"])
@pytest.mark.parametrize("align_corners", [True, False])
def test_sampler_matches_grid_sample(device, padding_mode, align_corners):
    """The reference's own pin: libs/smooth-sampler/smooth_sampler/modules.py:104-130."""
    from ponderv2_amd.smooth_sampler import SmoothSampler

    torch.manual_seed(3)
    inp = torch.rand(2, 2, 2, 3, 11, device=device, requires_grad=True)
    grid = (torch.rand(2, 2, 1, 5, 3, device=device) * 2 - 1).requires_grad_(True)
    out1 = SmoothSampler.apply(inp, grid, padding_mode, align_corners, False)
    out2 = torch.nn.functional.grid_sample(inp, grid, padding_mode=padding_mode,
                                           align_corners=align_corners)
    assert torch.allclose(out1, out2, atol=1e-6)
    g1 = torch.autograd.grad(out1, [inp, grid], torch.ones_like(out1), create_graph=True)
    g2 = torch.autograd.grad(out2, [inp, grid], torch.ones_like(out2))
    assert torch.allclose(g1[0], g2[0], atol=1e-5)
    assert torch.allclose(g1[1], g2[1], atol=1e-4)


@pytest.mark.parametrize("padding_mode", ["zeros", "border", "reflection"])
@pytest.mark.parametrize("align_corners", [True, False])
@pytest.mark.parametrize("smooth", [True, False])
def test_sampler_gradcheck_fp64(device, padding_mode, align_corners, smooth):
    """modules.py:132-156: gradcheck + gradgradcheck in float64 (eps 1e-4, atol 1e-3, rtol 1e-2)."""
    from ponderv2_amd.smooth_sampler import SmoothSampler

    torch.manual_seed(11)
    inp = torch.rand(2, 2, 2, 3, 11, dtype=torch.double, device=device).requires_grad_(True)
    grid = torch.rand(2, 2, 1, 5, 3, dtype=torch.double) * 2 - 1
    if padding_mode == "zeros":
        grid = grid * 1.2  # exercise out-of-volume corners too
    grid = away_from_kinks(grid, (11, 3, 2), align_corners).to(device).requires_grad_(True)
    fn = lambda a, b: SmoothSampler.apply(a, b, padding_mode, align_corners, smooth)  # noqa: E731
    torch.autograd.gradcheck(fn, [inp, grid], eps=1e-4, atol=1e-3, rtol=1e-2, nondet_tol=1e-9)
    torch.autograd.gradgradcheck(fn, [inp, grid], eps=1e-4, atol=1e-3, rtol=1e-2, nondet_tol=1e-9)


@pytest.mark.parametrize("C,B,channels_last,padding_mode,smooth", [
    (128, 1, True, "zeros", False), (128, 1, False, "zeros", False),   # indoor head, both layouts
    (32, 2, True, "zeros", False),     # outdoor head: 8 points per wave, batched volume
    (96, 1, True, "zeros", False),     # 24 float4 per point on 32 lanes (idle lanes)
    (20, 2, True, "zeros", False),     # 5 float4 on 8 lanes
    (256, 1, True, "zeros", False),    # one full wave per point
    (512, 1, True, "zeros", False),    # two channel panels per lane
    (64, 2, True, "border", True), (64, 1, True, "reflection", True),
    (30, 1, True, "zeros", False),     # C % 4 != 0: generic kernels on a channels-last volume
])
def test_sampler_second_order_vs_oracle_f32(device, C, B, channels_last, padding_mode, smooth):
    """fp32 against the float64 oracle, up to a loss that uses d(out)/d(grid) (backward-of-backward,
    like the eikonal term).  Channels-last volumes with C % 4 == 0 run the vectorised lane-group
    kernels (csrc/trilinear.hip), the rest the generic one-wave-per-point kernels; 407 points per
    volume leave the last wave partially filled."""
    from oracle.sampler import SmoothSampler as OSampler
    from ponderv2_amd.smooth_sampler import SmoothSampler

    torch.manual_seed(5 + C)
    D, H, W, R, S = 6, 9, 10, 37, 11
    vol = torch.randn(B, C, D, H, W)
    grid = away_from_kinks(torch.rand(B, 1, R, S, 3) * 2.3 - 1.15, (W, H, D), True)
    proj = torch.randn(C) / C ** 0.5

    def run(sampler, vol_t, grid_t):
        vol_t = vol_t.requires_grad_(True)
        grid_t = grid_t.requires_grad_(True)
        out = sampler.apply(vol_t, grid_t, padding_mode, True, smooth)
        feat = out.squeeze(2).permute(0, 2, 3, 1)  # (B,R,S,C)
        sdf = torch.tanh(feat @ proj.to(feat))
        (gp,) = torch.autograd.grad(sdf.sum(), grid_t, create_graph=True)
        loss = ((gp.norm(dim=-1) - 1) ** 2).mean() + (feat ** 2).mean() + sdf.mean()
        gv, gg = torch.autograd.grad(loss, [vol_t, grid_t])
        return out.detach(), gp.detach(), gv, gg

    ref = run(OSampler, vol.double(), grid.double())
    v_dev = vol.to(device)
    if channels_last:
        v_dev = v_dev.contiguous(memory_format=torch.channels_last_3d)
    got = run(SmoothSampler, v_dev, grid.to(device))
    for name, a, b in zip(("out", "dgrid", "gvol", "ggrid"), got, ref):
        err = (a.double().cpu() - b).abs().max().item() / (b.abs().max().item() + 1e-12)
        assert err < 2e-4, (name, err)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("channels_last", [True, False])
def test_sampler_half_dispatch_vs_oracle(device, dtype, channels_last):
    """The sampler on 16-bit tensors - the half branch of the reference's dispatch
    (smooth_sampler_kernel.cu:630,670,726 AT_DISPATCH_FLOATING_TYPES_AND_HALF), which its shipped
    ``enable_amp=True`` configs reach.  Each of the three entry points is compared with the float64
    oracle evaluated on the SAME 16-bit-rounded operands; what is left is the rounding of the 16-bit
    results (fp32 arithmetic, fp32 volume-gradient accumulation): a few units of the type's epsilon
    relative to the largest entry."""
    from oracle.sampler import SmoothSampler as OSampler
    from ponderv2_amd.smooth_sampler import SmoothSampler

    torch.manual_seed(23)
    B, C, D, H, W, R, S = 2, 24, 5, 7, 9, 29, 7
    vol = torch.randn(B, C, D, H, W).to(dtype)
    grid = away_from_kinks(torch.rand(B, 1, R, S, 3) * 2.3 - 1.15, (W, H, D), True).to(dtype)
    go = torch.randn(B, C, 1, R, S).to(dtype)
    hV = torch.randn(B, C, D, H, W).to(dtype)
    hG = torch.randn(B, 1, R, S, 3).to(dtype)

    def run(sampler, cast):
        v, g, o = (cast(t).requires_grad_(True) for t in (vol, grid, go))
        out = sampler.apply(v, g, "zeros", True, False)
        g_in, g_grid = torch.autograd.grad(out, [v, g], o, create_graph=True)
        second = torch.autograd.grad([g_in, g_grid], [v, g, o], [cast(hV), cast(hG)])
        return (out.detach(), g_in.detach(), g_grid.detach()) + tuple(second)

    def to_dev(t):
        t = t.to(device)
        if channels_last and t.dim() == 5 and t.shape[1] == C and t.shape[2] == D:
            t = t.contiguous(memory_format=torch.channels_last_3d)
        return t

    ref = run(OSampler, lambda t: t.double())
    got = run(SmoothSampler, to_dev)
    eps = torch.finfo(dtype).eps
    names = ("out", "grad_input", "grad_grid", "grad_input2", "grad_grid2", "grad_grad_out")
    for name, a, b in zip(names, got, ref):
        assert a.dtype == dtype, (name, a.dtype)
        err = (a.double().cpu() - b).abs().max().item() / (b.abs().max().item() + 1e-12)
        assert err < 4 * eps, (name, err, eps)


def test_sampler_rejects_mixed_dtypes(device):
    """Like the reference's extension (one scalar type per call), no silent widening."""
    from ponderv2_amd.smooth_sampler import SmoothSampler

    vol = torch.randn(1, 4, 3, 3, 3, device=device, dtype=torch.float16)
    grid = torch.zeros(1, 1, 1, 2, 3, device=device)
    with pytest.raises(TypeError, match="expected every tensor"):
        SmoothSampler.apply(vol, grid, "zeros", True, False)


# ------------------------------------------------------------------ sparse first dense layer
@pytest.mark.parametrize("with_bn", [True, False])
def test_sparse_first_layer_equals_dense_layer_gpu(device, with_bn):
    """Same check as the CPU one, on the HIP sparse-conv kernels in fp32 (reference: float64)."""
    import sparse_input_cases as sic

    errs = sic.run(device, torch.float32, with_bn)
    assert max(errs.values()) < 2e-5, errs


@pytest.mark.parametrize("c_in,c_out,dims,n_vox", [(8, 4, (5, 12, 9), 220), (96, 32, (4, 18, 37), 900),
                                                   (16, 8, (1, 2, 33), 40), (12, 16, (3, 1, 2), 14)])
def test_cells_level_node_equals_dense_layer(device, c_in, c_out, dims, n_vox):
    """The first projection level as one node on the HIP kernels (cells_level.py): output, all
    gradients and the running statistics against the dense float64 layer; grids with axes of size
    one and two included (every border class)."""
    import sparse_input_cases as sic
    from ponderv2_amd import cells_level

    calls = []
    real = cells_level._CellsLevel.apply
    cells_level._CellsLevel.apply = lambda *a: (calls.append(1), real(*a))[1]
    try:
        errs = sic.run(device, torch.float32, True, seed=3, c_in=c_in, c_out=c_out, dims=dims, n_vox=n_vox)
    finally:
        cells_level._CellsLevel.apply = real
    assert calls, "the composite ran instead of the node"
    assert max(errs.values()) < 3e-5, errs


def test_cells_level_premasked_gradient_and_tap_table(device):
    """A consumer that claims the ReLU mask hands the node a premasked gradient: same gradients as
    when the node masks itself.  And the tap table kernel equals the torch composite."""
    import os

    import sparse_input_cases as sic
    from ponderv2_amd import cells_level
    from ponderv2_amd.ponder.models.ponder import sparse_input as si

    lin, feat, B, dims = sic.make_case(5, dims=(4, 10, 7), c_in=16, n_vox=300)
    torch.manual_seed(0)
    bn = torch.nn.BatchNorm3d(16, eps=1e-3).to(device).train()
    conv = torch.nn.Conv3d(16, 8, 3, padding=1, bias=False).to(device)
    probe = torch.randn(B, 8, *dims, device=device)

    def once(premask):
        f = feat.float().to(device).requires_grad_(True)
        cells = si.cells_from_voxels(f, lin.to(device), B, dims)
        vol = si.bn_conv_relu_on_cells(bn, conv, cells)
        g = probe
        if premask:
            assert cells_level.claim_premasked(vol)
            g = probe * (vol.detach() > 0)
        for p in (bn.weight, bn.bias, conv.weight):
            p.grad = None
        vol.backward(g)
        torch.cuda.synchronize()
        return [f.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone(), conv.weight.grad.clone()]

    for a, b in zip(once(False), once(True)):   # (equal up to the atomics of scatter-mean / the conv)
        assert (a - b).abs().max() <= 1e-4 * b.abs().max() + 1e-6

    cells = si.cells_from_voxels(feat.float().to(device), lin.to(device), B, dims)
    fast = cells_level.tap_table(cells)
    cells_level.ENABLED = False
    try:
        slow = si._tap_table(cells)
    finally:
        cells_level.ENABLED = True
    assert torch.equal(fast, slow)


# ------------------------------------------------------------------ compositing (raymarch.hip)
@pytest.mark.parametrize("rays,samples", [(1, 1), (5, 64), (1030, 132), (300, 96), (17, 256)])
def test_raymarch_weights_vs_oracle(device, rays, samples):
    from oracle import raymarch as orm
    from ponderv2_amd.raymarch import composite_weights

    torch.manual_seed(rays + samples)
    alphas = torch.rand(rays, samples, 1) * 0.95
    if rays > 4:
        alphas[0, samples // 2:] = 1.0
        alphas[1] = 0.0
    gw = torch.randn(rays, samples, 1)
    w_ref, t_ref = orm.weights_from_alphas(alphas.double())
    ga_ref = orm.grad_alpha_closed_form(alphas.double(), gw.double())
    a_dev = alphas.to(device).requires_grad_(True)
    w, t = composite_weights(a_dev)
    (ga,) = torch.autograd.grad(w, a_dev, gw.to(device))
    for name, got, ref in (("weights", w, w_ref), ("transmittance", t, t_ref), ("galpha", ga, ga_ref)):
        err = (got.double().cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)
        assert err < 1e-5, (name, err)


@pytest.mark.parametrize("features", [1, 3, 31, 32, 128, 131, 512])
def test_raymarch_weighted_sum_vs_oracle(device, features):
    from oracle import raymarch as orm
    from ponderv2_amd.raymarch import weighted_sum

    torch.manual_seed(features)
    rays, samples = 130, 132
    w, x = torch.rand(rays, samples, 1), torch.randn(rays, samples, features)
    gout = torch.randn(rays, features)
    ref = orm.weighted_sum(w.double(), x.double())
    gw_ref, gx_ref = orm.weighted_sum_grads_closed_form(w.double(), x.double(), gout.double())
    w_dev, x_dev = w.to(device).requires_grad_(True), x.to(device).requires_grad_(True)
    out = weighted_sum(w_dev, x_dev)
    gw, gx = torch.autograd.grad(out, [w_dev, x_dev], gout.to(device))
    for name, got, r in (("out", out, ref), ("gw", gw, gw_ref), ("gx", gx, gx_ref)):
        err = (got.double().cpu() - r).abs().max().item() / (r.abs().max().item() + 1e-12)
        assert err < 1e-5, (name, err)


@pytest.mark.parametrize("c_in,c_out", [(32, 32), (96, 128), (256, 96), (64, 12)])
def test_grad_input_reads_the_forward_weight_in_place(device, c_in, c_out, monkeypatch):
    """spconv_grad_input on the scatter path (pv2_spconv_forward_wt: the forward weight staged
    reduction-major through LDS) equals the conv of grad_out with an explicitly transposed copy."""
    from ponderv2_amd import kernels as K

    monkeypatch.setattr(K, "USE_OS", False)
    monkeypatch.setattr(K, "USE_PR", False)   # this test is about the scatter-add kernel
    torch.manual_seed(c_in * 7 + c_out)
    coords = random_voxels(13, batch=2, n_per_batch=1200)
    n = len(coords)
    rb = K.build_subm_rulebook(torch.from_numpy(coords).to(device), 3)
    w = torch.randn(c_out, 27, c_in, device=device) * 0.1
    g = torch.randn(n, c_out, device=device)
    got = K.spconv_grad_input(g, w, rb)
    ref = K.spconv_forward(g, w.permute(2, 1, 0).contiguous(), rb.transposed())
    assert got.shape == (n, c_in)
    assert (got - ref).abs().max() <= 2e-5 * ref.abs().max()


# ------------------------------------------------------------------ output-stationary conv
@pytest.mark.parametrize("c_in,c_out,ksize", [(6, 32, 5), (32, 64, 3), (128, 128, 3), (256, 96, 3)])
def test_output_stationary_conv_is_bitwise_reproducible_and_equals_scatter_kernel(
        device, c_in, c_out, ksize, monkeypatch):
    """Forward and grad-input on the output-stationary kernel: identical bits on every call, and
    the same numbers (fp32 re-association only) as the pair-major scatter-add kernels."""
    from ponderv2_amd import kernels as K

    torch.manual_seed(c_in + c_out)
    coords = random_voxels(8, batch=2, n_per_batch=900)
    n = len(coords)
    monkeypatch.setattr(K, "USE_OS", True)
    monkeypatch.setattr(K, "USE_PR", False)   # this test is about the gather-table kernel
    rb = K.build_subm_rulebook(torch.from_numpy(coords).to(device), ksize)
    assert rb.nbr is not None
    x = torch.randn(n, c_in, device=device)
    w = torch.randn(c_out, ksize ** 3, c_in, device=device) * 0.1
    g = torch.randn(n, c_out, device=device)
    w_t = w.permute(2, 1, 0).contiguous()
    runs = [(K.spconv_forward(x, w, rb), K.spconv_forward(g, w_t, rb.transposed())) for _ in range(3)]
    for y, dx in runs[1:]:
        assert torch.equal(y, runs[0][0]) and torch.equal(dx, runs[0][1])
    monkeypatch.setattr(K, "USE_OS", False)
    y_ref, dx_ref = K.spconv_forward(x, w, rb), K.spconv_forward(g, w_t, rb.transposed())
    assert (runs[0][0] - y_ref).abs().max() <= 2e-5 * y_ref.abs().max()
    assert (runs[0][1] - dx_ref).abs().max() <= 2e-5 * dx_ref.abs().max()


def test_output_stationary_strided_and_inverse_conv(device, monkeypatch):
    """Strided conv (children gathered per output voxel), its grad-input / the inverse conv (one
    parent per input voxel) and a fused bias, against the scatter-add kernels; bitwise repeatable."""
    from ponderv2_amd import kernels as K

    monkeypatch.setattr(K, "USE_PR", False)   # the gather-table kernel against the scatter-add one
    torch.manual_seed(11)
    coords = random_voxels(9, batch=2, n_per_batch=2500)
    shape = [68, 66, 58]
    rb, oc = K.build_downsample_rulebook(torch.from_numpy(coords).to(device), 2, shape)
    n, m = len(coords), rb.n_out
    x = torch.randn(n, 32, device=device)
    w = torch.randn(64, 8, 32, device=device) * 0.1
    w_inv = torch.randn(48, 8, 64, device=device) * 0.1
    bias = torch.randn(64, device=device)
    down = K.spconv_forward(x, w, rb, bias=bias)
    up = K.spconv_forward(down, w_inv, rb.transposed())
    assert down.shape == (m, 64) and up.shape == (n, 48)
    assert torch.equal(down, K.spconv_forward(x, w, rb, bias=bias))
    assert torch.equal(up, K.spconv_forward(down, w_inv, rb.transposed()))
    monkeypatch.setattr(K, "USE_OS", False)
    down_ref = K.spconv_forward(x, w, rb) + bias
    up_ref = K.spconv_forward(down_ref, w_inv, rb.transposed())
    assert (down - down_ref).abs().max() <= 2e-5 * down_ref.abs().max()
    assert (up - up_ref).abs().max() <= 2e-5 * up_ref.abs().max()


def test_spunet_forward_is_bitwise_reproducible(device, monkeypatch):
    """The DEFAULT kernel selection (product-row convs, output-stationary stem): forward passes of
    the backbone on the same input give identical bits - no atomics on the forward path."""
    import golden_cases as gc
    from ponderv2_amd import kernels as K

    assert K.USE_PR == "all" and K.USE_OS == "auto"
    from oracle.detweights import fill_deterministic, formula_tensor
    from ponderv2_amd.ponder.models import build_model
    from ponderv2_amd.ponder.utils.config import ConfigDict

    g = np.load(gc.os.path.join(gc.GOLDEN, "spunet_small.npz"))
    coords = g["coords"]
    counts = np.bincount(coords[:, 0])
    model = build_model(ConfigDict(gc.SMALL_BACKBONE))
    fill_deterministic(model)
    model = model.to(device).train()
    feat = formula_tensor("spunet.feat", (len(coords), 6), 1.0).to(device)
    batch = dict(grid_coord=torch.from_numpy(coords[:, 1:].astype(np.int64)).to(device), feat=feat,
                 offset=torch.from_numpy(np.cumsum(counts)).long().to(device))
    outs = [model(dict(batch)).clone() for _ in range(3)]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("seed,batch,n", [(0, 2, 1500), (21, 1, 6000), (22, 3, 50)])
def test_one_read_geometry_prepass_equals_the_lazy_builders(device, seed, batch, n):
    """kernels.prepare_unet_geometry (all ten rulebooks of the U-Net chained on capacity-sized,
    padded arrays, one device->host read) produces exactly the rulebooks, output voxels and gather
    tables of the per-rulebook builders."""
    from ponderv2_amd import kernels as K

    coords = torch.from_numpy(random_voxels(seed, batch=batch, n_per_batch=n)).to(device)
    shape = [40 + 96, 36 + 96, 20 + 96]
    geo = K.prepare_unet_geometry(coords, shape, n_levels=4)
    assert set(geo) == {"stem", "subm0", "subm1", "subm2", "subm3", "subm4",
                        "spconv1", "spconv2", "spconv3", "spconv4"}
    level, lshape = coords, shape
    for l in range(5):
        for key, ks in ([("stem", 5)] if l == 0 else []) + [(f"subm{l}", 3)]:
            ref, got = K.build_subm_rulebook(level, ks), geo[key]["rulebook"]
            assert geo[key]["n"] == len(level) and got.n_pairs == ref.n_pairs
            assert np.array_equal(got.kstart_host, ref.kstart_host)
            assert torch.equal(got.pair_in, ref.pair_in) and torch.equal(got.pair_out, ref.pair_out)
            assert torch.equal(got.kstart, ref.kstart)
            tbl = got.nbr.reshape(got.K, got.nbr_stride)[:, :len(level)]
            assert torch.equal(tbl, ref.nbr.reshape(ref.K, ref.nbr_stride))
        if l == 4:
            break
        lshape = [(s - 2) // 2 + 1 for s in lshape]
        ref, oc = K.build_downsample_rulebook(level, 2, lshape)
        e = geo[f"spconv{l + 1}"]
        got = e["rulebook"]
        assert e["out_shape"] == lshape and torch.equal(e["out_indices"], oc)
        assert (got.n_in, got.n_out, got.n_pairs) == (ref.n_in, ref.n_out, ref.n_pairs)
        assert torch.equal(got.pair_in, ref.pair_in) and torch.equal(got.pair_out, ref.pair_out)
        x = torch.randn(got.n_in, 32, device=device)
        w = torch.randn(48, 8, 32, device=device) * 0.1
        y, y_ref = K.spconv_forward(x, w, got), K.spconv_forward(x, w, ref)
        assert torch.equal(y, y_ref)            # same table, same row order -> same bits
        g = torch.randn(got.n_out, 48, device=device)
        assert torch.equal(K.spconv_grad_input(g, w, got), K.spconv_grad_input(g, w, ref))
        level = oc


def test_prefetched_geometry_equals_the_blocking_build(device, monkeypatch):
    """kernels.prefetch_unet_geometry (side stream, asynchronous copy of the counts) hands the
    backbone the same rulebooks as the blocking build, and a step that uses tensors allocated on
    the side stream computes the same bits - also while the main stream is busy."""
    from golden_cases import FULL_BACKBONE
    from ponderv2_amd import kernels as K
    from ponderv2_amd.ponder.models import build_model

    monkeypatch.setattr(K, "USE_OS", True)  # deterministic forward: bitwise comparison
    torch.manual_seed(0)
    model = build_model(dict(FULL_BACKBONE)).to(device).train()
    coords = random_voxels(31, batch=2, n_per_batch=5000)
    counts = np.bincount(coords[:, 0], minlength=2)
    shape = [int(v) + 96 for v in coords[:, 1:].max(0)]
    data = dict(grid_coord=torch.from_numpy(coords[:, 1:]).to(device),
                feat=torch.randn(len(coords), 6, device=device),
                offset=torch.from_numpy(np.cumsum(counts)).to(device), sparse_shape=shape)
    ref = model(dict(data))
    busy = torch.randn(4096, 4096, device=device)
    for _ in range(3):
        ahead = model.prefetch_geometry(dict(data))
        assert isinstance(ahead["geometry"], K.PendingGeometry)
        for _ in range(4):
            busy = busy @ busy * 1e-3   # keep the main stream busy while the tables are built
        out = model(ahead)
        assert torch.equal(out, ref)
    geo_a = K.prepare_unet_geometry(torch.from_numpy(coords).to(device), shape)
    geo_b = ahead["geometry"].result()
    assert set(geo_a) == set(geo_b)
    for key in geo_a:
        a, b = geo_a[key]["rulebook"], geo_b[key]["rulebook"]
        assert np.array_equal(a.kstart_host, b.kstart_host)
        assert torch.equal(a.pair_in, b.pair_in) and torch.equal(a.pair_out, b.pair_out)


@pytest.mark.parametrize("shape,c_in,c_out,bias", [((2, 16, 32, 32), 32, 128, True),
                                                   ((1, 32, 128, 128), 32, 128, True),   # >= 512 k rows: the long-run reduction
                                                   ((1, 8, 24, 20), 64, 40, False)])
def test_pointwise_conv_equals_the_library_conv(device, shape, c_in, c_out, bias):
    """UNet3D's final 1x1x1 convolution as one tall GEMM over the channels-last rows
    (unet3d._PointwiseConv on pv2_gemm_nt / pv2_gemm_tn / pv2_col_sum) against ``F.conv3d`` in
    float64: output, grad-input, grad-weight, grad-bias."""
    import torch.nn as nn

    from ponderv2_amd.ponder.models.ponder import unet3d as U

    torch.manual_seed(0)
    b, z, y, x = shape
    conv = nn.Conv3d(c_in, c_out, 1, bias=bias).to(device)
    inp = torch.randn(b, c_in, z, y, x, device=device).contiguous(memory_format=torch.channels_last_3d)
    inp.requires_grad_(True)
    assert U.pointwise_conv_supported(conv, inp)
    out = U.library_conv(conv, inp)
    assert out.is_contiguous(memory_format=torch.channels_last_3d) and out.dtype == torch.float32
    g = torch.randn(b, z, y, x, c_out, device=device).permute(0, 4, 1, 2, 3)   # as the ray march hands it back
    out.backward(g)
    torch.cuda.synchronize()
    conv64 = nn.Conv3d(c_in, c_out, 1, bias=bias).to(device).double()
    conv64.load_state_dict({k: v.double() for k, v in conv.state_dict().items()})
    inp64 = inp.detach().double().requires_grad_(True)
    ref = conv64(inp64)
    ref.backward(g.double())

    def rel(a, b_):
        return float((a.double() - b_).abs().max() / b_.abs().max())

    assert rel(out, ref) < 1e-5
    assert rel(inp.grad, inp64.grad) < 1e-5
    assert rel(conv.weight.grad, conv64.weight.grad) < 2e-5     # sums over up to 524 288 rows
    if bias:
        assert rel(conv.bias.grad, conv64.bias.grad) < 2e-5


@pytest.mark.parametrize("shape", [(2, 32, 8, 12, 16), (1, 64, 5, 7, 9), (2, 128, 4, 4, 4)])
def test_channels_last_max_pool_kernels_equal_max_pool3d(device, shape):
    """csrc/dense_pool.hip against F.max_pool3d(x, 2): values and gradients EQUAL, also where a
    window holds several equal maxima (zeros after a ReLU: the first in window order gets the
    gradient on both sides), odd extents floored, layout kept channels-last."""
    import torch.nn.functional as F

    from ponderv2_amd.ponder.models.ponder.unet3d import _MaxPoolCL, channels_last_max_pool3d

    torch.manual_seed(0)
    for relu in (False, True):
        x = torch.randn(*shape, device=device)
        if relu:
            x = x.relu()
        x = x.contiguous(memory_format=torch.channels_last_3d)
        a = x.clone().requires_grad_(True)
        b = x.clone().requires_grad_(True)
        got, ref = channels_last_max_pool3d(a), F.max_pool3d(b, 2)
        assert isinstance(got.grad_fn, _MaxPoolCL._backward_cls)
        assert torch.equal(got, ref) and got.is_contiguous(memory_format=torch.channels_last_3d)
        probe = torch.randn_like(ref)
        (got * probe).sum().backward()
        (ref * probe).sum().backward()
        assert torch.equal(a.grad, b.grad)


@pytest.mark.parametrize("n,batch", [(4, 13), (3, 2), (4, 1), (2, 700)])
def test_small_inverse_matches_float64_inverse(device, n, batch):
    """pv2_small_inverse (the camera / unit-cube transforms of the ray set-up: torch.linalg.inv at
    ponder_indoor_base.py:380-470 of the reference) against numpy's float64 inverse: rigid poses with
    large translations, scaled intrinsics-like matrices and rows that need pivoting."""
    from ponderv2_amd.ponder.models.ponder.ponder_indoor_base import _inv

    rng = np.random.default_rng(n * 100 + batch)
    a = rng.normal(size=(batch, n, n))
    q, _ = np.linalg.qr(a)
    a = q * rng.uniform(0.5, 600.0, size=(batch, 1, n))          # well conditioned, mixed scales
    a[:, :-1, -1] += rng.uniform(-40, 40, size=(batch, n - 1))   # pose-like translation column
    a[0] = a[0][::-1].copy()                                      # a zero-ish leading pivot
    a32 = a.astype(np.float32)
    ref = np.linalg.inv(a32.astype(np.float64))
    x = torch.from_numpy(a32).to(device)
    got = _inv(x)
    assert got.shape == x.shape and got.dtype == torch.float32
    err = np.abs(got.cpu().numpy().astype(np.float64) - ref).max(axis=(1, 2)) / np.abs(ref).max(axis=(1, 2))
    assert err.max() < 2e-6, err.max()
    lib_err = np.abs(torch.linalg.inv(x).cpu().numpy().astype(np.float64) - ref).max(axis=(1, 2)) / np.abs(ref).max(axis=(1, 2))
    assert err.max() <= max(2 * lib_err.max(), 3e-7)            # at least as accurate as the library's fp32 LU
    # batch dimensions are kept: (B, V, n, n) as the pose stack arrives
    assert torch.equal(_inv(x[None]), got[None])


@pytest.mark.parametrize("kind", ["SGD", "AdamW"])
def test_lean_fused_optimizer_steps_equal_torch(device, kind):
    """utils/optimizer.py replaces the per-step Python of torch's fused SGD / AdamW by direct calls of
    the same multi-tensor kernels on cached lists: parameters and optimizer state after several steps
    under a OneCycle schedule EQUAL torch's own fused step bit for bit - including a parameter that
    never receives a gradient and one whose gradient disappears for a step."""
    from ponderv2_amd.ponder.utils.optimizer import build_optimizer

    def make():
        torch.manual_seed(3)
        m = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Linear(16, 4)).to(device)
        m.unused = torch.nn.Parameter(torch.ones(3, device=device))
        return m

    cfg = (dict(type="SGD", lr=0.1, momentum=0.9, weight_decay=1e-4, nesterov=True) if kind == "SGD"
           else dict(type="AdamW", lr=2e-3, weight_decay=0.01))
    ours, ref = make(), make()
    opt = build_optimizer(dict(cfg), ours)
    cls = torch.optim.SGD if kind == "SGD" else torch.optim.AdamW
    ref_opt = cls(ref.parameters(), fused=True, **{k: v for k, v in cfg.items() if k != "type"})
    if kind == "SGD":
        for p in ref.parameters():      # (build_optimizer creates the momentum buffers up front)
            ref_opt.state[p]["momentum_buffer"] = torch.zeros_like(p)
    scheds = [torch.optim.lr_scheduler.OneCycleLR(o, max_lr=cfg["lr"], total_steps=12) for o in (opt, ref_opt)]
    x = torch.randn(5, 8, device=device)
    for it in range(8):
        for m, o, sc in ((ours, opt, scheds[0]), (ref, ref_opt, scheds[1])):
            o.zero_grad(set_to_none=True)
            loss = m(x).square().sum() if it != 4 else m[1](torch.randn(5, 16, device=device, generator=None) * 0 + 1).sum()
            loss.backward()
            o.step()
            sc.step()
    assert opt._pv2_lean_steps >= 5, opt._pv2_lean_steps   # (the rest went through torch's own step)
    for (n, a), (_, b) in zip(ours.named_parameters(), ref.named_parameters()):
        assert torch.equal(a, b), n
    sa, sb = opt.state_dict()["state"], ref_opt.state_dict()["state"]
    assert sa.keys() == sb.keys()
    for k in sa:
        for name in sa[k]:
            assert torch.equal(torch.as_tensor(sa[k][name]).float().cpu(), torch.as_tensor(sb[k][name]).float().cpu()), (k, name)


@pytest.mark.parametrize("with_rgb,with_eik", [(True, True), (False, False)])
def test_fused_surface_losses_equal_get_loss(device, with_rgb, with_eik):
    """csrc/surface_loss.hip against SurfaceModel.get_loss's torch formulas (base_surface_model.py:
    102-211 of the reference) evaluated in float64: every term, and the gradients of their sum with
    respect to depth, colour, SDF and SDF gradient; rays without a depth target, samples in front of,
    around and behind the surface, zero differences (sign(0) = 0)."""
    from ponderv2_amd import surface_loss
    from ponderv2_amd.ponder.models.ponder.render_utils.models.base_surface_model import SurfaceModel
    from ponderv2_amd.ponder.utils.config import ConfigDict

    torch.manual_seed(0)
    R, S = 301, 45
    weights = dict(depth_loss=1.0, free_space_loss=1.0, sdf_loss=10.0)
    if with_rgb:
        weights.update(rgb_loss=10.0)
    if with_eik:
        weights.update(eikonal_loss=0.01)
    loss_cfg = ConfigDict(dict(sensor_depth_truncation=0.05, weights=weights))
    model = SurfaceModel.__new__(SurfaceModel)
    torch.nn.Module.__init__(model)
    model.loss = loss_cfg
    z = torch.sort(torch.rand(R, S, 1, device=device) * 2.0, dim=1).values
    depth_gt = torch.rand(R, 1, device=device) * 2.0
    depth_gt[::7] = 0.0                                   # rays without a target
    leaves32 = dict(depth=torch.rand(R, 1, device=device) * 2.0, rgb=torch.rand(R, 3, device=device),
                    sdf=torch.randn(R, S, 1, device=device) * 0.1, gradients=torch.randn(R, S, 3, device=device))
    leaves32["depth"][5] = depth_gt[5]                    # an exact zero difference
    targets32 = dict(depth=depth_gt, rgb=torch.rand(R, 3, device=device))
    res = {}
    for dt in (torch.float32, torch.float64):
        leaves = {k: v.to(dt).clone().requires_grad_(True) for k, v in leaves32.items()}
        preds = dict(leaves, z_vals=z.to(dt))
        before = surface_loss.CALLS
        out = model.get_loss(preds, {k: v.to(dt) for k, v in targets32.items()})
        assert (surface_loss.CALLS > before) == (dt == torch.float32)
        total = sum(v for k, v in out.items() if "loss" in k)
        total.backward()
        res[dt] = ({k: float(v) for k, v in out.items()}, {k: v.grad for k, v in leaves.items()})
    a, b = res[torch.float32], res[torch.float64]
    assert a[0].keys() == b[0].keys() and ("psnr" in a[0]) == with_rgb
    for k in a[0]:
        assert abs(a[0][k] - b[0][k]) <= 2e-6 * (1 + abs(b[0][k])), (k, a[0][k], b[0][k])
    for k in ("depth", "sdf") + (("rgb",) if with_rgb else ()) + (("gradients",) if with_eik else ()):
        ga, gb = a[1][k], b[1][k]
        assert ga is not None and float((ga.double() - gb).abs().max()) <= 1e-6 * (float(gb.abs().max()) + 1e-12), k
