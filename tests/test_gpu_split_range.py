"""Range / special-value tests of the bf16-PIECE matrix products (csrc/mfma_split.h): every fp32 product
of the sparse and dense convolutions runs as six bf16 MFMAs over three bf16 pieces per operand, and the
round-4 tests fed them randn only.  Here each kernel family - sparse forward,
sparse grad-input (transposed weight read), sparse weight gradient, dense conv k3 s1, transposed conv,
strided conv, dense weight gradient - gets

  * operands spanning 2^-60 .. 2^60 inside one reduction,
  * heavy cancellation (channel pairs +v / -v against almost equal weights),
  * operands next to FLT_MAX whose products are representable (no spurious overflow in a piece),
  * operands next to FLT_MIN and fp32 subnormals (the third / second piece leaves the bf16 normal range),
  * Inf and NaN inputs.

Finite cases are held to ``|got - ref| <= 2e-6 * sum |a b|`` per output element against float64 - the
bound a correctly rounded fp32 dot product of this length meets, and the honest one under cancellation
(an error relative to max|ref| hides it).  Tiny operands: the measured behaviour is asserted and printed
(-s): below 2^-110 the smallest pieces are bf16 subnormals.  Non-finite inputs: the SET of non-finite
outputs equals the float64 reference's; an Inf may surface as NaN (inf * 0-piece), documented in
mfma_split.h - a GradScaler's isfinite test sees the same thing.  Run with -m gpu on an MI355X."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import random_voxels

pytestmark = pytest.mark.gpu
BOUND = 2e-6


def cl(t):
    return t.contiguous(memory_format=torch.channels_last_3d)


def wide(shape, gen, lo, hi):
    """sign * 2^e * (1 + u), e uniform in [lo, hi]: every binade of the range appears in a reduction"""
    e = torch.empty(shape).uniform_(lo, hi, generator=gen)
    m = 1.0 + torch.rand(shape, generator=gen)
    s = torch.where(torch.rand(shape, generator=gen) < 0.5, -1.0, 1.0)
    return (s * m * torch.exp2(e)).float()


def cancelling(shape, gen):
    """last-axis pairs (+v, -v): against equal weights the two products cancel exactly"""
    v = torch.randn(shape, generator=gen)
    out = v.clone()
    out[..., 1::2] = -v[..., 0::2][..., :out[..., 1::2].shape[-1]]
    return out.float()


def paired_weights(shape, gen, axis):
    """weights equal within channel pairs along ``axis`` up to a 2^-12 relative perturbation"""
    w = torch.randn(shape, generator=gen)
    w = w.movedim(axis, -1).clone()
    w[..., 1::2] = w[..., 0::2][..., :w[..., 1::2].shape[-1]] * (1 + 2.0 ** -12 * torch.randn(
        w[..., 1::2].shape, generator=gen))
    return w.movedim(-1, axis).contiguous().float()


def check(got, ref, refabs, what, bound=BOUND):
    got = got.double().cpu()
    assert torch.isfinite(got).all(), what
    err = ((got - ref).abs() / (refabs + 1e-300)).max().item()
    assert err < bound, (what, err)
    return err


# ------------------------------------------------------------------ sparse convolutions
def _sparse_setup(device, c_in, c_out, seed=5):
    from oracle import rulebook as orb
    from oracle.sparse_ops import sparse_conv
    from ponderv2_amd import kernels as K

    coords = random_voxels(seed, batch=2, n_per_batch=900)
    pin, pout, ks = orb.subm_rulebook(coords, 3)
    rb = K.build_subm_rulebook(torch.from_numpy(coords).to(device), 3)
    pin_t, pout_t = torch.from_numpy(pin.astype(np.int64)), torch.from_numpy(pout.astype(np.int64))

    def fwd(x, w):
        return sparse_conv(x, w, pin_t, pout_t, ks, len(coords))

    def dgrad(g, w):   # d/dx of fwd: the conv over swapped pair roles with the transposed weight
        return sparse_conv(g, w.permute(2, 1, 0).contiguous(), pout_t, pin_t, ks, len(coords))

    def wgrad(x, g):
        w = torch.zeros(c_out, 27, c_in, dtype=torch.double, requires_grad=True)
        fwd(x, w).backward(g)
        return w.grad

    return K, rb, len(coords), fwd, dgrad, wgrad


@pytest.mark.parametrize("case", ["wide", "cancel", "flt_max"])
@pytest.mark.parametrize("c_in,c_out", [(64, 96), (128, 128)])
def test_sparse_split_products_over_the_range(device, case, c_in, c_out):
    K, rb, n, fwd, dgrad, wgrad = _sparse_setup(device, c_in, c_out)
    gen = torch.Generator().manual_seed(c_in + len(case))
    if case == "wide":
        x, g = wide((n, c_in), gen, -60, 60), wide((n, c_out), gen, -60, 60)
        w = wide((c_out, 27, c_in), gen, -4, 4)
    elif case == "cancel":
        x, g = cancelling((n, c_in), gen), cancelling((n, c_out), gen)
        w = paired_weights((c_out, 27, c_in), gen, 2)
    else:   # |x| up to 1.99 * 2^126, weights 2^-9 .. 2^-5: every product and every sum is finite in fp32
        x, g = wide((n, c_in), gen, 120, 126), wide((n, c_out), gen, 120, 126)
        w = wide((c_out, 27, c_in), gen, -16, -14)
    xd, gd, wd = x.to(device), g.to(device), w.to(device)
    x64, g64, w64 = x.double(), g.double(), w.double()
    errs = [check(K.spconv_forward(xd, wd, rb), fwd(x64, w64), fwd(x64.abs(), w64.abs()), case + " forward")]
    wt = w64 if case != "cancel" else w64
    errs.append(check(K.spconv_grad_input(gd, wd, rb), dgrad(g64, wt), dgrad(g64.abs(), wt.abs()),
                      case + " grad-input"))
    if case != "flt_max":   # (x * g would overflow)
        errs.append(check(K.spconv_backward_weight(xd, gd, rb, c_out), wgrad(x64, g64),
                          wgrad(x64.abs(), g64.abs()), case + " weight gradient"))
    print("sparse %s %d->%d: max err / sum|ab| = %s" % (case, c_in, c_out, ["%.2e" % e for e in errs]))


def test_sparse_split_products_of_tiny_operands(device):
    """Operands below 2^-110 (their third / second bf16 piece is a bf16 subnormal) and fp32 subnormals
    against large weights: the products are ordinary numbers.  Asserted: the result is finite and within
    2^-7 of sum|ab| (the leading piece always survives); printed: what was actually measured."""
    K, rb, n, fwd, _, _ = _sparse_setup(device, 64, 64)
    gen = torch.Generator().manual_seed(77)
    for name, lo, hi, bound in (("2^-120..2^-112", -120, -112, 2.0 ** -7),
                                # fp32 SUBNORMAL features: the matrix pipe flushes subnormal bf16 inputs, so
                                # such a term may vanish altogether - anything between the exact product
                                # and zero is accepted (measured on MI355X: 0.15 of sum|ab|), finite always
                                ("fp32 subnormal 2^-140..2^-128", -140, -128, 1.0 + 1e-6)):
        x = wide((n, 64), gen, lo, hi)
        w = wide((64, 27, 64), gen, 90, 96)
        got = K.spconv_forward(x.to(device), w.to(device), rb)
        err = check(got, fwd(x.double(), w.double()), fwd(x.double().abs(), w.double().abs()),
                    "tiny " + name, bound=bound)
        print("sparse forward, features %s: max err / sum|ab| = %.2e" % (name, err))


@pytest.mark.parametrize("value", [float("inf"), float("-inf"), float("nan")])
def test_sparse_split_products_propagate_non_finite_inputs(device, value):
    K, rb, n, fwd, dgrad, wgrad = _sparse_setup(device, 64, 64)
    gen = torch.Generator().manual_seed(3)
    x, g = torch.randn(n, 64, generator=gen), torch.randn(n, 64, generator=gen)
    w = torch.randn(64, 27, 64, generator=gen) * 0.1
    x[17, 5] = value
    g[40, 9] = value
    x[0, 33] = value     # row 0 is what padding pairs of the weight-gradient tiles read
    g[0, 2] = value
    with np.errstate(all="ignore"):
        ref_f, ref_b = fwd(x.double(), w.double()), dgrad(g.double(), w.double())
        ref_w = wgrad(x.double(), g.double())
    got_f = K.spconv_forward(x.to(device), w.to(device), rb).cpu()
    got_b = K.spconv_grad_input(g.to(device), w.to(device), rb).cpu()
    got_w = K.spconv_backward_weight(x.to(device), g.to(device), rb, 64).cpu()
    for got, ref, what in ((got_f, ref_f, "forward"), (got_b, ref_b, "grad-input"), (got_w, ref_w, "wgrad")):
        bad_ref = ~torch.isfinite(ref)
        assert bad_ref.any()
        assert torch.equal(~torch.isfinite(got), bad_ref), what
        ok = ~bad_ref
        scale = ref[ok].abs().max().item()
        assert (got[ok].double() - ref[ok]).abs().max().item() < 1e-5 * scale, what


# ------------------------------------------------------------------ dense convolutions
def _dense_refs(mode, vol, w):
    if mode == 0:
        return F.conv3d(vol, w, padding=1)
    if mode == 1:
        return F.conv_transpose3d(vol, w, stride=2, padding=1, output_padding=1)
    return F.conv3d(vol, w, stride=2, padding=1)


def _dense_run(dc, mode, vol_d, w_d, c_out):
    if mode == 0:
        return dc.conv3_forward(vol_d, dc.pack_weights(w_d, 0, False), c_out, 0)
    if mode == 1:    # weight [c_in, c_out, 3,3,3]
        return dc.conv3_forward(vol_d, dc.pack_weights(w_d, 1, False, mode=1), c_out, 1)
    return dc.conv3_forward(vol_d, dc.pack_weights(w_d, 0, False, mode=2), c_out, 2)


@pytest.mark.parametrize("case", ["wide", "cancel", "flt_max"])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_dense_split_products_over_the_range(device, mode, case):
    from ponderv2_amd import dense_conv as dc

    gen = torch.Generator().manual_seed(11 * mode + len(case))
    c_in, c_out = 64, 32
    shape = (2, c_in, 4, 6, 18)
    wshape = (c_in, c_out, 3, 3, 3) if mode == 1 else (c_out, c_in, 3, 3, 3)
    red_axis = 0 if mode == 1 else 1
    if case == "wide":
        vol = wide(shape, gen, -60, 60)
        w = wide(wshape, gen, -4, 4)
    elif case == "cancel":
        vol = cancelling((2, 4, 6, 18, c_in), gen).permute(0, 4, 1, 2, 3).contiguous()
        w = paired_weights(wshape, gen, red_axis)
    else:
        vol = wide(shape, gen, 120, 126)
        w = wide(wshape, gen, -16, -14)
    got = _dense_run(dc, mode, cl(vol.to(device)), w.to(device), c_out)
    ref = _dense_refs(mode, vol.double(), w.double())
    refabs = _dense_refs(mode, vol.double().abs(), w.double().abs())
    err = check(got, ref, refabs, "dense mode %d %s" % (mode, case))
    print("dense mode %d %s: max err / sum|ab| = %.2e" % (mode, case, err))


@pytest.mark.parametrize("case", ["wide", "cancel"])
@pytest.mark.parametrize("mode", [0, 1])
def test_dense_weight_gradient_over_the_range(device, mode, case):
    """mode 0: the bf16-piece weight gradient; mode 1 (transposed conv): the fp32-MFMA one, same bound."""
    from ponderv2_amd import dense_conv as dc

    gen = torch.Generator().manual_seed(5 + mode)
    c_in, c_out = 32, 64
    b, z, y, x = 2, 3, 5, 19
    gshape = (b, c_out, 2 * z, 2 * y, 2 * x) if mode == 1 else (b, c_out, z, y, x)
    if case == "wide":   # (the product of the two exponents stays below 2^100, 2850+ terms per weight)
        vol, gy = wide((b, c_in, z, y, x), gen, -40, 40), wide(gshape, gen, -40, 40)
    else:
        vol = cancelling((b, c_in, z, y, x), gen)
        gy = torch.ones(gshape) * (1 + 2.0 ** -12 * torch.randn(gshape, generator=gen))
    wz = torch.zeros((c_in, c_out, 3, 3, 3) if mode == 1 else (c_out, c_in, 3, 3, 3), dtype=torch.double)

    def ref_of(v, g):
        w = wz.clone().requires_grad_(True)
        _dense_refs(mode, v, w).backward(g)
        return w.grad

    ref, refabs = ref_of(vol.double(), gy.double()), ref_of(vol.double().abs(), gy.double().abs())
    like = torch.empty(wz.shape, device=device)
    got = dc.conv3_backward_weight(cl(vol.to(device)), cl(gy.to(device)), like, mode, n_dim=1 if mode == 1 else 0)
    err = check(got, ref, refabs, "dense wgrad mode %d %s" % (mode, case))
    print("dense weight gradient mode %d %s: max err / sum|ab| = %.2e" % (mode, case, err))


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("value", [float("inf"), float("nan")])
def test_dense_split_products_propagate_non_finite_inputs(device, mode, value):
    from ponderv2_amd import dense_conv as dc

    gen = torch.Generator().manual_seed(21 + mode)
    c_in, c_out = 32, 32
    vol = torch.randn(1, c_in, 4, 6, 18, generator=gen)
    w = torch.randn((c_in, c_out, 3, 3, 3) if mode == 1 else (c_out, c_in, 3, 3, 3), generator=gen) * 0.1
    vol[0, 7, 2, 3, 9] = value
    got = _dense_run(dc, mode, cl(vol.to(device)), w.to(device), c_out).cpu()
    ref = _dense_refs(mode, vol.double(), w.double())
    bad = ~torch.isfinite(ref)
    assert bad.any() and torch.equal(~torch.isfinite(got), bad)
    ok = ~bad
    assert (got[ok].double() - ref[ok]).abs().max().item() < 1e-5 * ref[ok].abs().max().item()
