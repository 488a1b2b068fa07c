"""The product's trilinear sampler (csrc/trilinear.hip) against the REFERENCE'S OWN BINARY: the three
entry points of libs/smooth-sampler/smooth_sampler/csrc/smooth_sampler.cpp:36-98 compiled for gfx950
through hipify by oracle/build_ref_sampler.py (oracle/_ref/pv2_ref_smooth_sampler.so, built where a
reference checkout exists and shipped with the snapshot).  Same operands, fp32 and fp64, every padding
mode, both align_corners settings, with and without the smoothstep.  Skipped when the binary was never
built.  Run with -m gpu on an MI355X."""
import pytest
import torch

from helpers import away_from_kinks

pytestmark = pytest.mark.gpu
PAD = {"zeros": 0, "border": 1, "reflection": 2}


@pytest.fixture(scope="module")
def ref():
    from oracle import build_ref_sampler

    mod = build_ref_sampler.load()
    if mod is None:
        pytest.skip("oracle/_ref/pv2_ref_smooth_sampler.so not built (no reference checkout at build time)")
    return mod


def _close(a, b, tol, what):
    scale = b.abs().max().item() + 1e-30
    err = (a.double() - b.double()).abs().max().item() / scale
    assert err < tol, (what, err)
    return err


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-12), (torch.float32, 2e-5)])
@pytest.mark.parametrize("padding_mode", ["zeros", "border", "reflection"])
@pytest.mark.parametrize("align_corners", [True, False])
@pytest.mark.parametrize("smooth", [False, True])
def test_three_entry_points_equal_the_reference_binary(device, ref, dtype, tol, padding_mode,
                                                       align_corners, smooth):
    from ponderv2_amd import kernels as K

    if dtype == torch.float64 and smooth:
        # the reference evaluates its smoothstep in SINGLE precision whatever the tensor type
        # (smooth_sampler_kernel.cu:27-37 ``float smoothstep(float)``): in float64 it is only float-exact,
        # the product's double path is (measured 1.1e-7 apart on MI355X, everything else 1e-15)
        tol = 1e-6
    torch.manual_seed(17)
    B, C, D, H, W, R, S = 2, 12, 5, 7, 9, 33, 13
    vol = torch.randn(B, C, D, H, W, dtype=dtype)
    grid = torch.rand(B, 1, R, S, 3, dtype=torch.double) * 2.4 - 1.2
    grid = away_from_kinks(grid, (W, H, D), align_corners).to(dtype)
    vol, grid = vol.to(device), grid.to(device)
    pm = PAD[padding_mode]

    out_ref = ref.forward(vol, grid, pm, align_corners, smooth)
    out = K.trilinear_forward(vol, grid, padding_mode, align_corners, smooth)
    _close(out, out_ref, tol, "forward")

    gout = torch.randn_like(out_ref)
    gi_ref, gg_ref = ref.backward(gout, vol, grid, pm, align_corners, smooth, True)
    gi, gg = K.trilinear_backward(gout.contiguous(), vol, grid, padding_mode, align_corners, smooth, True)
    _close(gi, gi_ref, tol * 4, "backward: grad_input")
    _close(gg, gg_ref, tol * 4, "backward: grad_grid")

    ggi = torch.randn_like(vol)
    ggg = torch.randn_like(grid)
    r_in, r_grid, r_gout = ref.backward_backward(ggi, ggg, vol, grid, gout, pm, align_corners, smooth, True)
    p_in, p_grid, p_gout = K.trilinear_backward_backward(ggi, ggg, vol, grid, gout.contiguous(), padding_mode,
                                                         align_corners, smooth, True)
    _close(p_in, r_in, tol * 4, "backward_backward: grad_input")
    _close(p_gout, r_gout, tol * 4, "backward_backward: grad_grad_out")
    _close(p_grid, r_grid, tol * 16, "backward_backward: grad_grid")


def test_head_sized_volume_equals_the_reference_binary(device, ref):
    """The indoor head's shape class: 128 channels, channels-last on the product side (the vectorised
    kernels), a 24 x 40 x 48 volume, 4 096 points, smoothstep off as the shipped configs have it."""
    from ponderv2_amd import kernels as K

    torch.manual_seed(23)
    vol = torch.randn(1, 128, 24, 40, 48, device=device)
    grid = away_from_kinks(torch.rand(1, 1, 64, 64, 3, dtype=torch.double) * 2.2 - 1.1, (48, 40, 24), True)
    grid = grid.float().to(device)
    vol_cl = vol.contiguous(memory_format=torch.channels_last_3d)
    out_ref = ref.forward(vol, grid, 0, True, False)
    out = K.trilinear_forward(vol_cl, grid, "zeros", True, False)
    _close(out, out_ref, 2e-5, "forward")
    gout = torch.randn_like(out_ref)
    gi_ref, gg_ref = ref.backward(gout, vol, grid, 0, True, False, True)
    gi, gg = K.trilinear_backward(gout, vol_cl, grid, "zeros", True, False, True)
    _close(gi, gi_ref, 1e-4, "grad_input")
    _close(gg, gg_ref, 1e-4, "grad_grid")
