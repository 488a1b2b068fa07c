"""Dense 3x3x3 convolution kernels (csrc/dense_conv.hip, ponderv2_amd/dense_conv.py) against
``F.conv3d`` / ``F.conv_transpose3d`` in float64 on the host - forward, grad-input, grad-weight of
the two convolutions UNet3D-v1m2 is built from (reference: ponder/models/ponder/unet3d.py:45-156,
:359-493), with the fused BatchNorm / ReLU / skip-sum paths, ragged tiles and both tile shapes."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 2e-5   # fp32 MFMA (exact fp32 products, fp32 accumulation) against float64, relative to max|ref|


def rel(a, b):
    return (a.double().cpu() - b).abs().max().item() / (b.abs().max().item() + 1e-30)


def cl(t):
    return t.contiguous(memory_format=torch.channels_last_3d)


@pytest.mark.parametrize("shape,c_in,c_out", [
    ((2, 5, 7, 37), 32, 64),      # ragged in every axis, two channel chunks, two output blocks
    ((1, 4, 16, 16), 128, 32),    # the coarsest level's grid: X = 16 tiles
    ((1, 3, 6, 70), 16, 32),      # a single 16-channel chunk
    ((2, 4, 16, 16), 256, 128),   # the coarsest level at the bench's widths: split-K over the chunks (8 planes)
    ((1, 8, 128, 256), 32, 32),   # >= 262144 cells: the two-M-tiles-per-wave variant
])
def test_conv_forward_fused_paths(device, shape, c_in, c_out):
    from ponderv2_amd import dense_conv as dc

    torch.manual_seed(sum(shape) + c_in)
    b, z, y, x = shape
    vol = torch.randn(b, c_in, z, y, x)
    w = torch.randn(c_out, c_in, 3, 3, 3) / (27 * c_in) ** 0.5
    scale, shift = torch.rand(c_in) + 0.5, torch.randn(c_in) * 0.3
    bias = torch.randn(c_out)
    addend = torch.randn(b, c_out, z, y, x)
    mask = torch.randn(b, c_in, z, y, x)
    d = lambda t: cl(t.to(device))  # noqa: E731
    wd = w.to(device)
    packed = dc.pack_weights(wd, 0, False)
    # plain conv
    ref = F.conv3d(vol.double(), w.double(), padding=1)
    got = dc.conv3_forward(d(vol), packed, c_out, 0)
    assert got.shape == ref.shape and rel(got, ref) < TOL
    if shape[3] == 256:
        return
    # BatchNorm affine in front (zero padding AFTER it), ReLU behind
    xin = vol.double() * scale.double().view(1, -1, 1, 1, 1) + shift.double().view(1, -1, 1, 1, 1)
    ref = F.relu(F.conv3d(xin, w.double(), padding=1))
    got = dc.conv3_forward(d(vol), packed, c_out, 0, in_scale=scale.to(device),
                           in_shift=shift.to(device), relu=True)
    assert rel(got, ref) < TOL
    # masked input (ReLU backward), bias, addend; weights in channels-last storage
    packed_cl = dc.pack_weights(cl(wd), 0, False)
    assert torch.equal(packed_cl, packed)
    ref = F.conv3d(vol.double() * (mask > 0).double(), w.double(), bias.double(), padding=1) + addend.double()
    got = dc.conv3_forward(d(vol), packed_cl, c_out, 0, mask_src=d(mask), bias=bias.to(device),
                           addend=d(addend))
    assert rel(got, ref) < TOL


@pytest.mark.parametrize("shape,c_in,c_out", [((2, 5, 7, 37), 64, 32), ((1, 4, 16, 16), 32, 128)])
def test_conv_grad_input_is_the_flipped_conv(device, shape, c_in, c_out):
    from ponderv2_amd import dense_conv as dc

    torch.manual_seed(3)
    b, z, y, x = shape
    w = torch.randn(c_out, c_in, 3, 3, 3) / (27 * c_in) ** 0.5
    gy = torch.randn(b, c_out, z, y, x)
    vol = torch.randn(b, c_in, z, y, x, dtype=torch.double, requires_grad=True)
    F.conv3d(vol, w.double(), padding=1).backward(gy.double())
    packed_t = dc.pack_weights(w.to(device), 1, True)
    got = dc.conv3_forward(cl(gy.to(device)), packed_t, c_in, 0)
    assert rel(got, vol.grad) < TOL


@pytest.mark.parametrize("shape,c_in,c_out", [((2, 3, 5, 19), 64, 32), ((1, 4, 16, 16), 32, 64),
                                              ((2, 4, 16, 16), 256, 128)])   # (split-K planes in both directions)
def test_transposed_conv_forward_and_grad_input(device, shape, c_in, c_out):
    from ponderv2_amd import dense_conv as dc

    torch.manual_seed(5)
    b, z, y, x = shape
    w = torch.randn(c_in, c_out, 3, 3, 3) / (27 * c_in / 8) ** 0.5
    bias = torch.randn(c_out)
    vol = torch.randn(b, c_in, z, y, x, dtype=torch.double, requires_grad=True)
    skip = torch.randn(b, c_out, 2 * z, 2 * y, 2 * x)
    ref = F.conv_transpose3d(vol, w.double(), bias.double(), stride=2, padding=1, output_padding=1)
    assert ref.shape[2:] == (2 * z, 2 * y, 2 * x)
    gy = torch.randn_like(ref)
    ref.backward(gy)
    wd = w.to(device)
    got = dc.conv3_forward(cl(vol.detach().float().to(device)), dc.pack_weights(wd, 1, False, mode=1), c_out, 1,
                           bias=bias.to(device), addend=cl(skip.to(device)))
    assert rel(got, ref.detach() + skip.double()) < TOL
    gx = dc.conv3_forward(cl(gy.float().to(device)), dc.pack_weights(wd, 0, False, mode=2), c_in, 2)
    assert gx.shape == vol.shape and rel(gx, vol.grad) < TOL


@pytest.mark.parametrize("shape,c_in,c_out", [((2, 5, 7, 37), 32, 64), ((1, 4, 16, 16), 64, 32),
                                              ((2, 8, 32, 32), 32, 32)])
def test_conv_weight_gradient(device, shape, c_in, c_out):
    from ponderv2_amd import dense_conv as dc

    torch.manual_seed(7)
    b, z, y, x = shape
    vol = torch.randn(b, c_in, z, y, x)
    out = torch.randn(b, c_out, z, y, x)            # plays the conv's output (its sign = the ReLU mask)
    gy = torch.randn(b, c_out, z, y, x)
    scale, shift = torch.rand(c_in) + 0.5, torch.randn(c_in) * 0.3
    w = torch.zeros(c_out, c_in, 3, 3, 3, dtype=torch.double, requires_grad=True)
    xin = vol.double() * scale.double().view(1, -1, 1, 1, 1) + shift.double().view(1, -1, 1, 1, 1)
    F.conv3d(xin, w, padding=1).backward(gy.double() * (out > 0).double())
    like = torch.empty(c_out, c_in, 3, 3, 3, device=device)
    args = (cl(vol.to(device)), cl(gy.to(device)), like, 0)
    kw = dict(in_scale=scale.to(device), in_shift=shift.to(device), mask_src=cl(out.to(device)))
    got = dc.conv3_backward_weight(*args, **kw)
    assert rel(got, w.grad) < TOL
    again = dc.conv3_backward_weight(*args, **kw)
    assert torch.equal(got, again)                  # fixed summation order: bitwise repeatable
    got_cl = dc.conv3_backward_weight(args[0], args[1], cl(like), 0, **kw)   # channels-last weight
    assert got_cl.stride() == cl(like).stride() and torch.equal(got_cl, got)


@pytest.mark.parametrize("shape,c_in,c_out", [((2, 3, 5, 19), 64, 32), ((1, 4, 16, 16), 32, 64)])
def test_transposed_conv_weight_gradient(device, shape, c_in, c_out):
    from ponderv2_amd import dense_conv as dc

    torch.manual_seed(9)
    b, z, y, x = shape
    vol = torch.randn(b, c_in, z, y, x)
    gy = torch.randn(b, c_out, 2 * z, 2 * y, 2 * x)
    w = torch.zeros(c_in, c_out, 3, 3, 3, dtype=torch.double, requires_grad=True)
    F.conv_transpose3d(vol.double(), w, stride=2, padding=1, output_padding=1).backward(gy.double())
    like = torch.empty(c_in, c_out, 3, 3, 3, device=device)
    got = dc.conv3_backward_weight(cl(vol.to(device)), cl(gy.to(device)), like, 1, n_dim=1)
    assert rel(got, w.grad) < TOL


def test_bn_conv_relu_unit_and_upsample_add_autograd(device):
    """The two autograd units of the dense U-Net against the stock modules in float64: values,
    input / weight / BatchNorm gradients, running statistics."""
    import copy

    from ponderv2_amd import dense_conv as dc

    torch.manual_seed(11)
    bn = torch.nn.BatchNorm3d(32)
    conv = torch.nn.Conv3d(32, 64, 3, padding=1, bias=False)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.3)
    x = torch.randn(2, 32, 4, 6, 34)
    probe = torch.randn(2, 64, 4, 6, 34)
    bn64, conv64 = copy.deepcopy(bn).double(), copy.deepcopy(conv).double()
    x64 = x.double().requires_grad_(True)
    ref = F.relu(conv64(bn64(x64)))
    (ref * probe.double()).sum().backward()
    bn_d, conv_d = bn.to(device), conv.to(device)
    xd = cl(x.to(device)).requires_grad_(True)
    assert dc.bn_conv_supported(bn_d, conv_d, xd)
    got = dc.bn_conv_relu(bn_d, conv_d, xd)
    (got * cl(probe.to(device))).sum().backward()
    torch.cuda.synchronize()
    assert rel(got, ref.detach()) < TOL
    assert rel(xd.grad, x64.grad) < 1e-4
    assert rel(conv_d.weight.grad, conv64.weight.grad) < 1e-4
    assert rel(bn_d.weight.grad, bn64.weight.grad) < 1e-4
    assert rel(bn_d.bias.grad, bn64.bias.grad) < 1e-4
    assert rel(bn_d.running_mean, bn64.running_mean) < 1e-5
    assert rel(bn_d.running_var, bn64.running_var) < 1e-5
    assert int(bn_d.state_dict()["num_batches_tracked"]) == 1

    up = torch.nn.ConvTranspose3d(64, 32, 3, stride=2, padding=1)
    xs = torch.randn(2, 64, 2, 3, 17)
    skip = torch.randn(2, 32, 4, 6, 34)
    probe = torch.randn(2, 32, 4, 6, 34)
    up64 = copy.deepcopy(up).double()
    xs64, skip64 = xs.double().requires_grad_(True), skip.double().requires_grad_(True)
    ref = skip64 + up64(xs64, output_size=[4, 6, 34])
    (ref * probe.double()).sum().backward()
    up_d = up.to(device)
    xsd, skd = cl(xs.to(device)).requires_grad_(True), cl(skip.to(device)).requires_grad_(True)
    assert dc.upsample_supported(up_d, xsd, [4, 6, 34])
    got = dc.upsample_add(up_d, skd, xsd)
    (got * cl(probe.to(device))).sum().backward()
    torch.cuda.synchronize()
    assert rel(got, ref.detach()) < TOL
    assert rel(xsd.grad, xs64.grad) < TOL and rel(skd.grad, skip64.grad) < 1e-6
    assert rel(up_d.weight.grad, up64.weight.grad) < TOL
    assert rel(up_d.bias.grad, up64.bias.grad) < TOL
