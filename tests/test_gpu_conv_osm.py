"""GPU tests of the mask-grouped output-stationary sparse conv (csrc/sparse_conv_osm.hip): the plan
against a numpy restatement, forward / grad-input against the float64 oracle for every (NB, WR)
configuration the dispatcher can pick, the block statistics against torch, bit-for-bit repeatability,
and the conv + BatchNorm unit on this route against the product-row route.  Run with -m gpu on an MI355X."""
import numpy as np
import pytest
import torch

from helpers import random_voxels

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


def _oracle_conv(feats, w, pin, pout, ks, n_out):
    from oracle.sparse_ops import sparse_conv

    return sparse_conv(feats, w, torch.from_numpy(pin.astype(np.int64)),
                       torch.from_numpy(pout.astype(np.int64)), ks, n_out)


def _rel(a, b):
    return (a.double().cpu() - b.double().cpu()).abs().max().item() / (b.abs().max().item() + 1e-12)


def _check_plan(plan, tbl, K, n, stride):
    """The device plan equals oracle/osm_plan.py's restatement bit for bit: perm sorts the rows stably by
    their offset mask, tblp is the table in that order (padding -1), tmask the OR over 32 sorted rows."""
    from oracle.osm_plan import osm_plan

    n_pad = plan.n_pad
    assert n_pad % 256 == 0 and n_pad >= n
    perm, tblp, tmask = osm_plan(_np(tbl).reshape(K, stride), n, n_pad)
    assert np.array_equal(np.sort(_np(plan.perm)[:n]), np.arange(n))
    assert np.array_equal(_np(plan.perm)[:n], perm)
    assert np.array_equal(_np(plan.tblp).reshape(K, n_pad), tblp)
    assert np.array_equal(_np(plan.tmask).view(np.uint32), tmask)


@pytest.mark.parametrize("seed,batch,n", [(0, 2, 1500), (1, 1, 4000), (2, 3, 3)])
def test_osm_plans_match_their_definition(device, seed, batch, n, monkeypatch):
    from ponderv2_amd import kernels as K

    monkeypatch.setattr(K, "OSM_MODE", "1")
    coords = torch.from_numpy(random_voxels(seed, batch=batch, n_per_batch=n)).to(device)
    rb3 = K.build_subm_rulebook(coords, 3)
    _check_plan(rb3.osm, rb3.nbr, 27, rb3.n_out, rb3.nbr_stride)
    assert rb3.osm_t.kflip == 1 and rb3.osm_t.tblp is rb3.osm.tblp
    shape = [(s - 2) // 2 + 1 for s in (40 + 96, 36 + 96, 20 + 96)]
    rbd, _ = K.build_downsample_rulebook(coords, 2, shape)
    _check_plan(rbd.osm, rbd.nbr, 8, rbd.n_out, rbd.nbr_stride)
    parent = rbd._transposed_os[0]
    _check_plan(rbd.osm_t, parent, 8, rbd.n_in, rbd._transposed_os[1])


@pytest.mark.parametrize("nb,wr", [(0, 0), (1, 2), (1, 4), (2, 2), (2, 4), (3, 2), (3, 4), (4, 2), (4, 4)])
@pytest.mark.parametrize("c_in,c_out", [(32, 32), (64, 128), (96, 96), (256, 192)])
def test_osm_conv_vs_oracle_every_configuration(device, c_in, c_out, nb, wr, monkeypatch):
    """Forward, grad-input (+ addend, also in place) and the block statistics of a submanifold conv for
    every (column group, row tiles) shape of the kernel; (0, 0) is the dispatcher's own choice."""
    from ponderv2_amd import _lib, kernels as K

    monkeypatch.setattr(K, "OSM_MODE", "1")

    if nb and ((c_out // 32) % nb or (c_in // 32) % nb):
        pytest.skip("column group does not divide the channel blocks")
    _lib.lib().pv2_debug_set_osm(-1, nb, wr, 0)
    try:
        _conv_case(device, c_in, c_out)
    finally:
        _lib.lib().pv2_debug_set_osm(-1, 0, 0, 0)


def _conv_case(device, c_in, c_out):
    from oracle import rulebook as orb
    from ponderv2_amd import kernels as K

    torch.manual_seed(c_in * 1000 + c_out)
    coords = random_voxels(5, batch=2, n_per_batch=1100)
    n = len(coords)
    feats = torch.randn(n, c_in)
    w = torch.randn(c_out, 27, c_in) * 0.1
    gout = torch.randn(n, c_out)
    addend = torch.randn(n, c_in)
    pin, pout, ks = orb.subm_rulebook(coords, 3)
    f_ref = feats.double().requires_grad_(True)
    ref = _oracle_conv(f_ref, w.double(), pin, pout, ks, n)
    ref.backward(gout.double())

    rb = K.build_subm_rulebook(torch.from_numpy(coords).to(device), 3)
    assert rb.osm is not None
    out, partial, blocks, rpb = K.spconv_osm(feats.to(device), w.to(device), rb, stats=True)
    assert _rel(out, ref.detach()) < 1e-5
    dx = K.spconv_osm(gout.to(device), w.to(device), rb, transposed=True, addend=addend.to(device))
    assert _rel(dx, f_ref.grad + addend.double()) < 1e-5
    # in place: the addend is the output buffer
    buf = addend.to(device).clone()
    from ponderv2_amd import _lib
    import ctypes
    blocks_c, rpb_c = ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(_lib.lib().pv2_spconv_osm(
        K._ptr(gout.to(device)), c_out, K._ptr(w.to(device)), 27, c_in, 1, ctypes.byref(rb.osm_t.struct), n,
        K._ptr(K.zero_row(device)), K._ptr(buf), K._ptr(buf), None, ctypes.byref(blocks_c),
        ctypes.byref(rpb_c), K._stream(buf)), "pv2_spconv_osm")
    assert torch.equal(buf, dx)
    # block statistics: sums and centred sums of squares of each block of `rpb` SORTED rows
    assert blocks == (n + rpb - 1) // rpb and rpb in (64, 128)
    perm = rb.osm.perm[:n].long()
    srt = out[perm].double()
    part = partial[:blocks * 2 * c_out].view(blocks, 2, c_out).double()
    for b in range(blocks):
        rows = srt[b * rpb:(b + 1) * rpb]
        assert torch.allclose(part[b, 0], rows.sum(0), rtol=1e-5, atol=1e-4)
        assert torch.allclose(part[b, 1], ((rows - rows.mean(0)) ** 2).sum(0), rtol=1e-4, atol=1e-4)
    # identical bits on every call
    again = K.spconv_osm(feats.to(device), w.to(device), rb)
    assert torch.equal(out, again)


def test_osm_strided_and_inverse_conv_vs_oracle(device, monkeypatch):
    from oracle import rulebook as orb
    from ponderv2_amd import kernels as K

    monkeypatch.setattr(K, "OSM_MODE", "1")
    torch.manual_seed(7)
    coords = random_voxels(6, batch=2, n_per_batch=2500)
    n = len(coords)
    shape = [68, 66, 58]
    ooc, pin, pout, ks = orb.downsample_rulebook(coords, 2, shape)
    m = len(ooc)
    c_in, c_out = 32, 64
    feats, w = torch.randn(n, c_in), torch.randn(c_out, 8, c_in) * 0.1
    w_inv = torch.randn(96, 8, c_out) * 0.1
    g_up = torch.randn(n, 96)
    rb, _ = K.build_downsample_rulebook(torch.from_numpy(coords).to(device), 2, shape)
    rbt = rb.transposed()
    assert rb.osm is not None and rbt.osm is rb.osm_t
    d_ref = feats.double().requires_grad_(True)
    ref_down = _oracle_conv(d_ref, w.double(), pin, pout, ks, m)
    down = K.spconv_osm(feats.to(device), w.to(device), rb)
    assert _rel(down, ref_down.detach()) < 1e-5
    u_ref = ref_down.detach().clone().requires_grad_(True)
    ref_up = _oracle_conv(u_ref, w_inv.double(), pout, pin, ks, n)
    up = K.spconv_osm(down, w_inv.to(device), rbt)
    assert _rel(up, ref_up.detach()) < 1e-5
    ref_up.backward(g_up.double())
    dx_up = K.spconv_osm(g_up.to(device), w_inv.to(device), rbt, transposed=True)
    assert _rel(dx_up, u_ref.grad) < 1e-5
    g_down = torch.randn(m, c_out)
    ref_down.backward(g_down.double())
    dx_down = K.spconv_osm(g_down.to(device), w.to(device), rb, transposed=True)
    assert _rel(dx_down, d_ref.grad) < 1e-5


def test_osm_rows_without_any_neighbour_are_written(device, monkeypatch):
    """Grad-input of a strided conv whose out_shape drops some inputs: those rows have no parent under
    any offset and must come out as zeros (+ addend), not stay uninitialised."""
    from ponderv2_amd import kernels as K

    monkeypatch.setattr(K, "OSM_MODE", "1")
    coords = random_voxels(3, batch=1, n_per_batch=1500)
    shape = [12, 11, 9]   # smaller than the extent: inputs beyond it have no output voxel
    rb, _ = K.build_downsample_rulebook(torch.from_numpy(coords).to(device), 2, shape)
    n, m = len(coords), rb.n_out
    assert rb.n_pairs < n
    g = torch.randn(m, 64, device=device)
    w = torch.randn(64, 8, 32, device=device) * 0.1
    dx = K.spconv_osm(g, w, rb, transposed=True)
    paired = torch.zeros(n, dtype=torch.bool, device=device)
    paired[rb.pair_in.long()] = True
    assert (dx[~paired] == 0).all() and (~paired).any()
    ref = K.spconv_grad_input(g, w, rb)
    assert _rel(dx, ref) < 2e-5


@pytest.mark.parametrize("kind", ["subm", "down", "up"])
def test_conv_bn_unit_on_the_osm_route_equals_the_product_row_route(device, kind, monkeypatch):
    """pv2_convbn_forward / _backward with a planned rulebook against the same unit with the switch off:
    outputs, every gradient and the running statistics."""
    from ponderv2_amd import _lib, kernels as K

    monkeypatch.setattr(K, "OSM_MODE", "1")
    outs = {}
    try:
        for mode in (1, 0):
            _lib.lib().pv2_debug_set_osm(mode, -1, -1, 0)
            outs[mode] = _unit_case(kind)
    finally:
        _lib.lib().pv2_debug_set_osm(2, -1, -1, 0)
    names = ("out", "dx", "dw", "d bn weight", "d bn bias", "d residual", "running_mean", "running_var")
    for name, a, b in zip(names, outs[1], outs[0]):
        assert _rel(a, b) < 3e-5, (name, _rel(a, b), float(a.abs().max()), float(b.abs().max()))


def _unit_case(kind):
    import torch.nn as nn
    from ponderv2_amd import convbn, kernels as K

    device = torch.device("cuda:0")
    torch.manual_seed(3)
    coords = torch.from_numpy(random_voxels(4, batch=2, n_per_batch=2500)).to(device)
    shape = [68, 66, 58]
    if kind == "subm":
        rb = K.build_subm_rulebook(coords, 3)
        c_in, c_out = 64, 96
    else:
        rb, _ = K.build_downsample_rulebook(coords, 2, shape)
        c_in, c_out = (32, 64) if kind == "down" else (64, 32)
        if kind == "up":
            rb = rb.transposed()
    assert rb.osm is not None and rb.osm_t is not None
    x = torch.randn(rb.n_in, c_in, device=device, requires_grad=True)
    w = (torch.randn(c_out, rb.K, c_in, device=device) * 0.1).requires_grad_(True)
    bn = nn.BatchNorm1d(c_out).to(device)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    res = torch.randn(rb.n_out, c_out, device=device, requires_grad=True)
    assert convbn.supported(x, w, rb, bn)
    out = convbn.ConvBNFunction.apply(x, w, bn.weight, bn.bias, res, rb, bn.running_mean, bn.running_var,
                                      True, bn.eps, bn.momentum)
    g = torch.randn_like(out)
    out.backward(g)
    torch.cuda.synchronize()
    return [t.detach().clone() for t in (out, x.grad, w.grad, bn.weight.grad, bn.bias.grad, res.grad,
                                         bn.running_mean, bn.running_var)]
