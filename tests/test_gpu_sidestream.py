"""The backward side stream (ponderv2_amd/sidestream.py) and the zero arenas (kernels.py) change
WHEN and WHERE kernels run, never what they compute: the same training steps with both switched
off and on must give the same losses and the same gradient for every parameter."""
import pytest
import torch

import golden_cases as gc

pytestmark = pytest.mark.gpu


def _steps(device, monkeypatch, side, arena, n_steps=3):
    from ponderv2_amd import kernels as K, sidestream

    monkeypatch.setattr(K, "USE_OS", True)          # deterministic forward (see test_gpu_golden)
    monkeypatch.setattr(K, "USE_ZERO_ARENA", arena)
    monkeypatch.setattr(sidestream, "ENABLED", side)
    K._ARENAS.clear()
    model, batch = gc.small_indoor(device)
    losses = []
    for step in range(n_steps):                      # the arenas open from the second step on
        torch.manual_seed(step)
        model.zero_grad(set_to_none=True)
        out = model({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
        out["loss"].backward()
        losses.append(float(out["loss"]))
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    used = {pool: a.used for (_, pool), a in K._ARENAS.items()}
    return losses, grads, used


def test_side_stream_and_arenas_do_not_change_the_step(device, monkeypatch):
    from ponderv2_amd import sidestream

    base_l, base_g, used0 = _steps(device, monkeypatch, side=False, arena=False)
    forks = []
    orig = sidestream.fork
    monkeypatch.setattr(sidestream, "fork", lambda fn, reads: (forks.append(1), orig(fn, reads))[1])
    new_l, new_g, used1 = _steps(device, monkeypatch, side=True, arena=True)
    # both mechanisms really ran: weight gradients forked (sparse convs + the dense U-Net's library
    # convs), slices drawn from every pool in the last step
    # (in the deterministic mode only the weight gradients still accumulate by atomics: "dw")
    assert len(forks) >= 3 * 20, len(forks)
    assert not any(used0.values()) and used1["dw"] > 0, (used0, used1)
    for a, b in zip(base_l, new_l):
        assert abs(a - b) <= 1e-5 * abs(a), (base_l, new_l)
    assert base_g.keys() == new_g.keys()
    worst = {}
    for name, g0 in base_g.items():
        if name.endswith("upsampling.upsample.bias"):
            continue    # a bias in front of a BatchNorm: its gradient is zero up to rounding noise
        worst[name] = gc.rel_err(new_g[name], g0.cpu().numpy())
    bad = {k: v for k, v in worst.items() if not v < 2e-3}
    assert not bad, bad


def test_accumulating_gradients_stay_on_the_main_stream(device, monkeypatch):
    """With ``param.grad`` already set autograd ADDS the new gradient on the main stream as soon as
    the node returns; such weight gradients must not be forked (sidestream.safe_leaf)."""
    from ponderv2_amd import kernels as K, sidestream

    monkeypatch.setattr(K, "USE_OS", True)
    monkeypatch.setattr(sidestream, "ENABLED", True)
    model, batch = gc.small_indoor(device)
    forks = []
    orig = sidestream.fork
    monkeypatch.setattr(sidestream, "fork", lambda fn, reads: (forks.append(1), orig(fn, reads))[1])
    grads = []
    for micro in range(2):      # no zero_grad in between: the second backward accumulates
        torch.manual_seed(0)
        out = model({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
        out["loss"].backward()
        if micro == 0:
            first = len(forks)
            torch.cuda.synchronize()
            grads = {n: p.grad.detach().clone() for n, p in model.named_parameters()
                     if p.grad is not None}
    torch.cuda.synchronize()
    assert first > 0 and len(forks) == first, (first, len(forks))
    worst = {n: gc.rel_err(p.grad, 2.0 * grads[n].cpu().numpy())
             for n, p in model.named_parameters()
             if p.grad is not None and not n.endswith("upsampling.upsample.bias")}
    bad = {k: v for k, v in worst.items() if not v < 2e-3}
    assert not bad, bad


def test_zero_arena_slices_behave_like_fresh_zero_buffers(device, monkeypatch):
    """Scatter-add convs (forward, grad-input, grad-weight) drawing their cleared targets from the
    step arenas give what they give on individually cleared buffers, and every pool is used."""
    from helpers import random_voxels
    from ponderv2_amd import kernels as K, sidestream

    monkeypatch.setattr(K, "USE_OS", False)           # scatter-add everywhere
    monkeypatch.setattr(sidestream, "ENABLED", False)
    coords = torch.from_numpy(random_voxels(7, batch=2, n_per_batch=3000)).to(device)
    rb = K.build_subm_rulebook(coords, 3)
    torch.manual_seed(0)
    x0 = torch.randn(rb.n_in, 32, device=device)
    ws = [torch.randn(c_out, 27, c_in, device=device) * 0.1 for c_in, c_out in ((32, 64), (64, 64), (64, 32))]

    def run(arena):
        monkeypatch.setattr(K, "USE_ZERO_ARENA", arena)
        K._ARENAS.clear()
        res = None
        for step in range(3):
            K.begin_zero_arenas(device)
            x = x0.clone().requires_grad_(True)
            w = [t.clone().requires_grad_(True) for t in ws]
            h = x
            for t in w:
                h = K.SparseConvFunction.apply(h, t, rb)
            h.square().sum().backward()
            res = [h.detach(), x.grad] + [t.grad for t in w]
        torch.cuda.synchronize()
        return res, {pool: a.used for (_, pool), a in K._ARENAS.items()}

    ref, used0 = run(False)
    got, used1 = run(True)
    assert not used0 and all(used1[p] > 0 for p in K.ZERO_POOLS), (used0, used1)
    for a, b in zip(got, ref):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())
