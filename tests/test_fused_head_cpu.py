"""CPU pinning of oracle/fused_head.py (the checker of the fused ray-march kernels):
* its hand-derived backward - the formulas the HIP kernels evaluate - equals autograd through its
  forward (float64);
* its forward and its coarse sampler equal the product's modular render head, i.e. the restatement
  of the reference's ray_samplers.py / sdf_field.py / decoders.py / renderers.py that the golden
  vectors pin (tests/test_golden_cpu.py runs the same comparison against the reference's numbers).
"""
import os
import sys

import pytest
import torch

from oracle import cpu_backend, fused_head as fh

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

KEYS = ("vol", "origins", "dirs", "starts", "deltas", "MW", "c0", "bc1", "W1", "b1", "A", "b_rgb", "inv_s")


def _small_problem(seed=0):
    g = torch.Generator().manual_seed(seed)
    dt = torch.float64
    rn = lambda *s: torch.randn(*s, generator=g, dtype=dt)
    ru = lambda *s: torch.rand(*s, generator=g, dtype=dt)
    B, Z, Y, X, Fh, F2, H, G, R, S = 2, 5, 7, 6, 8, 8, 12, 8, 6, 10
    p = dict(vol=rn(B, Z, Y, X, Fh + F2), origins=(ru(R, 3) - 0.5) * 0.6,
             dirs=torch.nn.functional.normalize(rn(R, 3), dim=-1),
             starts=torch.sort(ru(R, S) * 0.8, dim=-1).values, deltas=ru(R, S) * 0.05 + 0.01,
             MW=rn(2 * H, Fh) * 0.3, c0=rn(H) * 0.02, bc1=rn(H) * 0.1, W1=rn(1 + G, H) * 0.3,
             b1=rn(1 + G) * 0.1, A=rn(3, 3 + F2 + G + 3) * 0.3, b_rgb=rn(3) * 0.1,
             inv_s=torch.tensor(5.0, dtype=dt))
    p["MW"][:H] *= 0.05
    return p


@pytest.mark.parametrize("seed", [0, 1])
def test_hand_derived_backward_equals_autograd(seed):
    p = _small_problem(seed)
    leaves = [p[k].clone().requires_grad_(k in ("vol",) + KEYS[5:]) for k in KEYS]
    out = fh.field_render(*leaves)
    g = torch.Generator().manual_seed(100 + seed)
    gs, gg = torch.randn(out["sdf"].shape, generator=g, dtype=torch.float64), \
        torch.randn(out["grad"].shape, generator=g, dtype=torch.float64)
    gc_ = torch.randn(out["comp"].shape, generator=g, dtype=torch.float64)
    loss = (out["sdf"] * gs).sum() + (out["grad"] * gg).sum() + (out["comp"] * gc_).sum()
    diff = [leaves[0]] + leaves[5:]
    auto = torch.autograd.grad(loss, diff)
    hand = fh.field_render_backward(*[p[k] for k in KEYS], gs, gg, gc_)
    assert 0.0 < float(out["weights"].min()) and float(out["weights"].max()) < 1.0
    for name, a in zip(("vol", "MW", "c0", "bc1", "W1", "b1", "A", "b_rgb", "inv_s"), auto):
        assert (a - hand[name]).abs().max().item() <= 1e-10 * (1 + a.abs().max().item()), name


def test_softplus_threshold_branch():
    """Softplus(beta=100, threshold=20): past the threshold value = identity, slope 1, curvature 0 -
    the branch the kernels replicate."""
    h = torch.tensor([-0.3, 0.0, 0.1, 0.19999, 0.20001, 0.5], dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.softplus(h, beta=100.0, threshold=20.0)
    (d1,) = torch.autograd.grad(y.sum(), h, create_graph=True)
    (d2,) = torch.autograd.grad(d1.sum(), h)
    assert torch.allclose(fh.softplus(h), y) and torch.allclose(fh.softplus_d1(h), d1)
    assert torch.allclose(fh.softplus_d2(h), d2)


@pytest.fixture
def cpu_kernels(monkeypatch):
    cpu_backend.install(monkeypatch)


def test_oracle_equals_modular_head(cpu_kernels, monkeypatch):
    """Same renderer, same weights, same random draws: outputs, losses and gradients of the fused
    glue (host doubles = oracle/fused_head.py) and of the modular head agree to fp32 round-off."""
    import golden_cases as gc
    from ponderv2_amd import fused_head as fhd

    res = {}
    for fused in (True, False):
        monkeypatch.setattr(fhd, "ENABLED", fused)
        res[fused] = gc.run_neus(torch.device("cpu"), return_values=True)
    a, b = res[True], res[False]
    assert a["fused_calls"] == 1 and b["fused_calls"] == 0
    for k in a["values"]:
        va, vb = a["values"][k].double(), b["values"][k].double()
        assert (va - vb).abs().max().item() <= 2e-4 * (vb.abs().max().item() + 1e-12), k
