"""CPU pinning of oracle/narrow_head.py (the checker of csrc/raymarch_narrow.hip):
* its hand-derived backward - reverse mode through the value and tangent recursion, the formulas the
  HIP kernels evaluate - equals autograd through its forward (float64), for every upstream gradient
  the operation takes (sdf, grad sdf, weights, depth sums);
* the product's glue on host doubles built from it equals the modular render head on the
  nuScenes-shaped model: loss and gradients, i.e. the restatement of the reference's ray_samplers.py /
  sdf_field.py / decoders.py that tests/test_golden_cpu.py pins to the reference's own numbers."""
import pytest
import torch

from oracle import cpu_backend, narrow_head as nh

ARGS = ("vol", "origins", "dirs", "starts", "deltas", "theta", "inv_s")


def _small_problem(seed=0):
    g = torch.Generator().manual_seed(seed)
    dt = torch.float64
    rn = lambda *s: torch.randn(*s, generator=g, dtype=dt)
    ru = lambda *s: torch.rand(*s, generator=g, dtype=dt)
    B, Z, Y, X, C, H, L, R, S = 2, 3, 7, 6, 8, 4, 4, 6, 10
    theta = rn(nh.layout(C, H, L)["_size"][0]) * 0.3
    p = dict(vol=rn(B, Z, Y, X, C), origins=ru(R, 3) * 0.5 + 0.1,
             dirs=torch.nn.functional.normalize(rn(R, 3), dim=-1),
             starts=torch.sort(ru(R, S) * 0.5, dim=-1).values, deltas=ru(R, S) * 0.05 + 0.01,
             theta=theta, inv_s=torch.tensor(5.0, dtype=dt))
    return p, (H, L, 0.7)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_hand_derived_backward_equals_autograd(seed):
    p, (H, L, pf) = _small_problem(seed)
    leaves = {k: v.clone().requires_grad_(k in ("vol", "theta", "inv_s")) for k, v in p.items()}
    out = nh.field_render(*[leaves[k] for k in ARGS], H, L, pf)
    g = torch.Generator().manual_seed(100 + seed)
    ups = [torch.randn(out[k].shape, generator=g, dtype=torch.float64) for k in ("sdf", "grad", "weights", "comp")]
    loss = sum((out[k] * u).sum() for k, u in zip(("sdf", "grad", "weights", "comp"), ups))
    auto = torch.autograd.grad(loss, [leaves["vol"], leaves["theta"], leaves["inv_s"]])
    hand = nh.field_render_backward(*[p[k] for k in ARGS], H, L, pf, *ups)
    assert 0.0 < float(out["weights"].min()) and float(out["weights"].max()) < 1.0
    for name, a in zip(("vol", "theta", "inv_s"), auto):
        assert (a - hand[name]).abs().max().item() <= 1e-10 * (1 + a.abs().max().item()), name


def test_theta_layout_matches_the_decoder():
    """pack_theta (product) and unpack (oracle) agree on the layout: the packed decoder evaluates to
    the module's own output."""
    from ponderv2_amd import narrow_head as prod
    from ponderv2_amd.ponder.models.ponder.render_utils.decoders import SDFDecoder

    torch.manual_seed(0)
    sd = SDFDecoder(in_dim=prod.C, out_dim=prod.H + 1, hidden_size=prod.H, n_blocks=prod.L - 1).double()
    theta = prod.pack_theta(sd).double()
    assert theta.numel() == prod.NTHETA
    theta = torch.cat([p.reshape(-1) for p in (sd.fc_p.weight, sd.fc_p.bias)]
                      + [q.reshape(-1) for l in range(prod.L) for q in (sd.fc_c[l].weight, sd.fc_c[l].bias)]
                      + [q.reshape(-1) for l in range(prod.L - 1)
                         for q in (getattr(sd, f"lin{l}").weight, getattr(sd, f"lin{l}").bias)]
                      + [sd.last_linear.weight[0], sd.last_linear.bias[0:1]]).detach()
    assert torch.equal(theta.float(), prod.pack_theta(sd).detach())
    pts, feat = torch.rand(9, 3, dtype=torch.float64), torch.randn(9, prod.C, dtype=torch.float64)
    want = sd(pts, feat)[:, 0]
    got = nh.mlp(nh.unpack(theta, prod.C, prod.H, prod.L), prod.L, sd.points_factor, pts, feat)[0]
    assert (want - got).abs().max().item() < 1e-12


@pytest.fixture
def cpu_kernels(monkeypatch):
    cpu_backend.install(monkeypatch)


def test_oracle_equals_modular_head(cpu_kernels, monkeypatch):
    """Same model, same weights, same random draws: loss and gradients of the fused glue (host doubles
    = oracle/narrow_head.py) and of the modular head agree to fp32 round-off."""
    import golden_cases as gc
    from ponderv2_amd import narrow_head as prod

    res = {}
    for fused in (True, False):
        monkeypatch.setattr(prod, "ENABLED", fused)
        monkeypatch.setattr(prod, "CALLS", 0)
        res[fused] = gc.run_ponder_outdoor(torch.device("cpu"))
        assert prod.CALLS == (1 if fused else 0)
    a, b = res[True], res[False]
    for k in a:
        if isinstance(a[k], dict):
            continue
        assert abs(float(a[k]) - float(b[k])) <= 2e-4, (k, a[k], b[k])
