"""Stage-by-stage comparison of the fused ray-march kernels (csrc/raymarch_fused.hip) with the
fp64 restatement in oracle/fused_head.py on one seeded problem.  Prints one line per intermediate
(max abs error / max abs reference), so that a single GPU run localises a defect.  Used by
tests/test_gpu_fused_head.py (asserting) and by hand (`python tools/check_fused_head.py`).
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_problem(seed=0, B=2, Z=8, Y=16, X=16, R=12, S=132, S0=96, n_imp=36, dtype=torch.float64):
    from ponderv2_amd import fused_head as fhd

    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g, dtype=dtype)
    ru = lambda *s: torch.rand(*s, generator=g, dtype=dtype)
    H, FS, F2, G = fhd.H, fhd.FS, fhd.F2, fhd.G
    C = FS + F2
    p = dict(vol=rn(B, Z, Y, X, C) * 0.6)
    p["origins"] = (ru(R, 3) - 0.5) * 0.5
    p["dirs"] = torch.nn.functional.normalize(rn(R, 3), dim=-1)
    p["nears"] = ru(R) * 0.05 + 0.01
    p["fars"] = p["nears"] + 0.4 + ru(R) * 0.6
    p["MW"] = rn(2 * H, FS) * 0.08
    p["MW"][:H] *= 0.15                      # h0 ~ 1e-2: the curved part of softplus(beta=100)
    p["c0"] = rn(H) * 0.01
    p["bc1"] = rn(H) * 0.1
    p["W1"] = rn(1 + G, H) * 0.1
    p["b1"] = rn(1 + G) * 0.1
    p["A"] = rn(3, 3 + F2 + G + 3) * 0.2
    p["b_rgb"] = rn(3) * 0.1
    p["inv_s"] = torch.tensor(20.0, dtype=dtype)
    p["t_rand"] = ru(R, S0 + 1)
    p["u_rand"] = ru(R, n_imp + 1)
    p["lin_bins"] = torch.linspace(0.0, 1.0, S0 + 1).to(dtype)
    nb = n_imp + 1
    p["lin_u"] = torch.linspace(0.0, 1.0 - 1.0 / nb, nb).to(dtype)
    p["n_imp"] = n_imp
    st = torch.sort(ru(R, S + 1), dim=-1).values
    e = st * p["fars"][:, None] + (1 - st) * p["nears"][:, None]
    p["starts"], p["deltas"] = e[:, :-1].contiguous(), (e[:, 1:] - e[:, :-1]).contiguous()
    p["g_sdf"] = rn(R, S)
    p["g_grad"] = rn(R, S, 3)
    p["g_comp"] = rn(R, F2 + G + 12)
    return p


def _cmp(rows, name, got, ref):
    got = got.detach().double().cpu().reshape(-1)
    ref = ref.detach().double().cpu().reshape(-1)
    err = (got - ref).abs().max().item() if ref.numel() else 0.0
    rows.append((name, err, ref.abs().max().item() if ref.numel() else 0.0))


def run(device, seed=0, verbose=True, **shape):
    from oracle import fused_head as fh
    from ponderv2_amd import _lib, fused_head as fhd
    from ponderv2_amd.kernels import _ptr, _stream

    p = make_problem(seed, **shape)
    dev = lambda t: t.to(device=device, dtype=torch.float32).contiguous()
    d = {k: (dev(v) if torch.is_tensor(v) else v) for k, v in p.items()}
    rows = []
    # ---- coarse pass
    ref_bins, dbg = fh.coarse_sample(p["vol"], p["origins"], p["dirs"], p["nears"], p["fars"],
                                     p["lin_bins"], p["t_rand"], p["u_rand"], p["n_imp"], p["MW"], p["c0"],
                                     p["bc1"], p["W1"][0], p["b1"][0], 64.0, return_debug=True)
    bins, starts, deltas, gd = fhd.coarse_sample(
        d["vol"], d["origins"], d["dirs"], d["nears"], d["fars"], d["lin_bins"], d["t_rand"], d["lin_u"],
        d["u_rand"], p["n_imp"], d["MW"], d["c0"], d["bc1"], d["W1"], d["b1"], 64.0, debug=True)
    _cmp(rows, "coarse.sdf", gd["sdf"], dbg["sdf"])
    _cmp(rows, "coarse.weights", gd["weights"], dbg["weights"])
    flips = (gd["idx"].cpu().long() != dbg["idx"]).sum().item()
    rows.append(("coarse.idx flips (count)", float(flips), float(dbg["idx"].numel())))
    _cmp(rows, "coarse.bins", bins, ref_bins)
    rs, rd = fh.bins_to_samples(ref_bins, p["nears"], p["fars"])
    _cmp(rows, "coarse.starts", starts, rs)
    _cmp(rows, "coarse.deltas", deltas, rd)
    # ---- main pass forward (on the problem's own sorted samples, independent of the coarse result)
    args64 = [p[k] for k in ("vol", "origins", "dirs", "starts", "deltas", "MW", "c0", "bc1", "W1", "b1",
                             "A", "b_rgb", "inv_s")]
    ref = fh.field_render(*args64, norm_pts=True, norm_padding=0.1, keep=True)
    s = ref["_saved"]
    args32 = [d[k].clone().requires_grad_(k in ("vol", "MW", "c0", "bc1", "W1", "b1", "A", "b_rgb", "inv_s"))
              for k in ("vol", "origins", "dirs", "starts", "deltas", "MW", "c0", "bc1", "W1", "b1", "A",
                        "b_rgb", "inv_s")]
    sdf, grad, weights, comp = fhd.field_render(*args32, True, 1.0 + 0.1 + 10e-4)
    node = sdf.grad_fn
    saved = dict(zip(("vol5", "origins", "dirs", "starts", "deltas", "MW", "W1", "A", "inv_s", "sdf",
                      "alpha", "vals", "sf", "sh0", "sa1", "sq", "weights", "Mt"), node.saved_tensors))
    R, S = p["starts"].shape
    N = R * S
    vals = saved["vals"].reshape(N, -1)
    _cmp(rows, "fwd.f", saved["sf"], s["f"])
    _cmp(rows, "fwd.f'", vals[:, 0:64], s["f2"])
    _cmp(rows, "fwd.h0", saved["sh0"], s["h0"])
    _cmp(rows, "fwd.a1", saved["sa1"], s["a1"])
    _cmp(rows, "fwd.sdf", sdf, ref["sdf"])
    _cmp(rows, "fwd.geo", vals[:, 64:128], s["geo"])
    _cmp(rows, "fwd.q", saved["sq"], s["q"])
    _cmp(rows, "fwd.grad", grad, ref["grad"])
    _cmp(rows, "fwd.normal", vals[:, 131:134], torch.nn.functional.normalize(s["g"], dim=-1))
    _cmp(rows, "fwd.rgb", vals[:, 134:137], s["rgb"])
    _cmp(rows, "fwd.t,1,0", vals[:, 137:140],
         torch.stack([p["starts"].reshape(-1), torch.ones(N, dtype=torch.float64),
                      torch.zeros(N, dtype=torch.float64)], 1))
    _cmp(rows, "fwd.alpha", saved["alpha"], s["alpha"])
    _cmp(rows, "fwd.weights", weights, ref["weights"])
    _cmp(rows, "fwd.comp", comp, ref["comp"])
    # ---- backward
    loss = (sdf * d["g_sdf"]).sum() + (grad * d["g_grad"]).sum() + (comp * d["g_comp"]).sum()
    loss.backward()
    href = fh.field_render_backward(*args64, p["g_sdf"], p["g_grad"], p["g_comp"], norm_pts=True,
                                    norm_padding=0.1, debug=True)
    for name, t in zip(("vol", "MW", "c0", "bc1", "W1", "b1", "A", "b_rgb", "inv_s"),
                       [args32[0]] + args32[5:]):
        _cmp(rows, "bwd.d" + name, t.grad, href[name])
    # the backward kernel's own intermediates, by a direct call
    L = _lib.lib()
    db = href["_dbg"]
    new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=device)
    g_alpha = dev(db["g_alpha"].reshape(R, S))
    C = p["vol"].shape[-1]
    gfeat, gvec, gz, tmat = new(N, C), new(N, 4), new(N, 2 * fhd.H), new(N, fhd.H)
    gq, gh, gy, sums = new(N, fhd.FS), new(N, 68), new(N, 4), new(fhd.NSUM)
    MWd, W1d = d["MW"], d["W1"]
    # (named, so the transposed copies stay alive until the launch has been enqueued)
    Mt, W1gt, Wc1t = (MWd[:fhd.H].t().contiguous(), W1d[1:].t().contiguous(),
                      MWd[fhd.H:].t().contiguous())
    B, Z, Y, X, _ = p["vol"].shape
    _lib.check(L.pv2_neus_field_backward(
        _ptr(d["vol"]), B, Z, Y, X, C, _ptr(d["origins"]), _ptr(d["dirs"]), _ptr(d["starts"]),
        _ptr(d["deltas"]), R, S, _ptr(MWd), _ptr(W1d), _ptr(Mt), _ptr(W1gt), _ptr(Wc1t), _ptr(d["A"]),
        _ptr(d["inv_s"].reshape(1)), 1, 1.0 + 0.1 + 10e-4, _ptr(saved["sdf"]), _ptr(saved["vals"]),
        _ptr(saved["sh0"]), _ptr(saved["sq"]), _ptr(saved["weights"]), _ptr(g_alpha), _ptr(d["g_sdf"]),
        _ptr(d["g_grad"]), _ptr(d["g_comp"]), _ptr(gfeat), _ptr(gvec), _ptr(gz), _ptr(tmat), _ptr(gq),
        _ptr(gh), _ptr(gy), _ptr(sums), None, _stream(d["vol"])), "pv2_neus_field_backward")
    _cmp(rows, "bwdk.gvec", gvec[:, :3], db["gvec"])
    _cmp(rows, "bwdk.gy", gy[:, :3], db["gy"])
    _cmp(rows, "bwdk.gh", gh[:, :65], db["gh"])
    _cmp(rows, "bwdk.gq", gq, db["gq"])
    _cmp(rows, "bwdk.gh0", gz[:, :fhd.H], db["gh0"])
    _cmp(rows, "bwdk.ga1", gz[:, fhd.H:], db["ga1"])
    _cmp(rows, "bwdk.tmat", tmat, db["tmat"])
    _cmp(rows, "bwdk.gfeat.f", gfeat[:, :64], db["gfeat"][:, :64])
    _cmp(rows, "bwdk.gfeat.f'", gfeat[:, 64:], db["gfeat"][:, 64:])
    _cmp(rows, "bwdk.sum.c0", sums[fhd.SUM_C0:fhd.SUM_C0 + fhd.H], href["c0"])
    _cmp(rows, "bwdk.sum.bc1", sums[fhd.SUM_BC1:fhd.SUM_BC1 + fhd.H], href["bc1"])
    _cmp(rows, "bwdk.sum.b1", sums[fhd.SUM_B1:fhd.SUM_B1 + 65], href["b1"])
    _cmp(rows, "bwdk.sum.qsum", sums[fhd.SUM_Q:fhd.SUM_Q + 64], db["gq"].sum(0))
    _cmp(rows, "bwdk.sum.brgb", sums[fhd.SUM_RGB:fhd.SUM_RGB + 3], href["b_rgb"])
    _cmp(rows, "bwdk.sum.inv_s", sums[fhd.SUM_INVS:fhd.SUM_INVS + 1], href["inv_s"])
    if verbose:
        for name, err, mag in rows:
            print("%-28s err %.3e   ref max %.3e   rel %.2e" % (name, err, mag, err / (mag + 1e-30)))
    return rows


if __name__ == "__main__":
    assert torch.cuda.is_available()
    run(torch.device("cuda:0"))
    print("--- odd sizes: one scene, sample count not a multiple of 32")
    run(torch.device("cuda:0"), seed=1, B=1, R=5, S=45, S0=40, n_imp=7, Z=5, Y=9, X=11)
