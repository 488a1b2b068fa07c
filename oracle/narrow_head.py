"""CPU restatement (checker only) of the fused render head for NARROW SDF decoders -
csrc/raymarch_narrow.hip - in plain torch ops of any float dtype.  Test infrastructure: only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

The head it restates is the one the reference's nuScenes configuration builds
(configs/nuscenes/pretrain-ponder-spunet-v1m1-0-base.py: SDFField with
``sdf_decoder = dict(in_dim=32, out_dim=16 + 1, hidden_size=16, n_blocks=5)``, no colour or
semantic decoder, ``share_volume=True``, depth loss only), from

* ponder/models/ponder/render_utils/decoders.py:6-36   SDFDecoder: x = fc_p(p) * points_factor,
  then per layer ``x = lin_l(x + fc_c[l](feat))`` with Softplus(beta=100) on all but the last;
* fields/sdf_field.py:185-197, 211-284, 122-146         get_sdf, the autograd gradient of the SDF
  with respect to the sample position, NeuS alphas;
* rays.py:83-105, renderers.py:33-45                    compositing weights, expected depth;
* ray_samplers.py:355-463                               the coarse pass + importance sampling
  (shared with oracle/fused_head.py).

Parameters travel as ONE flat vector ``theta`` (the product builds it with torch.cat over the
nn.Linear parameters, so autograd splits its gradient back):

    Wp [H,3] bp [H] | l = 0..L-1: Wc_l [H,C] bc_l [H] | l = 0..L-2: W_l [H,H] b_l [H] | w_last [H] b_last

with L = n_blocks + 1 linear layers; of the last layer only the SDF row enters (the geometry
features of this head feed nothing).  ``field_render_backward`` is the HAND-DERIVED gradient -
reverse mode through the value AND the tangent (d/dp) recursion - exactly as the kernels evaluate
it; tests check it against autograd through ``field_render`` and against the modular head.
"""
import math

import torch

from . import fused_head as fh


def layout(C, H, L):
    """Offsets of the blocks of ``theta``: dict name -> (offset, shape)."""
    off, out = 0, {}

    def put(name, *shape):
        nonlocal off
        n = 1
        for s in shape:
            n *= s
        out[name] = (off, shape)
        off += n

    put("Wp", H, 3)
    put("bp", H)
    for l in range(L):
        put(f"Wc{l}", H, C)
        put(f"bc{l}", H)
    for l in range(L - 1):
        put(f"W{l}", H, H)
        put(f"b{l}", H)
    put("w_last", H)
    put("b_last", 1)
    out["_size"] = (off, ())
    return out


def unpack(theta, C, H, L):
    lay = layout(C, H, L)
    assert theta.numel() == lay["_size"][0]
    return {k: theta[o:o + math.prod(s)].reshape(s) for k, (o, s) in lay.items() if k != "_size"}


def mlp(P, L, pf, p, f, J=None, keep=False):
    """SDF of N samples.  p (N,3), f (N,C); with J (N,C,3) = d f / d p also the gradient d sdf / d p.
    Returns sdf (N,), grad (N,3) | None, saved."""
    x = (p @ P["Wp"].t() + P["bp"]) * pf                       # (N,H)
    X = None
    if J is not None:
        X = (P["Wp"] * pf)[None].expand(p.shape[0], -1, -1)     # (N,H,3)
    saved = dict(u=[], U=[], z=[], Z=[])
    for l in range(L - 1):
        u = x + f @ P[f"Wc{l}"].t() + P[f"bc{l}"]
        z = u @ P[f"W{l}"].t() + P[f"b{l}"]
        x = fh.softplus(z)
        if J is not None:
            U = X + torch.einsum("hc,nca->nha", P[f"Wc{l}"], J)
            Z = torch.einsum("gh,nha->nga", P[f"W{l}"], U)
            X = fh.softplus_d1(z)[..., None] * Z
        if keep:
            saved["u"].append(u)
            saved["z"].append(z)
            if J is not None:
                saved["U"].append(U)
                saved["Z"].append(Z)
    l = L - 1
    u = x + f @ P[f"Wc{l}"].t() + P[f"bc{l}"]
    sdf = u @ P["w_last"] + P["b_last"][0]
    grad = None
    if J is not None:
        U = X + torch.einsum("hc,nca->nha", P[f"Wc{l}"], J)
        grad = torch.einsum("h,nha->na", P["w_last"], U)
        if keep:
            saved["U"].append(U)
    if keep:
        saved["u"].append(u)
    return sdf, grad, saved


def sdf_fn(theta, C, H, L, pf):
    """The coarse pass's SDF (value only) in the signature oracle/fused_head.coarse_sample takes."""
    P = unpack(theta, C, H, L)

    def fn(vol_rows, p, scene, vol_shape):
        rows, _, w, _ = fh.corner_model(p, scene, vol_shape)
        f = (vol_rows[rows] * w[..., None]).sum(1)
        return mlp(P, L, pf, p, f)[0]

    return fn


def coarse_sample(vol, origins, dirs, nears, fars, lin_bins, t_rand, u_rand, n_importance, theta,
                  H, L, pf, base_inv_s=64.0, return_debug=False):
    C = vol.shape[-1]
    return fh.coarse_sample(vol, origins, dirs, nears, fars, lin_bins, t_rand, u_rand, n_importance,
                            None, None, None, None, None, base_inv_s, return_debug,
                            sdf_fn=sdf_fn(theta, C, H, L, pf))


def field_render(vol, origins, dirs, starts, deltas, theta, inv_s, H, L, pf, keep=False):
    """Differentiable (autograd) statement of the fused main pass.

    vol (B,Z,Y,X,C); starts/deltas (R,S) constants; theta flat parameters; inv_s () tensor.
    Returns dict: sdf (R,S), grad (R,S,3), weights (R,S), comp (R,2) = [sum_k w_k t_k, sum_k w_k]."""
    B, Z, Y, X, C = vol.shape
    R, S = starts.shape
    P = unpack(theta, C, H, L)
    scene = (torch.arange(R) // (R // B)).repeat_interleave(S)
    p = (origins[:, None, :] + dirs[:, None, :] * starts[..., None]).reshape(-1, 3)
    rows, inb, w8, dw8 = fh.corner_model(p, scene, (B, Z, Y, X))
    vr = vol.reshape(-1, C)[rows]                               # (N,8,C)
    f = (vr * w8[..., None]).sum(1)
    J = torch.einsum("nkc,nka->nca", vr, dw8)                   # (N,C,3)
    sdf, g, saved = mlp(P, L, pf, p, f, J, keep=keep)
    d = dirs[:, None, :].expand(R, S, 3).reshape(-1, 3)
    c = (g * d).sum(-1)
    half = -torch.relu(-c) * deltas.reshape(-1) * 0.5
    e1 = torch.sigmoid((sdf - half) * inv_s)
    e2 = torch.sigmoid((sdf + half) * inv_s)
    alpha = ((e1 - e2 + 1e-5) / (e1 + 1e-5)).clip(0.0, 1.0).reshape(R, S)
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-7], 1), 1)[:, :-1]
    w = alpha * T
    comp = torch.stack([(w * starts).sum(1), w.sum(1)], 1)
    out = dict(sdf=sdf.reshape(R, S), grad=g.reshape(R, S, 3), weights=w, comp=comp)
    if keep:
        out["_saved"] = dict(p=p, rows=rows, w8=w8, dw8=dw8, f=f, J=J, d=d, c=c, half=half, e1=e1,
                             e2=e2, alpha=alpha, T=T, sdf=sdf, mlp=saved, vr=vr)
    return out


def field_render_backward(vol, origins, dirs, starts, deltas, theta, inv_s, H, L, pf, g_sdf, g_grad,
                          g_weights, g_comp, debug=False):
    """Hand-derived gradients of ``field_render`` given upstream g_sdf (R,S), g_grad (R,S,3),
    g_weights (R,S), g_comp (R,2).  Returns dict(vol, theta, inv_s)."""
    with torch.no_grad():
        out = field_render(vol, origins, dirs, starts, deltas, theta, inv_s, H, L, pf, keep=True)
        s = out["_saved"]
        B, Z, Y, X, C = vol.shape
        R, S = starts.shape
        P = unpack(theta, C, H, L)
        lay = layout(C, H, L)
        w, alpha, T = out["weights"], s["alpha"], s["T"]
        sdf, d = s["sdf"], s["d"]
        # ray level: d w_k -> d alpha_k through the transmittance product
        gw = g_comp[:, 0:1] * starts + g_comp[:, 1:2] + g_weights
        qk = gw * w
        after = torch.flip(torch.cumsum(torch.flip(qk, [1]), 1), [1]) - qk
        g_alpha = (gw * T - after / (1.0 - alpha + 1e-7)).reshape(-1)
        e1, e2 = s["e1"], s["e2"]
        raw = (e1 - e2 + 1e-5) / (e1 + 1e-5)
        g_raw = torch.where((raw >= 0) & (raw <= 1), g_alpha, torch.zeros_like(g_alpha))
        g_e1 = g_raw * e2 / (e1 + 1e-5) ** 2
        g_e2 = -g_raw / (e1 + 1e-5)
        gu1 = g_e1 * e1 * (1 - e1)
        gu2 = g_e2 * e2 * (1 - e2)
        a = g_sdf.reshape(-1) + inv_s * (gu1 + gu2)                       # d L / d sdf_n
        g_half = inv_s * (gu2 - gu1)
        g_invs = (gu1 * (sdf - s["half"]) + gu2 * (sdf + s["half"])).sum()
        g_c = torch.where(s["c"] < 0, g_half * deltas.reshape(-1) * 0.5, torch.zeros_like(g_half))
        gam = g_grad.reshape(-1, 3) + g_c[:, None] * d                     # d L / d grad_n
        # reverse mode through the value / tangent recursion
        m = s["mlp"]
        gt = torch.zeros_like(theta)

        def acc(name, val):
            o, shape = lay[name]
            gt[o:o + val.numel()] += val.reshape(-1)

        l = L - 1
        acc("w_last", a @ m["u"][l] + torch.einsum("nha,na->h", m["U"][l], gam))
        acc("b_last", a.sum().reshape(1))
        ub = a[:, None] * P["w_last"][None]                                 # (N,H)   d/d u_l
        Ub = P["w_last"][None, :, None] * gam[:, None, :]                   # (N,H,3) d/d U_l
        gf = torch.zeros_like(s["f"])
        gJ = torch.zeros_like(s["J"])
        for l in range(L - 1, -1, -1):
            Wc = P[f"Wc{l}"]
            acc(f"Wc{l}", ub.t() @ s["f"] + torch.einsum("nha,nca->hc", Ub, s["J"]))
            acc(f"bc{l}", ub.sum(0))
            gf += ub @ Wc
            gJ += torch.einsum("hc,nha->nca", Wc, Ub)
            if l == 0:
                break
            # x_l = softplus(z_{l-1}), X_l = s(z_{l-1}) * Z_{l-1};  u_l = x_l + ...
            z, Zt = m["z"][l - 1], m["Z"][l - 1]
            s1, s2 = fh.softplus_d1(z), fh.softplus_d2(z)
            zb = ub * s1 + (Ub * Zt).sum(-1) * s2
            Zb = s1[..., None] * Ub
            Wl = P[f"W{l - 1}"]
            acc(f"W{l - 1}", zb.t() @ m["u"][l - 1] + torch.einsum("nga,nha->gh", Zb, m["U"][l - 1]))
            acc(f"b{l - 1}", zb.sum(0))
            ub = zb @ Wl
            Ub = torch.einsum("gh,nga->nha", Wl, Zb)
        acc("Wp", pf * (ub.t() @ s["p"] + Ub.sum(0)))
        acc("bp", pf * ub.sum(0))
        # volume: f = sum_c w_c V_c, J[:, :, a] = sum_c d_a w_c V_c
        contrib = s["w8"][..., None] * gf[:, None, :] + torch.einsum("nka,nca->nkc", s["dw8"], gJ)
        g_vol = torch.zeros_like(vol.reshape(-1, C))
        g_vol.index_add_(0, s["rows"].reshape(-1), contrib.reshape(-1, C))
        res = dict(vol=g_vol.reshape(vol.shape), theta=gt, inv_s=g_invs)
        if debug:
            res["_dbg"] = dict(a=a, gam=gam, gf=gf, gJ=gJ, g_alpha=g_alpha)
        return res
