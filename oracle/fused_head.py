"""CPU restatement (checker only) of the two fused render-head operations of csrc/raymarch_fused.hip,
in plain torch ops of any float dtype.  Test infrastructure: only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this file; the product never does.

What is restated, and from where in the reference (paths relative to the reference checkout):

* ``coarse_sample``  - NeuSSampler.generate_ray_samples with one up-sampling step
  (ponder/models/ponder/render_utils/ray_samplers.py:355-424): stratified uniform bins
  (:55-107), the no-grad SDF of the UN-normalised start positions (neus.py:17-21,
  sdf_field.py:185-197), fixed-inv_s section alphas (:426-463), weights
  (rays.py:83-105), inverse-CDF importance samples (:227-322) and the sorted merge
  (rays.py:118-153).
* ``field_render``   - SDFField.forward with return_alphas (fields/sdf_field.py:211-284, 122-146),
  decoders.py:6-109 for the shipped head shape (SDF MLP with ONE hidden block, colour / semantic
  heads with none, points_factor = 0), the compositing weights (rays.py:83-105) and the
  renderers' weighted sums (renderers.py:5-75).

Both are written on COLLAPSED parameters (the layers that have no activation between them are
multiplied together by the caller, in torch, so autograd carries the gradients back to the
nn.Linear parameters):

    MW  = [W0 Wc0 ; Wc1]  (2H x F)   c0 = W0 bc0 + b0   bc1          W1 (1+G x H), b1
    A   = Wr1 Wrc (3 x (3+F'+G+3))   b_rgb = Wr1 brc + br1

``field_render_backward`` is the HAND-DERIVED gradient (first order in the upstream gradients,
including every second-order term that enters through grad sdf) exactly as the HIP kernels
evaluate it; tests check it against autograd through ``field_render`` and through the modular
head.
"""
import torch
import torch.nn.functional as F

SOFTPLUS_BETA = 100.0
SOFTPLUS_THRESHOLD = 20.0


# ------------------------------------------------------------------------------------------
# trilinear corner model: zeros padding, align_corners=True, no smoothstep (sdf_field.py:164-166)
# ------------------------------------------------------------------------------------------
def corner_model(p, scene, vol_shape):
    """p (N,3) in grid-normalised [0,1] coordinates (x,y,z -> X,Y,Z axes), scene (N,) long.
    Returns rows (N,8) long flat row index into the (B*Z*Y*X, C) volume (0 where out of bounds),
    inb (N,8) bool, w (N,8) corner weights, dw (N,8,3) d w / d p."""
    B, Z, Y, X = vol_shape
    sizes = torch.tensor([X, Y, Z], dtype=p.dtype)
    x = p * (sizes - 1)                     # ((2p-1)+1)/2*(size-1)
    fl = torch.floor(x)
    t = x - fl
    i0 = fl.long()
    rows, inb, w, dw = [], [], [], []
    for c in range(8):
        bits = [(c >> 0) & 1, (c >> 1) & 1, (c >> 2) & 1]     # x, y, z
        idx = [i0[:, a] + bits[a] for a in range(3)]
        ok = torch.ones(p.shape[0], dtype=torch.bool)
        for a, size in enumerate((X, Y, Z)):
            ok &= (idx[a] >= 0) & (idx[a] < size)
        om = [t[:, a] if bits[a] else 1 - t[:, a] for a in range(3)]
        dom = [(1.0 if bits[a] else -1.0) * (sizes[a] - 1) for a in range(3)]
        wc = om[0] * om[1] * om[2]
        dwc = torch.stack([dom[0] * om[1] * om[2], om[0] * dom[1] * om[2], om[0] * om[1] * dom[2]], -1)
        row = ((scene * Z + idx[2].clamp(0, Z - 1)) * Y + idx[1].clamp(0, Y - 1)) * X + idx[0].clamp(0, X - 1)
        rows.append(torch.where(ok, row, torch.zeros_like(row)))
        inb.append(ok)
        w.append(wc * ok)
        dw.append(dwc * ok[:, None])
    return torch.stack(rows, 1), torch.stack(inb, 1), torch.stack(w, 1), torch.stack(dw, 1)


def normalize_points(p, padding=0.1):
    """sdf_field.py:58-74."""
    q = p / (1 + padding + 10e-4) + 0.5
    q = torch.where(q >= 1, torch.full_like(q, 1 - 10e-4), q)
    return torch.where(q < 0, torch.zeros_like(q), q)


def softplus(h):
    return F.softplus(h, beta=SOFTPLUS_BETA, threshold=SOFTPLUS_THRESHOLD)


def softplus_d1(h):
    """d softplus / dh as torch evaluates it (1 past the threshold)."""
    return torch.where(h * SOFTPLUS_BETA > SOFTPLUS_THRESHOLD, torch.ones_like(h),
                       torch.sigmoid(h * SOFTPLUS_BETA))


def softplus_d2(h):
    s = torch.sigmoid(h * SOFTPLUS_BETA)
    return torch.where(h * SOFTPLUS_BETA > SOFTPLUS_THRESHOLD, torch.zeros_like(h),
                       SOFTPLUS_BETA * s * (1 - s))


# ------------------------------------------------------------------------------------------
# coarse pass + importance sampling
# ------------------------------------------------------------------------------------------
def sdf_only(vol_rows, p, scene, vol_shape, MW, c0, bc1, v1, b1_0):
    """SDF value at grid-normalised points from the first F channels of the volume."""
    H = c0.shape[0]
    Fh = MW.shape[1]
    rows, _, w, _ = corner_model(p, scene, vol_shape)
    f = (vol_rows[rows][..., :Fh] * w[..., None]).sum(1)
    z = f @ MW.t()
    a1 = softplus(z[:, :H] + c0) + z[:, H:] + bc1
    return a1 @ v1 + b1_0


def coarse_sample(vol, origins, dirs, nears, fars, lin_bins, t_rand, u_rand, n_importance,
                  MW, c0, bc1, v1, b1_0, base_inv_s=64.0, return_debug=False, sdf_fn=None):
    """vol (B,Z,Y,X,C); origins/dirs (R,3); nears/fars (R,); lin_bins (S0+1,) = linspace(0,1);
    t_rand (R,S0+1) | (R,1) | None; u_rand (R,n_importance+1) | (R,1) | None.
    Returns bins (R, S0+n_importance+1): the sorted spacing edges of the merged samples."""
    B, Z, Y, X, C = vol.shape
    R = origins.shape[0]
    S0 = lin_bins.shape[0] - 1
    scene = torch.arange(R) // (R // B)
    bins = lin_bins.expand(R, -1)
    if t_rand is not None:
        centers = (bins[:, 1:] + bins[:, :-1]) / 2.0
        upper = torch.cat([centers, bins[:, -1:]], -1)
        lower = torch.cat([bins[:, :1], centers], -1)
        bins = lower + (upper - lower) * t_rand
    near, far = nears[:, None], fars[:, None]
    e = bins * far + (1 - bins) * near
    starts, ends = e[:, :-1], e[:, 1:]
    pts = origins[:, None, :] + dirs[:, None, :] * starts[..., None]          # NOT normalised (Q1)
    if sdf_fn is not None:   # another head's SDF (oracle/narrow_head.py), same sampling logic
        sdf = sdf_fn(vol.reshape(-1, C), pts.reshape(-1, 3), scene.repeat_interleave(S0),
                     (B, Z, Y, X)).reshape(R, S0)
    else:
        sdf = sdf_only(vol.reshape(-1, C), pts.reshape(-1, 3), scene.repeat_interleave(S0),
                       (B, Z, Y, X), MW, c0, bc1, v1, b1_0).reshape(R, S0)
    prev_sdf, next_sdf = sdf[:, :-1], sdf[:, 1:]
    dist = (ends - starts)[:, :-1]
    mid = (prev_sdf + next_sdf) * 0.5
    cos = (next_sdf - prev_sdf) / (dist + 1e-5)
    prev_cos = torch.cat([torch.zeros_like(cos[:, :1]), cos[:, :-1]], -1)
    cos = torch.minimum(prev_cos, cos).clip(-1e3, 0.0)
    prev_cdf = torch.sigmoid((mid - cos * dist * 0.5) * base_inv_s)
    next_cdf = torch.sigmoid((mid + cos * dist * 0.5) * base_inv_s)
    alpha = (prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5)
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-7], 1), 1)
    w = torch.cat([alpha * T[:, :-1], torch.zeros_like(alpha[:, :1])], 1)    # (R,S0)
    # PDFSampler
    nb = n_importance + 1
    w_sum = w.sum(-1, keepdim=True)
    pad = torch.relu(1e-5 - w_sum)
    w2 = w + pad / S0
    pdf = w2 / (w_sum + pad)
    cdf = torch.min(torch.ones_like(pdf), torch.cumsum(pdf, -1))
    cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], -1)                  # (R,S0+1)
    u = torch.linspace(0.0, 1.0 - 1.0 / nb, steps=nb).to(vol.dtype)
    if u_rand is not None:
        u = u.expand(R, nb) + u_rand / nb
    else:
        u = (u + 1.0 / (2 * nb)).expand(R, nb)
    u = u.contiguous()
    idx = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(idx - 1, 0, S0)
    above = torch.clamp(idx, 0, S0)
    cdf0, cdf1 = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    e0, e1 = torch.gather(bins, -1, below), torch.gather(bins, -1, above)
    denom = cdf1 - cdf0
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    tt = torch.clip((u - cdf0) / denom, 0, 1)
    new_bins = e0 + tt * (e1 - e0)                                            # (R,nb)
    merged = torch.sort(torch.cat([bins[:, :-1], new_bins[:, :-1]], -1), -1).values
    last = torch.maximum(bins[:, -1:], new_bins[:, -1:])
    out = torch.cat([merged, last], -1)
    if return_debug:
        return out, dict(sdf=sdf, weights=w, idx=idx, cdf=cdf, u=u, new_bins=new_bins, bins=bins)
    return out


def bins_to_samples(bins, nears, fars):
    """spacing edges (R,S+1) -> starts (R,S), deltas (R,S) in ray distance."""
    e = bins * fars[:, None] + (1 - bins) * nears[:, None]
    return e[:, :-1], e[:, 1:] - e[:, :-1]


# ------------------------------------------------------------------------------------------
# main pass
# ------------------------------------------------------------------------------------------
def field_render(vol, origins, dirs, starts, deltas, MW, c0, bc1, W1, b1, A, b_rgb, inv_s,
                 norm_pts=True, norm_padding=0.1, keep=False):
    """Differentiable (autograd) statement of the fused main pass.

    vol (B,Z,Y,X,C) with C = F + F'; starts/deltas (R,S) (constants); MW (2H,F); W1 (1+G,H);
    A (3, 3+F'+G+3); inv_s () tensor.  Returns dict: sdf (R,S), grad (R,S,3), weights (R,S),
    comp (R, F'+G+12) = sum_k w_k [f', geo, grad, normal, rgb, t, 1, 0] (the kernels' value row)."""
    B, Z, Y, X, C = vol.shape
    R, S = starts.shape
    H = c0.shape[0]
    Fh = MW.shape[1]
    scene = (torch.arange(R) // (R // B)).repeat_interleave(S)
    p = (origins[:, None, :] + dirs[:, None, :] * starts[..., None]).reshape(-1, 3)
    if norm_pts:
        p = normalize_points(p, norm_padding)
    rows, inb, w8, dw8 = corner_model(p, scene, (B, Z, Y, X))
    vr = vol.reshape(-1, C)[rows]                                   # (N,8,C)
    feat = (vr * w8[..., None]).sum(1)                              # (N,C)
    f, f2 = feat[:, :Fh], feat[:, Fh:]
    J = torch.einsum("nkc,nka->nca", vr[..., :Fh], dw8)             # (N,F,3) d f / d p
    z = f @ MW.t()
    h0 = z[:, :H] + c0
    sig0 = softplus_d1(h0)
    a1 = softplus(h0) + z[:, H:] + bc1
    h = a1 @ W1.t() + b1
    sdf, geo = h[:, 0], h[:, 1:]
    v1 = W1[0]
    q = (v1 * sig0) @ MW[:H] + v1 @ MW[H:]                          # (N,F) d sdf / d f
    g = torch.einsum("nca,nc->na", J, q)                            # (N,3)
    d = dirs[:, None, :].expand(R, S, 3).reshape(-1, 3)
    x = torch.cat([g, f2, geo], -1)
    rgb = torch.sigmoid(torch.cat([x, d], -1) @ A.t() + b_rgb)
    c = (g * d).sum(-1)
    half = -torch.relu(-c) * deltas.reshape(-1) * 0.5
    e1 = torch.sigmoid((sdf - half) * inv_s)
    e2 = torch.sigmoid((sdf + half) * inv_s)
    alpha = ((e1 - e2 + 1e-5) / (e1 + 1e-5)).clip(0.0, 1.0).reshape(R, S)
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-7], 1), 1)[:, :-1]
    w = alpha * T
    normal = F.normalize(g, dim=-1)
    one = torch.ones_like(sdf[:, None])
    vals = torch.cat([f2, geo, g, normal, rgb, starts.reshape(-1, 1), one, 0 * one], -1)
    comp = (w[..., None] * vals.reshape(R, S, -1)).sum(1)
    out = dict(sdf=sdf.reshape(R, S), grad=g.reshape(R, S, 3), weights=w, comp=comp)
    if keep:
        out["_saved"] = dict(p=p, scene=scene, rows=rows, w8=w8, dw8=dw8, f=f, f2=f2, h0=h0, a1=a1,
                             geo=geo, q=q, g=g, rgb=rgb, alpha=alpha, T=T, d=d, x=x, sdf=sdf,
                             e1=e1, e2=e2, c=c, half=half)
    return out


def field_render_backward(vol, origins, dirs, starts, deltas, MW, c0, bc1, W1, b1, A, b_rgb, inv_s,
                          g_sdf, g_grad, g_comp, norm_pts=True, norm_padding=0.1, debug=False):
    """Hand-derived gradients of ``field_render`` given upstream g_sdf (R,S), g_grad (R,S,3),
    g_comp (R,NC).  Returns dict(vol, MW, c0, bc1, W1, b1, A, b_rgb, inv_s).  The formulas below are
    the ones csrc/raymarch_fused.hip evaluates."""
    with torch.no_grad():
        out = field_render(vol, origins, dirs, starts, deltas, MW, c0, bc1, W1, b1, A, b_rgb, inv_s,
                           norm_pts, norm_padding, keep=True)
        s = out["_saved"]
        B, Z, Y, X, C = vol.shape
        R, S = starts.shape
        H = c0.shape[0]
        Fh = MW.shape[1]
        G = W1.shape[0] - 1
        F2 = C - Fh
        M, Wc1, v1 = MW[:H], MW[H:], W1[0]
        w = out["weights"]
        alpha, T = s["alpha"], s["T"]
        g, d, x, rgb, sdf = s["g"], s["d"], s["x"], s["rgb"], s["sdf"]
        gn = g.norm(dim=-1, keepdim=True).clamp_min(1e-12)
        normal = g / gn
        k0 = F2 + G
        u_x = torch.cat([g_comp[:, k0:k0 + 3], g_comp[:, :k0]], -1)      # order of x = [g, f', geo]
        u_n, u_rgb = g_comp[:, k0 + 3:k0 + 6], g_comp[:, k0 + 6:k0 + 9]
        u_t, u_w = g_comp[:, k0 + 9:k0 + 10], g_comp[:, k0 + 10:k0 + 11]
        rs = lambda v: v.reshape(R, S, -1)
        # ray level: d w_k, then d alpha_k through the transmittance product
        gw = (u_t * starts + (rs(normal) * u_n[:, None]).sum(-1) + (rs(rgb) * u_rgb[:, None]).sum(-1)
              + (rs(x) * u_x[:, None]).sum(-1) + u_w)
        qk = gw * w
        after = torch.flip(torch.cumsum(torch.flip(qk, [1]), 1), [1]) - qk
        g_alpha = (gw * T - after / (1.0 - alpha + 1e-7)).reshape(-1)
        wk = w.reshape(-1, 1)
        rep = lambda v: v[:, None, :].expand(R, S, v.shape[-1]).reshape(R * S, -1)
        # alpha -> sdf, cos, inv_s
        e1, e2 = s["e1"], s["e2"]
        raw = (e1 - e2 + 1e-5) / (e1 + 1e-5)
        g_raw = torch.where((raw >= 0) & (raw <= 1), g_alpha, torch.zeros_like(g_alpha))
        g_e1 = g_raw * e2 / (e1 + 1e-5) ** 2
        g_e2 = -g_raw / (e1 + 1e-5)
        gu1 = g_e1 * e1 * (1 - e1)
        gu2 = g_e2 * e2 * (1 - e2)
        g_sdf_t = g_sdf.reshape(-1) + inv_s * (gu1 + gu2)
        g_half = inv_s * (gu2 - gu1)
        g_invs = (gu1 * (sdf - s["half"]) + gu2 * (sdf + s["half"])).sum()
        g_c = torch.where(s["c"] < 0, g_half * deltas.reshape(-1) * 0.5, torch.zeros_like(g_half))
        gg = g_grad.reshape(-1, 3) + g_c[:, None] * d
        # normal composite
        gnrm = wk * rep(u_n)
        gg = gg + (gnrm - normal * (normal * gnrm).sum(-1, keepdim=True)) / gn
        # colour head
        gy = wk * rep(u_rgb) * rgb * (1 - rgb)
        xd = torch.cat([x, d], -1)
        g_A = gy.t() @ xd
        g_brgb = gy.sum(0)
        gx = gy @ A[:, :3 + F2 + G] + wk * rep(u_x)
        gg = gg + gx[:, :3]
        g_f2 = gx[:, 3:3 + F2]
        g_geo = gx[:, 3 + F2:]
        # SDF MLP, second order included
        gh = torch.cat([g_sdf_t[:, None], g_geo], -1)                   # (N,1+G)
        D8 = torch.einsum("nka,na->nk", s["dw8"], gg)                    # (N,8)  sum_a gg_a d_a w_c
        vr = vol.reshape(-1, C)[s["rows"]]
        gq = (vr[..., :Fh] * D8[..., None]).sum(1)                       # (N,F) = J gg
        sig0 = softplus_d1(s["h0"])
        t_ = v1 * sig0
        gt = gq @ M.t()                                                  # (N,H)
        ga1 = gh @ W1                                                    # (N,H)
        gh0 = ga1 * sig0 + gt * v1 * softplus_d2(s["h0"])
        gf = gh0 @ M + ga1 @ Wc1                                         # (N,F)
        g_W1 = gh.t() @ s["a1"]
        g_W1[0] += (gt * sig0).sum(0) + Wc1 @ gq.sum(0)
        g_b1 = gh.sum(0)
        g_M = gh0.t() @ s["f"] + t_.t() @ gq
        g_Wc1 = ga1.t() @ s["f"] + torch.outer(v1, gq.sum(0))
        g_MW = torch.cat([g_M, g_Wc1], 0)
        g_c0 = gh0.sum(0)
        g_bc1 = ga1.sum(0)
        # volume
        gfeat = torch.cat([gf, g_f2], -1)                                # (N,C)
        contrib = s["w8"][..., None] * gfeat[:, None, :]
        contrib[..., :Fh] += D8[..., None] * s["q"][:, None, :]
        g_vol = torch.zeros_like(vol.reshape(-1, C))
        g_vol.index_add_(0, s["rows"].reshape(-1), contrib.reshape(-1, C))
        res = dict(vol=g_vol.reshape(vol.shape), MW=g_MW, c0=g_c0, bc1=g_bc1, W1=g_W1, b1=g_b1, A=g_A,
                   b_rgb=g_brgb, inv_s=g_invs)
        if debug:  # the per-sample intermediates the backward kernel writes out
            res["_dbg"] = dict(gfeat=gfeat, gvec=gg, gh0=gh0, ga1=ga1, tmat=t_, gq=gq, gh=gh, gy=gy,
                               g_alpha=g_alpha, gw=gw, saved=s)
        return res
