"""numpy restatement of sparse-conv rulebook ("indice pair") generation.

Follows the definitions the reference relies on at
ponder/models/sparse_unet/spconv_unet_v1m1_base.py:112-119 (SubMConv3d k5), :47-66 (k3),
:135-142 (SparseConv3d k2 s2), :171-177 (SparseInverseConv3d) - spconv 2.x itself is out of tree.

Canonical order (shared with ponderv2_amd/csrc/rulebook.hip): pairs grouped by kernel offset
k = ((ix*K)+iy)*K+iz ascending, inside a group by output row ascending; strided-conv outputs
sorted by (b,x,y,z).
"""
import numpy as np


def _linear_key(coords, shift=0):
    c = coords.astype(np.int64)
    # 16 bits per field with a +16 bias, like the device key (order == lexicographic (b,x,y,z))
    return (c[:, 0] << 48) | ((c[:, 1] + 16 + shift) << 32) | ((c[:, 2] + 16 + shift) << 16) \
        | (c[:, 3] + 16 + shift)


def subm_rulebook(coords: np.ndarray, ksize: int):
    """coords int32 [N,4] -> (pair_in int32[P], pair_out int32[P], kstart int64[K^3+1])."""
    coords = np.asarray(coords, dtype=np.int32)
    n = coords.shape[0]
    K = ksize ** 3
    if n == 0:
        return np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(K + 1, np.int64)
    keys = _linear_key(coords)
    order = np.argsort(keys, kind="stable")
    skeys = keys[order]
    r = ksize // 2
    pin, pout, kstart = [], [], [0]
    rows = np.arange(n, dtype=np.int64)
    for dx in range(-r, r + 1):
        for dy in range(-r, r + 1):
            for dz in range(-r, r + 1):
                if dx == 0 and dy == 0 and dz == 0:
                    j, ok = rows, np.ones(n, bool)
                else:
                    q = coords.astype(np.int64) + np.array([0, dx, dy, dz], dtype=np.int64)
                    ok = (q[:, 1:] >= 0).all(axis=1)
                    qk = _linear_key(np.where(ok[:, None], q, 0))
                    pos = np.searchsorted(skeys, qk, side="left")  # first match = smallest row
                    pos_c = np.minimum(pos, n - 1)
                    ok &= skeys[pos_c] == qk
                    j = order[pos_c]
                pin.append(j[ok].astype(np.int32))
                pout.append(rows[ok].astype(np.int32))
                kstart.append(kstart[-1] + int(ok.sum()))
    return np.concatenate(pin), np.concatenate(pout), np.asarray(kstart, dtype=np.int64)


def downsample_rulebook(coords: np.ndarray, stride: int, out_shape):
    """kernel == stride, padding 0.  Returns (out_coords int32[M,4], pair_in, pair_out, kstart)."""
    coords = np.asarray(coords, dtype=np.int32)
    s = int(stride)
    K = s ** 3
    oc = coords.copy()
    oc[:, 1:] = coords[:, 1:] // s
    keep = (oc[:, 1:] < np.asarray(out_shape, dtype=np.int32)[None, :]).all(axis=1)
    keys = _linear_key(oc)
    uniq = np.unique(keys[keep])  # sorted
    m = uniq.shape[0]
    out_coords = np.stack(
        [(uniq >> 48), ((uniq >> 32) & 0xFFFF) - 16, ((uniq >> 16) & 0xFFFF) - 16,
         (uniq & 0xFFFF) - 16], axis=1).astype(np.int32)
    orow = np.searchsorted(uniq, keys)
    rem = coords[:, 1:] - oc[:, 1:] * s
    k = (rem[:, 0] * s + rem[:, 1]) * s + rem[:, 2]
    rows = np.arange(coords.shape[0])
    pin, pout, kstart = [], [], [0]
    for kk in range(K):
        sel = keep & (k == kk)
        o = orow[sel]
        i = rows[sel]
        srt = np.argsort(o, kind="stable")
        pin.append(i[srt].astype(np.int32))
        pout.append(o[srt].astype(np.int32))
        kstart.append(kstart[-1] + int(sel.sum()))
    if m == 0:
        out_coords = np.zeros((0, 4), np.int32)
    return out_coords, np.concatenate(pin), np.concatenate(pout), np.asarray(kstart, np.int64)
