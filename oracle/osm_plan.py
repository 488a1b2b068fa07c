"""numpy restatement of ``pv2_osm_plan`` (ponderv2_amd/csrc/sparse_conv_osm.hip): the row order, permuted gather
table and per-tile offset masks of the mask-grouped output-stationary sparse conv.

TEST INFRASTRUCTURE ONLY.  The plan is this repository's own construct (the reference reaches its sparse convs
through spconv 2.x's indice_conv, ponder/models/sparse_unet/spconv_unet_v1m1_base.py:41-66,111-181, which has no such
notion); what the reference fixes is the RESULT of the conv, which tests check against oracle/sparse_ops.py.  This
file pins the plan's definition so that the device kernels can be compared with it bit for bit."""
import numpy as np


def osm_plan(tbl: np.ndarray, n_valid: int, n_pad: int):
    """tbl int32 [K, stride] gather table (input row feeding output row o under offset k, or -1), the first
    ``n_valid`` columns valid -> (perm int32 [n_valid], tblp int32 [K, n_pad], tmask uint32 [n_pad // 32]).

    perm: rows sorted STABLY by the bit mask of their present offsets (bit k <=> tbl[k, o] >= 0);
    tblp[k, r] = tbl[k, perm[r]] for r < n_valid, -1 beyond; tmask[t] = OR of the masks of sorted rows
    32 t .. 32 t + 31."""
    K = tbl.shape[0]
    assert K <= 31 and n_pad % 256 == 0 and n_pad >= n_valid
    t = np.asarray(tbl)[:, :n_valid]
    mask = np.zeros(n_valid, np.int64)
    for k in range(K):
        mask |= (t[k] >= 0).astype(np.int64) << k
    perm = np.argsort(mask, kind="stable").astype(np.int32)
    tblp = -np.ones((K, n_pad), np.int32)
    tblp[:, :n_valid] = t[:, perm]
    sorted_mask = np.zeros(n_pad, np.int64)
    sorted_mask[:n_valid] = mask[perm]
    tmask = np.bitwise_or.reduce(sorted_mask.reshape(-1, 32), axis=1).astype(np.uint32)
    return perm, tblp, tmask


def tile_waste(tbl: np.ndarray, n_valid: int, order: np.ndarray, tile: int = 32) -> float:
    """(offsets present in a tile) x tile / pairs over all tiles of ``order``: the matrix work of an
    output-stationary kernel that walks every offset present in a tile, relative to the useful work."""
    t = (np.asarray(tbl)[:, :n_valid] >= 0)[:, order]
    K, n = t.shape
    nt = (n + tile - 1) // tile
    pad = np.zeros((K, nt * tile), bool)
    pad[:, :n] = t
    return float(pad.reshape(K, nt, tile).any(2).sum() * tile / max(int(t.sum()), 1))
