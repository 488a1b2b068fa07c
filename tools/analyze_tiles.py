#!/usr/bin/env python
"""How much matrix work does an output-stationary sparse conv waste on the bench geometry?  (CPU, numpy.)

For every level of the SpUNet geometry of BASELINE configs[1] (2 synthetic ScanNet-shaped scenes, 46 842 voxels)
and its 27-offset submanifold rulebook: the ratio  (offsets present in a T-row tile) x T / pairs  - the MFMA work
of a kernel that walks every offset present in a tile for all of the tile's rows, over the useful work - with rows
in storage order and with rows sorted by the bit mask of their present offsets (what pv2_osm_plan does), for
T = 32 / 64 / 128, plus a few alternative sort keys.  DESIGN.md section 3.2c quotes the numbers."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_batch  # noqa: E402
from oracle.rulebook import downsample_rulebook, subm_rulebook  # noqa: E402


def waste(tab, order, pairs, T):
    K, n = tab.shape
    nt = (n + T - 1) // T
    pad = np.zeros((K, nt * T), bool)
    pad[:, :n] = tab[:, order]
    return pad.reshape(K, nt, T).any(2).sum() * T / pairs


def main():
    b = make_batch(0, 2, 2, "cpu")
    gc, off = b["grid_coord"].numpy(), b["offset"].numpy()
    batch = np.zeros(len(gc), np.int32)
    batch[off[0]:] = 1
    cur = np.concatenate([batch[:, None], gc.astype(np.int32)], 1)
    shape = gc.max(0) + 1
    print("level rows pairs/row | storage order T=32 | mask order T=32 64 128 | popcount,mask | rare-bits-first | masks")
    for lvl in range(5):
        pin, pout, ks = subm_rulebook(cur, 3)
        n, K, P = len(cur), 27, len(pin)
        tab = np.zeros((K, n), bool)
        for k in range(K):
            tab[k, pout[ks[k]:ks[k + 1]]] = True
        masks = np.zeros(n, np.int64)
        for k in range(K):
            masks |= tab[k].astype(np.int64) << k
        by_mask = np.argsort(masks, kind="stable")
        freq = np.argsort(tab.sum(1))[::-1]
        m2 = np.zeros(n, np.int64)
        for j, k in enumerate(freq):
            m2 |= tab[k].astype(np.int64) << j
        print("L%d %6d %5.2f | %5.2f | %5.2f %5.2f %5.2f | %5.2f | %5.2f | %d" % (
            lvl, n, P / n, waste(tab, np.arange(n), P, 32),
            waste(tab, by_mask, P, 32), waste(tab, by_mask, P, 64), waste(tab, by_mask, P, 128),
            waste(tab, np.lexsort((masks, tab.sum(0))), P, 32),
            waste(tab, np.argsort(m2, kind="stable"), P, 32), len(np.unique(masks))))
        shape = (shape + 1) // 2
        cur, _, _, _ = downsample_rulebook(cur, 2, shape)


if __name__ == "__main__":
    main()
