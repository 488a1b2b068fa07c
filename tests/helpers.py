"""Shared synthetic inputs for the parity tests (seeded, no dataset needed)."""
import numpy as np
import torch


def random_voxels(seed, batch=2, extent=(40, 36, 20), n_per_batch=1500, surface=True):
    """int32 [N,4] (b,x,y,z) unique voxels in a shuffled row order; roughly planar clusters when
    `surface` so that neighbour statistics look like a scanned room."""
    rng = np.random.default_rng(seed)
    rows = []
    for b in range(batch):
        if surface:
            pts = []
            for _ in range(6):  # a few random axis-aligned slabs
                axis = rng.integers(0, 3)
                p = rng.integers(0, extent, size=(n_per_batch // 3, 3))
                p[:, axis] = rng.integers(0, extent[axis]) + rng.integers(0, 2, size=len(p))
                p[:, axis] = np.clip(p[:, axis], 0, extent[axis] - 1)
                pts.append(p)
            p = np.concatenate(pts)
        else:
            p = rng.integers(0, extent, size=(n_per_batch * 2, 3))
        p = np.unique(p, axis=0)
        rng.shuffle(p)
        p = p[:n_per_batch]
        rows.append(np.concatenate([np.full((len(p), 1), b), p], axis=1))
    out = np.concatenate(rows).astype(np.int32)
    return out


def away_from_kinks(grid, sizes, align_corners, margin=2e-3):
    """Nudge grid coords so that un-normalised source coordinates keep `margin` from integers
    (the sampler is only piecewise smooth; finite differences must not straddle a cell edge)."""
    g = grid.clone()
    for a, size in enumerate(sizes):  # sizes = (W, H, D)
        x = ((g[..., a] + 1) / 2) * (size - 1) if align_corners else ((g[..., a] + 1) * size - 1) / 2
        frac = x - torch.floor(x)
        bad = (frac < margin) | (frac > 1 - margin)
        scale = (size - 1) / 2 if align_corners else size / 2
        g[..., a] = torch.where(bad, g[..., a] + 4 * margin / scale, g[..., a])
    return g
