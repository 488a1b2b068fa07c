"""Block masking of the backbone's input features (masked-autoencoder style pre-training).

Both PonderIndoor (ponder_indoor_base.py:120-171) and PonderOutdoor (ponder_outdoor_base.py:93-137)
hide a random ``ratio`` of the occupied ``size``^3-voxel blocks of every scene by overwriting the
features of the voxels inside with the learnable ``mtoken`` row.  The reference loops over scenes,
draws ``torch.rand(1, n_blocks)`` on the host for each and syncs on ``.item()``; here all scenes are
ranked in one device pass.
"""
import os

import torch

from ..utils import offset2batch

_CHECK_RANGE = os.environ.get("PV2_CHECK_MASK_RANGE", "0") == "1"


def mask_blocks(grid_coord, feat, offset, size, ratio, mtoken, rand=None):
    """-> feat with the rows of masked blocks replaced by ``mtoken``.

    A block is kept iff the rank of its random key among its scene's blocks is below
    ``round(n_blocks * (1 - ratio))`` - the same set the reference keeps when ``rand`` carries its
    draws (one per block, blocks in lexicographic (scene, bx, by, bz) order: the order
    ``unique(dim=0)`` returns on every backend).

    No compaction, hence no device -> host read: ``unique(dim=0)`` (7 ms of HOST time per step on the
    nuScenes batch - a lexicographic row sort plus a blocking size read) is replaced by two sorts of
    per-VOXEL keys; a block is represented by the first voxel of its run in key order."""
    n = grid_coord.shape[0]
    dev = grid_coord.device
    batch = offset2batch(offset, n).long()
    b = torch.div(grid_coord, size).int().long()   # (true division, then truncation: the reference's :115)
    # one integer per block, ordered like the rows (scene, bx, by, bz): 16 bits per coordinate,
    # i.e. block coordinates in [-32768, 32767] and < 32768 scenes.  Out-of-range coordinates are
    # CLAMPED (no device -> host read, so no exception can be raised from here without a sync):
    # voxels beyond the limit share the outermost block of their axis instead of spilling into the
    # neighbouring field of the key.  With the reference's own shapes (nuScenes 1080 x 1080 x 80
    # voxels, ScanNet < 4096 per axis) block coordinates stay below 2^11.  PV2_CHECK_MASK_RANGE=1
    # turns the clamp into an assertion (one blocking read; for debugging a new dataset).
    if _CHECK_RANGE:
        assert int(b.abs().max()) < 32768 and offset.numel() < 32768, \
            "block masking packs 16 bits per block coordinate"
    b = b.clamp(-32768, 32767)
    key = (batch << 48) | ((b[:, 0] + 32768) << 32) | ((b[:, 1] + 32768) << 16) | (b[:, 2] + 32768)
    skey, perm = torch.sort(key)
    pos = torch.arange(n, device=dev)
    head = torch.ones(n, dtype=torch.bool, device=dev)
    head[1:] = skey[1:] != skey[:-1]
    head_pos = torch.cummax(torch.where(head, pos, torch.zeros_like(pos)), 0).values   # run start
    scene = skey >> 48
    n_scene = torch.zeros(offset.numel(), dtype=torch.long, device=dev).scatter_add_(0, scene, head.long())
    if rand is None:
        r = torch.rand(n, device=dev, dtype=torch.float64)[head_pos]                   # one draw per block
    else:   # the reference's draws, one per block in lexicographic order
        r = rand.to(dev, torch.float64)[torch.cumsum(head.long(), 0) - 1]
    # rank of every block among its scene's blocks: blocks (run heads) first, ordered by (scene, draw)
    inf = torch.full((n,), float("inf"), dtype=torch.float64, device=dev)
    order = torch.argsort(torch.where(head, r + scene.to(torch.float64) * 2.0, inf))
    rank = torch.empty_like(order)
    rank[order] = pos
    start = torch.cumsum(n_scene, 0) - n_scene
    n_keep = torch.round(n_scene.to(torch.float64) * (1 - ratio)).long()
    keep_head = (rank - start[scene]) < n_keep[scene]          # meaningful at run heads
    keep = torch.empty(n, dtype=torch.bool, device=dev)
    keep[perm] = keep_head[head_pos]
    # where() instead of a boolean-mask assignment: no nonzero(), hence no host sync
    return torch.where(keep[:, None], feat, mtoken.to(feat.dtype))
