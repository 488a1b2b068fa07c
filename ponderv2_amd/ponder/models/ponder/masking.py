"""Block masking of the backbone's input features (masked-autoencoder style pre-training).

Both PonderIndoor (ponder_indoor_base.py:120-171) and PonderOutdoor (ponder_outdoor_base.py:93-137)
hide a random ``ratio`` of the occupied ``size``^3-voxel blocks of every scene by overwriting the
features of the voxels inside with the learnable ``mtoken`` row.  The reference loops over scenes,
draws ``torch.rand(1, n_blocks)`` on the host for each and syncs on ``.item()``; here all scenes are
ranked in one device pass.
"""
import torch

from ..utils import offset2batch


def mask_blocks(grid_coord, feat, offset, size, ratio, mtoken, rand=None):
    """-> feat with the rows of masked blocks replaced by ``mtoken``.

    A block is kept iff the rank of its random key among its scene's blocks is below
    ``round(n_blocks * (1 - ratio))`` - the same set the reference keeps when ``rand`` carries its
    draws (blocks in lexicographic (scene, bx, by, bz) order, which is the order ``unique(dim=0)``
    returns on every backend)."""
    batch = offset2batch(offset, grid_coord.shape[0])
    block = torch.cat([batch[:, None], torch.div(grid_coord, size).int()], dim=-1)
    block, inverse = block.unique(sorted=True, return_inverse=True, dim=0)
    scene = block[:, 0].long()
    n_scene = torch.bincount(scene, minlength=offset.numel())
    if rand is None:
        rand = torch.rand(block.shape[0], device=block.device)
    key = rand.to(block.device, torch.float64) + scene.to(torch.float64) * 2.0
    order = torch.argsort(key)
    start = torch.cumsum(n_scene, 0) - n_scene
    rank = torch.empty_like(order)
    rank[order] = torch.arange(order.numel(), device=order.device)
    rank = rank - start[scene]
    n_keep = torch.round(n_scene.to(torch.float64) * (1 - ratio)).long()
    keep = rank < n_keep[scene]
    # where() instead of a boolean-mask assignment: no nonzero(), hence no host sync
    return torch.where(keep[inverse][:, None], feat, mtoken.to(feat.dtype))
