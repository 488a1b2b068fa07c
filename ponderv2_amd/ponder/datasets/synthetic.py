"""Seeded synthetic ScanNet-shaped RGB-D scenes (there is no dataset in the build environment).

Produces exactly the ``data_dict`` contract that ``point_collate_fn`` hands to
``PonderIndoor.forward`` (reference ponder/datasets/scannet.py:418-434,
ponder/datasets/utils.py:40-56): coord, grid_coord, feat=[color/127.5-1, normal], offset, segment,
condition, rgb, depth (raw mm), depth_scale, intrinsic, extrinsic (world->camera), semantic.

Scene: a 6.0 x 5.0 x 2.6 m room (floor + 4 walls) with 6 axis-aligned
boxes on the floor, surface-uniform samples with 4 mm noise, RandomDropout keep 20 %, then
``GridSample(0.02, fnv)``.  Views: pinhole cameras (ScanNet depth intrinsics scaled to the image
size) looking at the room centre; depth is the analytic z-depth ray cast of the same planes.
"""
import numpy as np
import torch

from .voxelize import GridSample

ROOM = np.array([6.0, 5.0, 2.6])


def _boxes(rng, n=6):
    out = []
    for _ in range(n):
        size = rng.uniform([0.4, 0.4, 0.3], [1.4, 1.2, 1.5])
        lo = np.concatenate([rng.uniform([0.2, 0.2], ROOM[:2] - size[:2] - 0.2), [0.0]])
        out.append((lo, lo + size))
    return out


def _faces(boxes):
    """List of axis-aligned rectangles: (axis, value, lo2, hi2, normal_sign, plane_id)."""
    faces = [(2, 0.0, np.zeros(2), ROOM[:2], +1.0, 0)]  # floor
    pid = 1
    for axis in (0, 1):
        other = [a for a in range(3) if a != axis]
        for val, sgn in ((0.0, +1.0), (ROOM[axis], -1.0)):
            faces.append((axis, val, np.zeros(2), ROOM[other], sgn, pid))
            pid += 1
    for lo, hi in boxes:
        for axis in range(3):
            other = [a for a in range(3) if a != axis]
            for val, sgn in ((lo[axis], -1.0), (hi[axis], +1.0)):
                if axis == 2 and sgn < 0:
                    continue  # bottom face lies on the floor
                faces.append((axis, val, lo[other], hi[other], sgn, pid))
                pid += 1
    return faces


def _sample_surface(rng, faces, n):
    area = np.array([np.prod(f[3] - f[2]) for f in faces])
    which = rng.choice(len(faces), size=n, p=area / area.sum())
    pts, nrm, pid = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros(n, dtype=np.int64)
    uv = rng.uniform(size=(n, 2))
    for i, (axis, val, lo2, hi2, sgn, p) in enumerate(faces):
        m = which == i
        other = [a for a in range(3) if a != axis]
        pts[np.ix_(m, other)] = lo2 + uv[m] * (hi2 - lo2)
        pts[m, axis] = val
        nrm[m, axis] = sgn
        pid[m] = p
    return pts, nrm, pid


def _ray_cast(faces, origin, dirs):
    """Nearest hit of rays (origin (3,), dirs (P,3)) with the rectangles and the ceiling.
    Returns (t (P,), plane id (P,), -1 where nothing is hit)."""
    P = dirs.shape[0]
    best = np.full(P, np.inf)
    pid = np.full(P, -1, dtype=np.int64)
    allf = list(faces) + [(2, ROOM[2], np.zeros(2), ROOM[:2], -1.0, -1)]  # ceiling: no label
    for axis, val, lo2, hi2, sgn, p in allf:
        d = dirs[:, axis]
        with np.errstate(divide="ignore", invalid="ignore"):
            t = (val - origin[axis]) / d
        other = [a for a in range(3) if a != axis]
        hit = origin[other] + t[:, None] * dirs[:, other]
        ok = (t > 1e-4) & np.isfinite(t) & (hit >= lo2 - 1e-9).all(1) & (hit <= hi2 + 1e-9).all(1)
        better = ok & (t < best)
        best[better] = t[better]
        pid[better] = p
    return best, pid


def _look_at(eye, target):
    """world->camera 4x4, OpenCV axes (x right, y down, z forward)."""
    z = target - eye
    z = z / np.linalg.norm(z)
    x = np.cross(z, np.array([0.0, 0.0, 1.0]))
    x = x / np.linalg.norm(x)
    y = np.cross(z, x)
    R = np.stack([x, y, z])  # rows = camera axes in world
    E = np.eye(4)
    E[:3, :3] = R
    E[:3, 3] = -R @ eye
    return E


def make_scene(seed, n_raw=120000, keep=0.2, grid_size=0.02, num_views=2, image_hw=(480, 640),
               n_voxels=None, condition="ScanNet", num_classes=20, voxelize=True):
    """One synthetic scene as numpy arrays (the per-sample dict a Dataset would return).
    ``voxelize=False`` skips the host GridSample and returns the raw (dropout-thinned) points, for
    runs that voxelise on the device (``datasets.voxelize.device_grid_sample``)."""
    rng = np.random.default_rng(seed)
    boxes = _boxes(rng)
    faces = _faces(boxes)
    pts, nrm, pid = _sample_surface(rng, faces, n_raw)
    pts = pts + rng.normal(scale=0.004, size=pts.shape)
    sel = rng.permutation(n_raw)[: int(n_raw * keep)]  # RandomDropout(keep 20 %)
    plane_rgb = rng.uniform(40, 255, size=(len(faces) + 1, 3))
    data = dict(coord=pts[sel].astype(np.float32),
                color=np.clip(plane_rgb[pid[sel]] + rng.uniform(-20, 20, (len(sel), 3)), 0, 255)
                .astype(np.float32),
                normal=nrm[sel].astype(np.float32), segment=(pid[sel] % num_classes).astype(np.int64))
    if voxelize:
        state = np.random.get_state()
        np.random.seed(seed)  # GridSample draws from numpy's global generator, as the reference does
        data = GridSample(grid_size=grid_size, hash_type="fnv", mode="train",
                          return_grid_coord=True)(data)
        np.random.set_state(state)
    if n_voxels is not None and voxelize:  # trim / pad-by-repeat-free: trim only (config 1 asks for 20 000)
        for k in ("coord", "color", "normal", "segment", "grid_coord"):
            data[k] = data[k][:n_voxels]
        data["grid_coord"] = data["grid_coord"] - data["grid_coord"].min(0)

    H, W = image_hw
    fx = fy = 577.87 * W / 640.0
    cx, cy = (W - 1) / 2.0, (H - 1) / 2.0
    K = np.eye(4)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = fx, fy, cx, cy
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    cam_dirs = np.stack([(xs - cx) / fx, (ys - cy) / fy, np.ones_like(xs, dtype=np.float64)], -1)
    rgb, depth, sem, extr = [], [], [], []
    for _ in range(num_views):
        eye = rng.uniform([0.8, 0.8, 1.2], [ROOM[0] - 0.8, ROOM[1] - 0.8, 1.8])
        target = ROOM / 2 + rng.uniform(-0.5, 0.5, 3) * np.array([1, 1, 0.3])
        E = _look_at(eye, target)
        dirs = cam_dirs.reshape(-1, 3) @ E[:3, :3]  # camera -> world (R^T applied to rows)
        t, p = _ray_cast(faces, eye, dirs)
        z = np.where(np.isfinite(t), t, 0.0)  # dirs have camera z == 1, so t IS the z-depth
        z_mm = np.round(z * 1000.0)
        z_mm[(z_mm > 65535) | (p < 0)] = 0
        col = plane_rgb[np.where(p >= 0, p, len(faces))] / 255.0
        col = np.clip(col + rng.uniform(-0.05, 0.05, col.shape), 0, 1)
        rgb.append(col.reshape(H, W, 3).astype(np.float32))
        depth.append(z_mm.reshape(H, W).astype(np.float32))
        sem.append(np.where(p >= 0, p % num_classes, -1).reshape(H, W).astype(np.int64))
        extr.append(E.astype(np.float32))
    data.update(rgb=np.stack(rgb), depth=np.stack(depth), semantic=np.stack(sem),
                extrinsic=np.stack(extr), intrinsic=np.stack([K.astype(np.float32)] * num_views),
                depth_scale=np.float32(1.0 / 1000.0), condition=condition)
    return data


def collate_fn(samples):
    """List of per-scene dicts -> batch dict of torch tensors (concatenate point keys, stack view
    keys, cumulative ``offset``)."""
    cat = lambda k, dt: torch.from_numpy(np.concatenate([s[k] for s in samples]).astype(dt))  # noqa
    stack = lambda k: torch.from_numpy(np.stack([s[k] for s in samples]))  # noqa: E731
    counts = [len(s["coord"]) for s in samples]
    feat = np.concatenate([np.concatenate([s["color"] / 127.5 - 1, s["normal"]], 1) for s in samples])
    grid = {}
    if "grid_coord" in samples[0]:
        # sparse_shape = max grid coordinate + 96 (spconv_unet_v1m1_base.py:248), computed here on
        # the host so the backbone does not read it back from the device
        top = np.max([s["grid_coord"].max(0) for s in samples], axis=0)
        grid = dict(grid_coord=cat("grid_coord", np.int64), sparse_shape=[int(v) + 96 for v in top])
    return dict(coord=cat("coord", np.float32), **grid,
                feat=torch.from_numpy(feat.astype(np.float32)), segment=cat("segment", np.int64),
                offset=torch.tensor(np.cumsum(counts), dtype=torch.int64),
                offset_host=[int(v) for v in np.cumsum(counts)],
                # largest side of every scene's bounding box: lets the model skip a device read
                # when no scene is anywhere near the "smaller than the dense grid" case
                extent_host=[float((s["coord"].max(0) - s["coord"].min(0)).max()) for s in samples],
                condition=[s["condition"] for s in samples], rgb=stack("rgb"), depth=stack("depth"),
                semantic=stack("semantic"), extrinsic=stack("extrinsic"), intrinsic=stack("intrinsic"),
                depth_scale=torch.tensor([s["depth_scale"] for s in samples], dtype=torch.float32))


class SyntheticRGBDDataset(torch.utils.data.Dataset):
    """Endless-ish stream of seeded scenes; ``len`` scenes, scene i uses seed ``base_seed + i``."""

    collate_fn = staticmethod(collate_fn)

    def __init__(self, length=64, base_seed=0, num_views=2, image_hw=(480, 640), n_raw=120000,
                 keep=0.2, grid_size=0.02, n_voxels=None, loop=1, condition="ScanNet",
                 num_classes=20, voxelize=True, **kwargs):
        self.length, self.base_seed, self.loop = length, base_seed, loop
        self.kw = dict(num_views=num_views, image_hw=tuple(image_hw), n_raw=n_raw, keep=keep,
                       grid_size=grid_size, n_voxels=n_voxels, condition=condition,
                       num_classes=num_classes, voxelize=voxelize)

    def __len__(self):
        return self.length * self.loop

    def __getitem__(self, idx):
        return make_scene(self.base_seed + idx % self.length, **self.kw)
