"""Outdoor (nuScenes-shaped) input side of PonderOutdoor: the host transforms that turn one lidar
sweep + six calibrated cameras into supervision rays, and a seeded synthetic sweep generator
(there is no dataset in the build environment).

Transforms restate ponder/datasets/transform.py: PointRangeFilter :232-264, ProjectOnImage
:267-318, RaySample :321-378.  They make the same numpy calls in the same order as the reference,
so the same ``np.random`` state selects the same rays (pinned by tests/test_oracle.py against the
reference's classes, and by tests/golden/lidar_transforms.npz where the reference is absent).

The produced sample follows the ``Collect`` of configs/nuscenes/pretrain-ponder-spunet-v1m1-0-base.py
:186-199: coord, grid_coord, segment, condition, ray_start, ray_end, feat = [coord, strength],
``offset`` over coord and ``ray_offset`` over ray_start.
"""
import numpy as np
import torch

from .voxelize import GridSample

POINT_KEYS = ("color", "normal", "strength", "segment", "instance")


class PointRangeFilter:
    """Keep the points strictly inside ``point_cloud_range`` shrunk by ``padding``."""

    def __init__(self, point_cloud_range=(-80, -80, -3, 80, 80, 1), padding=0.0):
        self.point_cloud_range, self.padding = point_cloud_range, padding

    def __call__(self, data_dict):
        c, r, p = data_dict["coord"], self.point_cloud_range, self.padding
        inside = np.ones(len(c), dtype=bool)
        for axis in range(3):
            inside &= (c[:, axis] > r[axis] + p) & (c[:, axis] < r[axis + 3] - p)
        idx = np.nonzero(inside)[0]
        data_dict["coord"] = c[idx]
        for k in POINT_KEYS:
            if k in data_dict:
                data_dict[k] = data_dict[k][idx]
        return data_dict


class ProjectOnImage:
    """Project the sweep into every camera; per camera a pixel-space coordinate (u, v, depth) and
    the mask of points that land inside the image, in front of the camera and farther than
    ``close_radius`` from the sensor; optionally only the nearest point per pixel survives."""

    def __init__(self, filter_overlap=True, close_radius=0.0):
        self.filter_overlap, self.close_radius = filter_overlap, close_radius

    def project_on_image(self, coord, lidar2img, img):
        eps = 1e-5
        homo = np.concatenate([coord, np.ones_like(coord[:, :1])], axis=-1)
        far_enough = np.linalg.norm(homo[:, :2], axis=-1) > self.close_radius
        img_coord, proj_mask = [], []
        for cam in range(len(img)):
            uvd = homo @ lidar2img[cam].T
            uvd[:, :2] /= np.maximum(uvd[:, 2:3], eps)
            h, w = img[cam].shape[0], img[cam].shape[1]
            proj_mask.append(far_enough & (uvd[:, 2] > eps) & (uvd[:, 0] > 0) & (uvd[:, 1] > 0)
                             & (uvd[:, 0] < w) & (uvd[:, 1] < h))
            img_coord.append(uvd[:, :3])
        return img_coord, proj_mask

    def filter_overlap_coord(self, img_coord, proj_mask, img):
        for cam in range(len(img)):
            cand = np.nonzero(proj_mask[cam])[0]
            uvd = img_coord[cam][cand]
            px, depth = uvd[:, :2].astype(np.int32), uvd[:, 2]
            pixel = px[:, 0] + px[:, 1] * img[cam].shape[1]
            order = (pixel + depth / 100.0).argsort()  # by pixel, nearest first inside a pixel
            pixel = pixel[order]
            first = np.ones((pixel.shape[0],), dtype=bool)
            first[1:] = pixel[1:] != pixel[:-1]
            proj_mask[cam][cand[order[~first]]] = False
        return proj_mask

    def __call__(self, data_dict):
        img, lidar2img = data_dict["img"], data_dict["lidar2img"]
        img_coord, proj_mask = self.project_on_image(data_dict["coord"], lidar2img, img)
        if self.filter_overlap:
            proj_mask = self.filter_overlap_coord(img_coord, proj_mask, img)
        data_dict["img_coord"], data_dict["img_proj_mask"] = img_coord, proj_mask
        return data_dict


class RaySample:
    """Per camera, draw ``point_nsample`` of the projected lidar returns: the ray starts at the
    camera centre (in lidar coordinates) and ends on the return."""

    def __init__(self, point_nsample, point_ratio=None, fetch_color=True, fetch_segment=True):
        self.point_nsample, self.point_ratio = point_nsample, point_ratio
        self.fetch_color, self.fetch_segment = fetch_color, fetch_segment

    def __call__(self, data_dict):
        img_coord, proj_mask = data_dict["img_coord"], data_dict["img_proj_mask"]
        lidar2cam = data_dict["lidar2cam"]
        starts, ends, colors, segments = [], [], [], []
        for cam in range(len(proj_mask)):
            cand = np.nonzero(proj_mask[cam])[0]
            want = (int(len(cand) * self.point_ratio) if self.point_nsample is None
                    else self.point_nsample)
            n = min(len(cand), want)
            if n == 0:
                continue
            cand = cand[np.random.choice(len(cand), n, replace=False)]
            centre = np.linalg.inv(lidar2cam[cam])[None, :3, 3]
            if self.fetch_segment:
                segments.append(data_dict["segment"][cand])
            if self.fetch_color:
                uv = img_coord[cam][cand]
                colors.append(data_dict["img"][cam][uv[:, 1].astype(np.int32),
                                                    uv[:, 0].astype(np.int32)] / 255.0)
            starts.append(np.repeat(centre, len(cand), axis=0))
            ends.append(data_dict["coord"][cand])
        data_dict["ray_start"] = np.concatenate(starts, axis=0)
        data_dict["ray_end"] = np.concatenate(ends, axis=0)
        if self.fetch_segment:
            data_dict["ray_segment"] = np.concatenate(segments, axis=0)
        if self.fetch_color:
            data_dict["ray_color"] = np.concatenate(colors, axis=0)
        return data_dict


# ---------------------------------------------------------------------------------------------
# synthetic sweep: 32-beam spinning lidar over a ground plane with boxes (cars, walls), 6 cameras
# ---------------------------------------------------------------------------------------------
SCENE_RANGE = (-54.0, -54.0, -5.0, 54.0, 54.0, 3.0)
LIDAR_HEIGHT = 1.84      # nuScenes: the lidar sits ~1.84 m above the road
IMAGE_HW = (900, 1600)   # nuScenes camera images
CAM_YAW_DEG = (0.0, -55.0, 55.0, 180.0, -110.0, 110.0)  # front, front-right/left, back, back-r/l


def _street_boxes(rng, n_cars=24, n_walls=10):
    """Axis-aligned boxes standing on the ground: (lo, hi, class id)."""
    boxes = []
    for _ in range(n_cars):
        size = rng.uniform([3.8, 1.7, 1.4], [5.2, 2.1, 2.0])
        if rng.uniform() < 0.5:
            size = size[[1, 0, 2]]
        r, a = rng.uniform(5.0, 45.0), rng.uniform(0, 2 * np.pi)
        c = np.array([r * np.cos(a), r * np.sin(a)])
        lo = np.array([c[0] - size[0] / 2, c[1] - size[1] / 2, -LIDAR_HEIGHT])
        boxes.append((lo, lo + size, 3))
    for _ in range(n_walls):
        length, height = rng.uniform(15.0, 40.0), rng.uniform(4.0, 9.0)
        size = np.array([length, 1.0, height]) if rng.uniform() < 0.5 else np.array([1.0, length, height])
        r, a = rng.uniform(15.0, 50.0), rng.uniform(0, 2 * np.pi)
        c = np.array([r * np.cos(a), r * np.sin(a)])
        lo = np.array([c[0] - size[0] / 2, c[1] - size[1] / 2, -LIDAR_HEIGHT])
        boxes.append((lo, lo + size, 14))
    return boxes


def _cast(boxes, dirs, max_range=100.0):
    """Nearest hit of rays from the origin with the ground plane and the boxes.
    -> (range (P,), class id (P,), surface id (P,)); range = inf where nothing is hit."""
    P = dirs.shape[0]
    best = np.full(P, np.inf)
    cls = np.full(P, -1, dtype=np.int64)
    sid = np.full(P, -1, dtype=np.int64)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = -LIDAR_HEIGHT / dirs[:, 2]
    ok = (t > 0) & np.isfinite(t) & (t < max_range)
    best[ok], cls[ok], sid[ok] = t[ok], 10, 0
    surface = 1
    for lo, hi, c in boxes:
        for axis in range(3):
            other = [a for a in range(3) if a != axis]
            for val in (lo[axis], hi[axis]):
                with np.errstate(divide="ignore", invalid="ignore"):
                    t = val / dirs[:, axis]
                hit = t[:, None] * dirs[:, other]
                ok = ((t > 1e-3) & np.isfinite(t) & (t < best)
                      & (hit >= lo[other]).all(1) & (hit <= hi[other]).all(1))
                best[ok], cls[ok], sid[ok] = t[ok], c, surface
                surface += 1
    return best, cls, sid


def _camera_rig():
    """lidar2cam (6,4,4), cam_intrinsic (6,4,4), lidar2img (6,4,4); OpenCV camera axes."""
    K = np.eye(4)
    K[0, 0] = K[1, 1] = 1266.4
    K[0, 2], K[1, 2] = 816.3, 491.5
    l2c, l2i = [], []
    for yaw in CAM_YAW_DEG:
        a = np.deg2rad(yaw)
        fwd = np.array([np.cos(a), np.sin(a), 0.0])
        right = np.array([np.sin(a), -np.cos(a), 0.0])
        down = np.array([0.0, 0.0, -1.0])
        sensor2lidar = np.eye(4)
        sensor2lidar[:3, :3] = np.stack([right, down, fwd], axis=1)  # camera axes in lidar frame
        sensor2lidar[:3, 3] = 1.5 * fwd + np.array([0.0, 0.0, -0.3])
        lidar2cam = np.linalg.inv(sensor2lidar)
        l2c.append(lidar2cam)
        l2i.append(K @ lidar2cam)
    return np.stack(l2c), np.stack([K] * len(CAM_YAW_DEG)), np.stack(l2i)


def make_sweep(seed, n_beams=32, n_azimuth=1084):
    """Raw sample as ``NuScenesDataset.get_data`` would return it (nuscenes.py:107-139): coord
    (M,3) f32 in the lidar frame, strength (M,1) in [0,1], segment (M,), camera calibration, and
    an ``img`` stand-in that only carries the image shape (pixels are never read unless
    ``fetch_color``)."""
    rng = np.random.default_rng(seed)
    boxes = _street_boxes(rng)
    elev = np.deg2rad(np.linspace(-30.67, 10.67, n_beams))
    azim = np.linspace(-np.pi, np.pi, n_azimuth, endpoint=False) + rng.uniform(0, 1e-3)
    ce, se = np.cos(elev)[:, None], np.sin(elev)[:, None]
    dirs = np.stack([ce * np.cos(azim)[None], ce * np.sin(azim)[None],
                     np.broadcast_to(se, (n_beams, n_azimuth))], -1).reshape(-1, 3)
    rng_m, cls, sid = _cast(boxes, dirs)
    hit = np.isfinite(rng_m) & (rng_m > 1.5)  # no returns from the ego vehicle
    rng_m = rng_m[hit] + rng.normal(scale=0.02, size=hit.sum())
    coord = (dirs[hit] * rng_m[:, None]).astype(np.float32)
    reflect = rng.uniform(0.05, 0.6, size=sid.max() + 2)[sid[hit]]
    strength = np.clip(reflect + rng.normal(scale=0.03, size=len(coord)), 0, 1)
    l2c, K, l2i = _camera_rig()
    img = [np.broadcast_to(np.zeros(3, dtype=np.float32), IMAGE_HW + (3,))] * len(l2c)
    return dict(coord=coord, strength=strength.reshape(-1, 1).astype(np.float32),
                segment=cls[hit].astype(np.int64), img=img, lidar2cam=l2c, cam_intrinsic=K,
                lidar2img=l2i)


def make_lidar_scene(seed, grid_size=0.1, point_nsample=512, n_azimuth=1084,
                     point_cloud_range=SCENE_RANGE):
    """One training sample after the transform chain of the nuScenes pre-training config
    (:136-200, augmentations left out): range filter -> GridSample(ravel) -> ProjectOnImage ->
    RaySample."""
    data = make_sweep(seed, n_azimuth=n_azimuth)
    state = np.random.get_state()
    np.random.seed(seed)  # the transforms draw from numpy's global generator, as the reference does
    data = PointRangeFilter(point_cloud_range=point_cloud_range, padding=0.1)(data)
    data = GridSample(grid_size=grid_size, hash_type="ravel", mode="train",
                      keys=("coord", "strength", "segment"), return_grid_coord=True)(data)
    data = ProjectOnImage(filter_overlap=True, close_radius=3.0)(data)
    data = RaySample(point_nsample=point_nsample, fetch_color=False, fetch_segment=False)(data)
    np.random.set_state(state)
    keep = ("coord", "grid_coord", "strength", "segment", "ray_start", "ray_end")
    out = {k: data[k] for k in keep}
    out["condition"] = "nuScenes"
    return out


def lidar_collate_fn(samples):
    """List of per-sweep dicts -> batch dict: point keys concatenated with cumulative ``offset``,
    ray keys concatenated with cumulative ``ray_offset`` (both also kept as host lists)."""
    cat = lambda k, dt: torch.from_numpy(np.concatenate([s[k] for s in samples]).astype(dt))  # noqa
    n_pts = np.cumsum([len(s["coord"]) for s in samples])
    n_ray = np.cumsum([len(s["ray_start"]) for s in samples])
    feat = np.concatenate([np.concatenate([s["coord"], s["strength"]], 1) for s in samples])
    return dict(coord=cat("coord", np.float32), grid_coord=cat("grid_coord", np.int64),
                feat=torch.from_numpy(feat.astype(np.float32)), segment=cat("segment", np.int64),
                ray_start=cat("ray_start", np.float32), ray_end=cat("ray_end", np.float32),
                offset=torch.tensor(n_pts, dtype=torch.int64), offset_host=[int(v) for v in n_pts],
                ray_offset=torch.tensor(n_ray, dtype=torch.int64),
                ray_offset_host=[int(v) for v in n_ray],
                sparse_shape=[int(v) + 96 for v in
                              np.max([s["grid_coord"].max(0) for s in samples], axis=0)],
                condition=[s["condition"] for s in samples])


class SyntheticLidarDataset(torch.utils.data.Dataset):
    """``length`` seeded sweeps; sweep i uses seed ``base_seed + i``."""

    collate_fn = staticmethod(lidar_collate_fn)

    def __init__(self, length=64, base_seed=0, grid_size=0.1, point_nsample=512, n_azimuth=1084,
                 point_cloud_range=SCENE_RANGE, loop=1, **kwargs):
        self.length, self.base_seed, self.loop = length, base_seed, loop
        self.kw = dict(grid_size=grid_size, point_nsample=point_nsample, n_azimuth=n_azimuth,
                       point_cloud_range=tuple(point_cloud_range))

    def __len__(self):
        return self.length * self.loop

    def __getitem__(self, idx):
        return make_lidar_scene(self.base_seed + idx % self.length, **self.kw)
