"""On-disk dataset readers of the pre-training configs (SURVEY.md Appendix B formats), under the
reference's registry names so ``data.train`` sections build unchanged:

  ScanNetRGBDDataset       ponder/datasets/scannet.py:213-600   scene ``.pth`` + RGB-D frame folders
  Structured3DRGBDDataset  ponder/datasets/structure3d.py:42-148  room ``.pth`` + ``<room>_rgbd/*.pth``
  S3DISRGBDDataset         ponder/datasets/s3dis.py:158-290     same layout, depth in 1/4000 m
  NuScenesDataset          ponder/datasets/nuscenes.py:13-200   info pickle + lidar ``.bin`` + 6 cameras

Each ``get_data`` returns the raw sample dict the transform chain expects (coord/color/normal/
segment + intrinsic, extrinsic = world->camera, rgb, depth, depth_scale[, semantic]); the same
``np.random`` draws pick the frames.  Images are decoded with Pillow (OpenCV is not a dependency
here); the two resizes the ScanNet reader needs follow ``cv2.resize``'s sampling conventions.
"""
import glob
import json
import os
import pickle
from collections import defaultdict
from collections.abc import Sequence

import numpy as np
import torch

from .collate import point_collate_fn
from .transform import Compose

# nyu40 ids of the 20 ScanNet benchmark classes, in benchmark order
VALID_CLASS_IDS_20 = (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 24, 28, 33, 34, 36, 39)


def _load(path):
    return torch.load(path, weights_only=False)  # dicts of numpy arrays


def read_image(path):
    """Image file -> numpy array (H,W,3) uint8 RGB, or (H,W) for single-channel / 16-bit files."""
    from PIL import Image

    with Image.open(path) as im:
        if im.mode in ("I;16", "I;16B", "I", "L", "P", "1", "F"):
            return np.array(im)
        return np.array(im.convert("RGB"))


def resize_nearest(img, height, width):
    """cv2.resize(..., INTER_NEAREST): source index = floor(dst * src/dst)."""
    ys = np.minimum((np.arange(height) * (img.shape[0] / height)).astype(np.int64), img.shape[0] - 1)
    xs = np.minimum((np.arange(width) * (img.shape[1] / width)).astype(np.int64), img.shape[1] - 1)
    return img[ys][:, xs]


def resize_bilinear(img, height, width):
    """cv2.resize(..., INTER_LINEAR) sampling (pixel centres, replicated border), computed in
    float and rounded for integer images (OpenCV's fixed-point path may differ by one level)."""
    if img.shape[0] == height and img.shape[1] == width:
        return img

    def taps(n_dst, n_src):
        pos = (np.arange(n_dst) + 0.5) * (n_src / n_dst) - 0.5
        lo = np.floor(pos)
        frac = pos - lo
        lo = lo.astype(np.int64)
        return np.clip(lo, 0, n_src - 1), np.clip(lo + 1, 0, n_src - 1), frac

    y0, y1, fy = taps(height, img.shape[0])
    x0, x1, fx = taps(width, img.shape[1])
    src = img.astype(np.float64)
    fy = fy.reshape(-1, *([1] * (src.ndim - 1)))
    rows = src[y0] * (1 - fy) + src[y1] * fy
    fx = fx.reshape(1, -1, *([1] * (src.ndim - 2)))
    out = rows[:, x0] * (1 - fx) + rows[:, x1] * fx
    if np.issubdtype(img.dtype, np.integer):
        out = np.clip(np.rint(out), np.iinfo(img.dtype).min, np.iinfo(img.dtype).max)
    return out.astype(img.dtype)


class PointDataset(torch.utils.data.Dataset):
    """Training-mode skeleton shared by the readers: ``data_list`` x ``loop`` samples, every
    sample = transform(get_data(idx))."""

    collate_fn = staticmethod(point_collate_fn)

    def __init__(self, split="train", data_root="data/dataset", transform=None, test_mode=False,
                 test_cfg=None, loop=1):
        if test_mode:
            raise NotImplementedError("test-time fragment voxelisation is outside the pre-training path")
        self.split, self.data_root, self.loop = split, data_root, loop
        self.transform = Compose(transform)
        self.data_list = self.get_data_list()

    def _glob_splits(self, pattern):
        splits = [self.split] if isinstance(self.split, str) else list(self.split)
        if not isinstance(self.split, (str, Sequence)):
            raise NotImplementedError
        found = []
        for split in splits:
            found += glob.glob(os.path.join(self.data_root, split, pattern))
        return found

    def get_data_list(self):
        return self._glob_splits("*.pth")

    def get_data_name(self, idx):
        return os.path.basename(self.data_list[idx % len(self.data_list)]).split(".")[0]

    def get_data(self, idx):
        raise NotImplementedError

    def __getitem__(self, idx):
        return self.transform(self.get_data(idx))

    def __len__(self):
        return len(self.data_list) * self.loop


def _segment_of(data, key, n):
    return data[key].reshape([-1]) if key in data else np.ones(n) * -1


class _FramePthRGBDDataset(PointDataset):
    """A room is ``<room>.pth`` (points) next to ``<room>_rgbd/*.pth`` (one dict per rendered
    frame: rgb, depth, depth_mask, intrinsic, camera->world extrinsic, semantic_map)."""

    pattern = "*.pth"
    depth_scale = 1.0 / 1000.0

    def __init__(self, num_cameras=5, render_semantic=True, **kwargs):
        self.num_cameras, self.render_semantic = num_cameras, render_semantic
        super().__init__(**kwargs)

    @staticmethod
    def _frame_paths(room_path):
        return glob.glob(os.path.join(room_path.split(".pth")[0] + "_rgbd", "*.pth"))

    def get_data_list(self):
        rooms = self._glob_splits(self.pattern)
        return [r for r in rooms if len(self._frame_paths(r)) > 0]  # rooms without frames are unusable

    def _points(self, data, room_path):
        n = data["coord"].shape[0]
        return dict(coord=data["coord"], normal=data["normal"], color=data["color"],
                    segment=_segment_of(data, "semantic_gt", n))

    def get_data(self, idx):
        room_path = self.data_list[idx % len(self.data_list)]
        data = _load(room_path)
        frames = self._frame_paths(room_path)
        if len(frames) <= 0:
            print(f"{room_path} has no rgbd data.")
            return self.get_data(np.random.randint(0, len(self)))
        frames = np.random.choice(frames, self.num_cameras, replace=self.num_cameras > len(frames))
        views = [_load(p) for p in frames]
        for path, view in zip(frames, views):
            if view["depth_mask"].mean() < 0.25:  # mostly invalid depth: retire the frame, redraw
                os.rename(path, path + ".bad")
                return self.get_data(idx)
        sample = self._points(data, room_path)
        sample.update(
            intrinsic=np.stack([v["intrinsic"] for v in views], axis=0).astype(np.float32),
            extrinsic=np.stack([np.linalg.inv(v["extrinsic"]) for v in views], axis=0).astype(np.float32),
            rgb=np.stack([v["rgb"].astype(np.float32) for v in views], axis=0),
            depth=np.stack([v["depth"].astype(np.float32) * v["depth_mask"].astype(np.float32)
                            * (v["depth"] < 65535).astype(np.float32) for v in views], axis=0),
            depth_scale=self.depth_scale)
        if self.render_semantic:
            maps = []
            for v in views:
                m = v["semantic_map"]
                m[m <= 0] = -1
                m[m > 40] = -1
                maps.append(m.astype(np.int16))
            sample["semantic"] = np.stack(maps, axis=0)
        return sample


class Structured3DRGBDDataset(_FramePthRGBDDataset):
    pattern = "*/*.pth"  # <split>/<scene>/<room>.pth

    def __init__(self, split="train", data_root="data/dataset", transform=None, test_mode=False,
                 test_cfg=None, num_cameras=5, render_semantic=True, loop=1):
        super().__init__(num_cameras=num_cameras, render_semantic=render_semantic, split=split,
                         data_root=data_root, transform=transform, test_mode=test_mode,
                         test_cfg=test_cfg, loop=loop)

    def get_data_name(self, idx):
        scene_dir, file_name = os.path.split(self.data_list[idx % len(self.data_list)])
        return f"{os.path.basename(scene_dir)}_{os.path.splitext(file_name)[0]}"


class S3DISRGBDDataset(_FramePthRGBDDataset):
    depth_scale = 1.0 / 4000.0

    def __init__(self, split=("Area_1", "Area_2", "Area_3", "Area_4", "Area_6"),
                 data_root="data/s3dis", transform=None, test_mode=False, test_cfg=None,
                 cache=False, num_cameras=5, render_semantic=True, six_fold=False, loop=1):
        if cache:
            raise NotImplementedError("shared-memory caching is not implemented")
        self.six_fold = six_fold
        super().__init__(num_cameras=num_cameras, render_semantic=render_semantic, split=split,
                         data_root=data_root, transform=transform, test_mode=test_mode,
                         test_cfg=test_cfg, loop=loop)

    def _points(self, data, room_path):
        n = data["coord"].shape[0]
        out = dict(name=os.path.basename(room_path).split("_")[0].replace("R", " r"),
                   coord=data["coord"], color=data["color"],
                   segment=_segment_of(data, "semantic_gt", n),
                   instance=_segment_of(data, "instance_gt", n), scene_id=room_path)
        if "normal" in data:
            out["normal"] = data["normal"]
        return out


class ScanNetRGBDDataset(PointDataset):
    """Scene points from ``<data_root>/<split>/<scene>.pth``; frames from
    ``<rgbd_root>/<scene>/{color/*.jpg, depth/*.png (mm), pose/*.txt (camera->world),
    label/*.png (nyu40), intrinsic/intrinsic_depth.txt}``."""

    def __init__(self, split="train", data_root="data/scannet", rgbd_root="data/scannet/rgbd",
                 transform=None, lr_file=None, la_file=None, ignore_index=-1, test_mode=False,
                 test_cfg=None, cache=False, frame_interval=10, nearby_num=2, nearby_interval=20,
                 num_cameras=5, render_semantic=True, align_axis=False, loop=1):
        if cache:
            raise NotImplementedError("shared-memory caching is not implemented")
        self.rgbd_root = rgbd_root
        self.frame_interval, self.nearby_num = frame_interval, nearby_num
        self.nearby_interval, self.num_cameras = nearby_interval, num_cameras
        self.render_semantic, self.align_axis = render_semantic, align_axis
        self.ignore_index = ignore_index
        self._frames, self._intrinsics, self._alignments = {}, {}, {}
        super().__init__(split=split, data_root=data_root, transform=transform,
                         test_mode=test_mode, test_cfg=test_cfg, loop=loop)
        if lr_file:  # limited reconstructions: keep the listed scenes only
            keep = set(np.loadtxt(lr_file, dtype=str).reshape(-1).tolist())
            self.data_list = [d for d in self.data_list if d["scene"] in keep]
        self.la = _load(la_file) if la_file else None  # limited annotations: scene -> point ids

    # ------------------------------------------------------------------ frame index
    def get_frame_list(self, scene):
        if scene not in self._frames:
            folder = os.path.join(self.rgbd_root, scene, "color")
            if not os.path.exists(folder):
                return []
            names = [f for f in os.listdir(folder) if f.endswith(".jpg")]
            names.sort(key=lambda f: int(f.split(".")[0]))
            self._frames[scene] = names
        return self._frames[scene]

    def get_data_list(self):
        """[{scene, frame: [positions in the scene's sorted frame list]}]; the per-split index is
        cached as ``<data_root>/<split>.json`` like the reference does."""
        index_file = os.path.join(self.data_root, self.split + ".json")
        if os.path.exists(index_file):
            with open(index_file) as f:
                flat = json.load(f)
        else:
            skip = set()
            skip_file = os.path.join(self.data_root, "skip.lst")
            if os.path.exists(skip_file):
                with open(skip_file) as f:
                    for line in f.read().split("\n"):
                        if line:
                            scene, frame = line.split()
                            skip.add((scene, int(frame)))
            flat = []
            scenes = [f.split(".")[0] for f in os.listdir(os.path.join(self.data_root, self.split))]
            for scene in scenes:
                frames = self.get_frame_list(scene)
                if self.split in ("val", "test"):
                    frames = frames[::10]
                margin = self.nearby_num * self.nearby_interval
                for name in frames[margin:-(margin + self.nearby_interval):self.frame_interval]:
                    frame = int(name.split(".")[0])
                    if (scene, frame) not in skip:
                        flat.append({"scene": scene, "frame": frame})
            with open(index_file, "w") as f:
                json.dump(flat, f)
        per_scene = defaultdict(list)
        for item in flat:
            per_scene[item["scene"]].append(item["frame"])
        return [{"scene": s, "frame": fr} for s, fr in per_scene.items()]

    # ------------------------------------------------------------------ per-scene metadata
    def get_intrinsic(self, scene):
        if scene not in self._intrinsics:
            self._intrinsics[scene] = np.loadtxt(
                os.path.join(self.rgbd_root, scene, "intrinsic", "intrinsic_depth.txt"))
        return self._intrinsics[scene]

    def get_axis_align_matrix(self, scene):
        if scene not in self._alignments:
            with open(os.path.join(self.rgbd_root, scene, f"{scene}.txt")) as f:
                for line in f:
                    if "axisAlignment" in line:
                        values = [float(x) for x in line.split("=")[1].split()]
                        self._alignments[scene] = np.array(values).reshape(4, 4)
                        break
        return self._alignments[scene]

    def read_frame(self, scene, name):
        folder = os.path.join(self.rgbd_root, scene)
        stem = name[:-len(".jpg")]
        rgb = read_image(os.path.join(folder, "color", name))
        depth = read_image(os.path.join(folder, "depth", stem + ".png"))
        pose = np.loadtxt(os.path.join(folder, "pose", stem + ".txt"))
        label = read_image(os.path.join(folder, "label", stem + ".png")) if self.render_semantic else None
        return rgb, depth, pose, label

    def get_2d_meta(self, scene, frame_pos):
        """-> intrinsic (4,4), world->camera rotation, translation, rgb (H,W,3) at depth
        resolution, depth (H,W) float32 [mm][, semantic (H,W) int16 in 0..19 / -1]."""
        rgb, depth, pose, label = self.read_frame(scene, self.get_frame_list(scene)[frame_pos])
        h, w = depth.shape[:2]
        rgb = resize_bilinear(rgb, h, w)
        if self.align_axis:
            pose = self.get_axis_align_matrix(scene) @ pose
        world2cam = np.linalg.inv(pose)
        out = [np.array(self.get_intrinsic(scene)), world2cam[:3, :3], world2cam[:3, 3], rgb,
               depth.astype(np.float32)]
        if self.render_semantic:
            nyu40 = resize_nearest(label, h, w).astype(np.int16)
            semantic = np.zeros_like(nyu40) - 1
            for i, class_id in enumerate(VALID_CLASS_IDS_20):
                semantic[nyu40 == class_id] = i
            out.append(semantic)
        return out

    # ------------------------------------------------------------------ sample
    def get_data(self, idx):
        entry = self.data_list[idx % len(self.data_list)]
        scene, frame_list = entry["scene"], entry["frame"]
        scene_path = os.path.join(self.data_root, self.split, f"{scene}.pth")
        data = _load(scene_path)
        if self.num_cameras > len(frame_list):
            print(f"Warning: {scene} has only {len(frame_list)} frames, "
                  f"but {self.num_cameras} cameras are required.")
        chosen = np.random.choice(frame_list, self.num_cameras,
                                  replace=self.num_cameras > len(frame_list))
        intrinsic, extrinsic, rgb, depth, semantic = [], [], [], [], []
        for frame_pos in chosen:
            meta = self.get_2d_meta(scene, frame_pos)
            intrinsic.append(meta[0])
            world2cam = np.eye(4)
            world2cam[:3, :3], world2cam[:3, 3] = meta[1], meta[2]
            extrinsic.append(world2cam)
            rgb.append(meta[3])
            depth.append(meta[4])
            if self.render_semantic:
                assert meta[5].max() <= 20, meta[5]
                semantic.append(meta[5])
        n = data["coord"].shape[0]
        sample = dict(coord=data["coord"], normal=data["normal"], color=data["color"],
                      segment=_segment_of(data, "semantic_gt20", n),
                      instance=_segment_of(data, "instance_gt", n), scene_id=data["scene_id"],
                      intrinsic=np.stack(intrinsic, axis=0), extrinsic=np.stack(extrinsic, axis=0),
                      rgb=np.stack(rgb, axis=0), depth=np.stack(depth, axis=0),
                      depth_scale=1.0 / 1000.0, id=f"{scene}/{chosen[0]}")
        if self.render_semantic:
            sample["semantic"] = np.stack(semantic, axis=0)
        if self.la:  # keep only the sampled annotations
            sampled = self.la[os.path.basename(scene_path).split(".")[0]]
            hidden = np.ones_like(sample["segment"]).astype(bool)
            hidden[sampled] = False
            sample["segment"][hidden] = self.ignore_index
            sample["sampled_index"] = sampled
            sample["semantic"] = np.zeros_like(sample["semantic"]) - 1
        return sample


class NuScenesDataset(PointDataset):
    """``info/nuscenes_infos_<N>sweeps_<split>.pkl`` lists the key frames; a frame is one lidar
    ``.bin`` (x, y, z, intensity, ring as float32) and, with ``use_camera``, six calibrated images."""

    # nuScenes-lidarseg label -> the 16 benchmark classes (everything else is ignored)
    LIDARSEG_TO_16 = {2: 6, 3: 6, 4: 6, 6: 6, 9: 0, 12: 7, 14: 1, 15: 2, 16: 2, 17: 3, 18: 4, 21: 5,
                      22: 8, 23: 9, 24: 10, 25: 11, 26: 12, 27: 13, 28: 14, 30: 15}

    def __init__(self, split="train", data_root="data/nuscenes", sweeps=10, use_camera=False,
                 transform=None, test_mode=False, test_cfg=None, loop=1, ignore_index=-1):
        self.sweeps, self.ignore_index, self.use_camera = sweeps, ignore_index, use_camera
        self.learning_map = self.get_learning_map(ignore_index)
        super().__init__(split=split, data_root=data_root, transform=transform,
                         test_mode=test_mode, test_cfg=test_cfg, loop=loop)

    @classmethod
    def get_learning_map(cls, ignore_index):
        return {raw: cls.LIDARSEG_TO_16.get(raw, ignore_index) for raw in range(32)}

    def get_info_path(self, split):
        assert split in ("train", "val", "test")
        return os.path.join(self.data_root, "info", f"nuscenes_infos_{self.sweeps}sweeps_{split}.pkl")

    def get_data_list(self):
        splits = [self.split] if isinstance(self.split, str) else list(self.split)
        frames = []
        for split in splits:
            with open(self.get_info_path(split), "rb") as f:
                frames.extend(pickle.load(f))
        return frames

    def get_camera_data(self, frame):
        img, shape, lidar2img, lidar2cam, intrinsic = [], [], [], [], []
        for cam in frame["cams"].values():
            pixels = read_image(os.path.join(self.data_root, "raw", cam["data_path"])).astype(np.float32)
            img.append(pixels)
            shape.append(pixels.shape)
            to_cam = np.linalg.inv(cam["sensor2lidar"])
            K = np.eye(4)
            K[:3, :3] = cam["cam_intrinsic"]
            lidar2cam.append(to_cam)
            intrinsic.append(K)
            lidar2img.append(K @ to_cam)
        return dict(img=np.stack(img, axis=0), ori_shape=np.stack(shape, axis=0),
                    lidar2img=np.stack(lidar2img, axis=0), lidar2cam=np.stack(lidar2cam, axis=0),
                    cam_intrinsic=np.stack(intrinsic, axis=0))

    def get_data(self, idx):
        frame = self.data_list[idx % len(self.data_list)]
        points = np.fromfile(os.path.join(self.data_root, "raw", frame["lidar_path"]),
                             dtype=np.float32, count=-1).reshape([-1, 5])
        if "gt_segment_path" in frame:
            raw = np.fromfile(os.path.join(self.data_root, "raw", frame["gt_segment_path"]),
                              dtype=np.uint8, count=-1).reshape([-1])
            segment = np.vectorize(self.learning_map.__getitem__)(raw).astype(np.int64)
        else:
            segment = np.ones((points.shape[0],), dtype=np.int64) * self.ignore_index
        sample = dict(coord=points[:, :3], strength=points[:, 3].reshape([-1, 1]) / 255,
                      segment=segment)
        if self.use_camera:
            sample.update(self.get_camera_data(frame))
        sample["lidar_token"] = frame["lidar_token"]
        return sample

    def get_data_name(self, idx):
        return self.data_list[idx % len(self.data_list)]["lidar_token"]
