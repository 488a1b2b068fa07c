"""On-disk readers (datasets/readers.py) on miniature dataset trees written into tmp_path, and the
path from disk to one training step of the host model code (CPU doubles for the kernels)."""
import json
import os
import pickle
import random

import numpy as np
import pytest
import torch

from test_transforms import SCANNET_CHAIN


def _write_scannet_tree(root, scene="scene0000_00", n_frames=110, hw=(12, 16), color_hw=(24, 32)):
    from PIL import Image

    from ponderv2_amd.ponder.datasets import make_scene

    s = make_scene(5, n_raw=6000, keep=1.0, num_views=1, image_hw=hw)
    os.makedirs(os.path.join(root, "train"), exist_ok=True)
    torch.save(dict(coord=s["coord"], color=s["color"], normal=s["normal"], scene_id=scene,
                    semantic_gt20=s["segment"].astype(np.int16),
                    instance_gt=np.arange(len(s["coord"])) % 7),
               os.path.join(root, "train", f"{scene}.pth"))
    rgbd = os.path.join(root, "rgbd", scene)
    for sub in ("color", "depth", "pose", "label", "intrinsic"):
        os.makedirs(os.path.join(rgbd, sub), exist_ok=True)
    K = np.eye(4)
    K[0, 0] = K[1, 1] = 14.0
    K[0, 2], K[1, 2] = 7.5, 5.5
    np.savetxt(os.path.join(rgbd, "intrinsic", "intrinsic_depth.txt"), K)
    rng = np.random.default_rng(0)
    poses = {}
    for i in range(n_frames):
        Image.fromarray(rng.integers(0, 255, (*color_hw, 3), dtype=np.uint8)).save(
            os.path.join(rgbd, "color", f"{i}.jpg"))
        depth = rng.integers(500, 4000, hw).astype(np.uint16)
        Image.fromarray(depth).save(os.path.join(rgbd, "depth", f"{i}.png"))
        label = rng.choice([0, 1, 2, 5, 13, 39, 40], size=color_hw).astype(np.uint8)
        Image.fromarray(label).save(os.path.join(rgbd, "label", f"{i}.png"))
        pose = np.eye(4)
        a = 0.1 * i
        pose[:3, :3] = [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]
        pose[:3, 3] = [1.0 + 0.01 * i, 2.0, 1.5]
        poses[i] = pose
        np.savetxt(os.path.join(rgbd, "pose", f"{i}.txt"), pose)
    return poses, K


def test_scannet_reader_contract(tmp_path):
    from ponderv2_amd.ponder.datasets import ScanNetRGBDDataset
    from ponderv2_amd.ponder.datasets.readers import read_image

    root = str(tmp_path / "scannet")
    poses, K = _write_scannet_tree(root)
    ds = ScanNetRGBDDataset(split="train", data_root=root, rgbd_root=os.path.join(root, "rgbd"),
                            num_cameras=3, transform=[], loop=2)
    # frames [nearby_num*interval : -(nearby_num+1)*interval : frame_interval] = [40:-60:10] of 110
    assert ds.data_list == [{"scene": "scene0000_00", "frame": [40]}] and len(ds) == 2
    assert json.load(open(os.path.join(root, "train.json"))) == [{"scene": "scene0000_00", "frame": 40}]
    np.random.seed(0)
    d = ds.get_data(0)
    assert d["coord"].shape[1] == 3 and d["segment"].shape == (len(d["coord"]),)
    assert d["rgb"].shape == (3, 12, 16, 3) and d["rgb"].dtype == np.uint8   # resized to depth size
    assert d["depth"].shape == (3, 12, 16) and d["depth"].dtype == np.float32
    assert d["semantic"].shape == (3, 12, 16) and d["semantic"].dtype == np.int16
    assert set(np.unique(d["semantic"])) <= {-1, 0, 1, 4, 11, 19}  # nyu40 1,2,5,13,39 -> 0,1,4,11,19
    assert d["depth_scale"] == 1.0 / 1000.0 and d["id"] == "scene0000_00/40"
    assert np.allclose(d["intrinsic"][0], K)
    assert np.allclose(d["extrinsic"][0], np.linalg.inv(poses[40]))  # world -> camera
    depth_file = read_image(os.path.join(root, "rgbd", "scene0000_00", "depth", "40.png"))
    assert np.array_equal(d["depth"][0], depth_file.astype(np.float32))


def test_resize_conventions():
    from ponderv2_amd.ponder.datasets.readers import resize_bilinear, resize_nearest

    img = np.arange(24, dtype=np.uint8).reshape(4, 6)
    assert np.array_equal(resize_nearest(img, 2, 3), img[::2, ::2])       # floor(dst * 2)
    half = resize_bilinear(img.astype(np.float32), 2, 3)                  # centres fall between 4 pixels
    assert np.allclose(half, [[3.5, 5.5, 7.5], [15.5, 17.5, 19.5]])
    up = resize_bilinear(np.array([[0.0, 10.0]]), 1, 4)                   # replicated border
    assert np.allclose(up, [[0.0, 2.5, 7.5, 10.0]])
    assert resize_bilinear(img, 4, 6) is img


def _write_frame_tree(root, rel_room, n_frames=3, bad=()):
    from ponderv2_amd.ponder.datasets import make_scene

    s = make_scene(9, n_raw=5000, keep=1.0, num_views=n_frames, image_hw=(12, 16))
    room = os.path.join(root, rel_room)
    os.makedirs(os.path.dirname(room), exist_ok=True)
    torch.save(dict(coord=s["coord"], color=s["color"], normal=s["normal"],
                    semantic_gt=s["segment"].astype(np.int16)), room)
    os.makedirs(room[:-4] + "_rgbd", exist_ok=True)
    for i in range(n_frames):
        mask = np.ones((12, 16), bool)
        if i in bad:
            mask[:] = False
        depth = s["depth"][i].astype(np.uint16)
        depth[0, 0] = 65535
        sem = (s["semantic"][i] + 1).astype(np.int32)
        sem[0, 1] = 41
        torch.save(dict(rgb=(s["rgb"][i] * 255).astype(np.uint8), depth=depth, depth_mask=mask,
                        intrinsic=s["intrinsic"][i], extrinsic=np.linalg.inv(s["extrinsic"][i]),
                        semantic_map=sem), os.path.join(room[:-4] + "_rgbd", f"{i}.pth"))
    return s


def test_structured3d_and_s3dis_readers(tmp_path):
    from ponderv2_amd.ponder.datasets import S3DISRGBDDataset, Structured3DRGBDDataset

    root = str(tmp_path / "s3d")
    s = _write_frame_tree(root, "train/scene_00000/room_1.pth")
    os.makedirs(os.path.join(root, "train", "scene_00001"), exist_ok=True)
    torch.save(dict(coord=s["coord"], color=s["color"], normal=s["normal"]),
               os.path.join(root, "train", "scene_00001", "room_2.pth"))   # no frames: filtered out
    ds = Structured3DRGBDDataset(split="train", data_root=root, num_cameras=2, transform=[])
    assert len(ds) == 1 and ds.get_data_name(0) == "scene_00000_room_1"
    np.random.seed(1)
    d = ds.get_data(0)
    assert d["rgb"].shape == (2, 12, 16, 3) and d["rgb"].dtype == np.float32
    assert d["extrinsic"].dtype == np.float32 and d["depth_scale"] == 1.0 / 1000.0
    assert (d["depth"][:, 0, 0] == 0).all()            # 65535 = invalid depth
    assert (d["semantic"][:, 0, 1] == -1).all() and d["semantic"].dtype == np.int16  # > 40 -> ignore
    assert "instance" not in d

    root2 = str(tmp_path / "s3dis")
    _write_frame_tree(root2, "Area_1/conferenceRoom_1.pth", n_frames=3, bad=(1,))
    ds2 = S3DISRGBDDataset(split=("Area_1",), data_root=root2, num_cameras=2, transform=[])
    np.random.seed(0)
    d2 = ds2.get_data(0)
    assert d2["depth_scale"] == 1.0 / 4000.0 and d2["name"] == "conference room"
    # the frame with an empty depth mask was retired on the way
    assert os.path.exists(os.path.join(root2, "Area_1", "conferenceRoom_1_rgbd", "1.pth.bad"))
    assert d2["rgb"].shape[0] == 2 and (d2["instance"] == -1).all()


def test_nuscenes_reader_and_outdoor_chain(tmp_path):
    """info pickle + lidar .bin + six camera images -> the nuScenes transform chain -> collate."""
    from PIL import Image

    from ponderv2_amd.ponder.datasets import NuScenesDataset, make_sweep

    root = str(tmp_path / "nuscenes")
    os.makedirs(os.path.join(root, "raw", "lidar"), exist_ok=True)
    os.makedirs(os.path.join(root, "raw", "cam"), exist_ok=True)
    os.makedirs(os.path.join(root, "info"), exist_ok=True)
    infos = []
    for f in range(2):
        sweep = make_sweep(20 + f, n_azimuth=200)
        pts = np.concatenate([sweep["coord"], sweep["strength"] * 255,
                              np.zeros((len(sweep["coord"]), 1), np.float32)], 1).astype(np.float32)
        pts.tofile(os.path.join(root, "raw", "lidar", f"{f}.bin"))
        seg = np.where(sweep["segment"] == 3, 17, 24).astype(np.uint8)   # car / driveable surface
        seg.tofile(os.path.join(root, "raw", "lidar", f"{f}.seg"))
        cams = {}
        for c in range(6):
            Image.fromarray(np.full((90, 160, 3), 40 * c, np.uint8)).save(
                os.path.join(root, "raw", "cam", f"{f}_{c}.jpg"))
            K = sweep["cam_intrinsic"][c][:3, :3].copy()
            K[:2] /= 10.0  # images are 10x smaller than nuScenes' 900x1600
            cams[f"CAM_{c}"] = dict(data_path=f"cam/{f}_{c}.jpg", cam_intrinsic=K,
                                    sensor2lidar=np.linalg.inv(sweep["lidar2cam"][c]))
        infos.append(dict(lidar_path=f"lidar/{f}.bin", gt_segment_path=f"lidar/{f}.seg", cams=cams,
                          lidar_token=f"tok{f}"))
    with open(os.path.join(root, "info", "nuscenes_infos_10sweeps_train.pkl"), "wb") as fh:
        pickle.dump(infos, fh)
    chain = [
        dict(type="RandomRotate", angle=[-0.25, 0.25], axis="z", center=[0, 0, 0], p=0.5,
             keys=["lidar2img", "lidar2cam"]),
        dict(type="RandomFlip", p=0.5, keys=["lidar2img", "lidar2cam"]),
        dict(type="PointRangeFilter", point_cloud_range=(-54.0, -54.0, -5.0, 54.0, 54.0, 3.0), padding=0.1),
        dict(type="GridSample", grid_size=0.1, hash_type="ravel", mode="train",
             keys=("coord", "strength", "segment"), return_grid_coord=True),
        dict(type="ProjectOnImage", filter_overlap=True, close_radius=3.0),
        dict(type="RaySample", point_nsample=16, fetch_color=True, fetch_segment=True),
        dict(type="Add", keys_dict={"condition": "nuScenes"}),
        dict(type="ToTensor"),
        dict(type="Collect", keys=("coord", "grid_coord", "segment", "condition", "ray_start",
                                   "ray_end", "ray_segment", "ray_color"),
             offset_keys_dict=dict(offset="coord", ray_offset="ray_start"),
             stack_keys=("lidar2img", "lidar2cam", "cam_intrinsic"), feat_keys=("coord", "strength")),
    ]
    ds = NuScenesDataset(split="train", data_root=root, use_camera=True, transform=chain)
    assert len(ds) == 2 and ds.get_data_name(1) == "tok1"
    raw = ds.get_data(0)
    assert raw["img"].shape == (6, 90, 160, 3) and raw["lidar2img"].shape == (6, 4, 4)
    assert set(np.unique(raw["segment"])) <= {3, 10} and raw["strength"].max() <= 1.0
    random.seed(0)
    np.random.seed(0)
    batch = ds.collate_fn([ds[0], ds[1]])
    n, r = int(batch["offset"][-1]), int(batch["ray_offset"][-1])
    assert batch["feat"].shape == (n, 4) and batch["ray_start"].shape == (r, 3)
    assert batch["condition"] == ["nuScenes", "nuScenes"] and batch["lidar2img"].shape == (2, 6, 4, 4)
    assert batch["ray_color"].shape == (r, 3) and r > 0


def test_scannet_disk_to_training_step(tmp_path, monkeypatch):
    """Reader -> reference transform chain -> point_collate_fn -> PonderIndoor (tiny) forward +
    backward on the host with the kernel doubles: the on-disk contract feeds the model."""
    import golden_cases as gc
    from oracle import cpu_backend
    from ponderv2_amd.ponder.datasets import ScanNetRGBDDataset
    from ponderv2_amd.ponder.models import build_model
    from ponderv2_amd.ponder.utils.config import ConfigDict

    root = str(tmp_path / "scannet")
    _write_scannet_tree(root, hw=(24, 32), color_hw=(24, 32))
    chain = [dict(t) for t in SCANNET_CHAIN]
    chain[1] = dict(type="RandomDropout", dropout_ratio=0.2, dropout_application_ratio=1.0)
    ds = ScanNetRGBDDataset(split="train", data_root=root, rgbd_root=os.path.join(root, "rgbd"),
                            num_cameras=2, transform=chain)
    random.seed(0)
    np.random.seed(0)
    batch = ds.collate_fn([ds[0], ds[0]])
    assert batch["rgb"].shape == (2, 2, 24, 32, 3) and batch["condition"] == ["ScanNet", "ScanNet"]
    cpu_backend.install(monkeypatch)
    cfg = gc.indoor_model_cfg(dict(gc.SMALL_BACKBONE, channels=(16, 32, 48, 64, 64, 48, 32, 96)),
                              grid_shape=(32, 32, 8), ray_nsample=8)
    model = build_model(ConfigDict(cfg)).train()
    torch.manual_seed(0)
    out = model(batch)
    out["loss"].backward()
    assert torch.isfinite(out["loss"]) and model.backbone.conv_input[0].weight.grad is not None


@pytest.mark.skipif(not os.path.isdir("/root/reference/configs"), reason="reference checkout not present")
def test_reference_scannet_config_builds_its_dataset_and_model(tmp_path):
    """The reference's own pre-training config, data paths pointed at a miniature tree: dataset,
    transform chain and model all build from the unchanged sections and produce a batch."""
    from ponderv2_amd.ponder.engines.train import build_dataset
    from ponderv2_amd.ponder.utils.config import Config

    root = str(tmp_path / "scannet")
    _write_scannet_tree(root, hw=(24, 32), color_hw=(24, 32))
    cfg = Config.fromfile("/root/reference/configs/scannet/pretrain-ponder-spunet-v1m1-0-base.py")
    train = cfg.data.train.to_dict() if hasattr(cfg.data.train, "to_dict") else dict(cfg.data.train)
    train.update(data_root=root, rgbd_root=os.path.join(root, "rgbd"))
    ds = build_dataset(train)
    assert type(ds).__name__ == "ScanNetRGBDDataset" and ds.num_cameras == 5
    assert [type(t).__name__ for t in ds.transform.transforms][:3] == ["CenterShift", "RandomDropout",
                                                                     "RandomRotate"]
    random.seed(1)
    np.random.seed(1)
    batch = ds.collate_fn([ds[0]])
    assert batch["feat"].shape[1] == 6 and batch["rgb"].shape[:2] == (1, 5)
    assert batch["grid_coord"].dtype == torch.int64 and batch["condition"] == ["ScanNet"]
