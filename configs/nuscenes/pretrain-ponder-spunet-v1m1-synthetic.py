# PonderV2 outdoor pre-training on SYNTHETIC nuScenes-shaped lidar sweeps (no dataset in this
# environment).  The model / optimiser sections carry the hyper-parameters of the reference's
# configs/nuscenes/pretrain-ponder-spunet-v1m1-0-base.py; only the data section differs (a seeded
# sweep generator replaces NuScenesDataset; the transform chain range-filter -> GridSample(ravel,
# 0.1 m) -> ProjectOnImage -> RaySample(512 per camera) is applied inside the dataset).
_base_ = ["../_base_/default_runtime.py"]

batch_size = 4          # total over all GPUs (the reference: 4 per GPU)
num_worker = 2
mix_prob = 0
enable_amp = False      # fp32: the parity configuration
find_unused_parameters = True   # laplace beta gets no gradient
epoch = 4
eval_epoch = 4

train = dict(type="MultiDatasetTrainer")

CLASSES = ("barrier", "bicycle", "bus", "car", "construction vehicle", "motorcycle", "pedestrian",
           "traffic cone", "trailer", "truck", "path suitable or safe for driving", "other flat",
           "sidewalk", "terrain", "man made", "vegetation")

model = dict(
    type="PonderOutdoor-v2",
    mask=dict(ratio=0.8, size=8, channel=4),
    backbone=dict(type="SpUNet-v1m1", in_channels=4, num_classes=0,
                  channels=(32, 64, 128, 256, 256, 128, 96, 96), layers=(2, 3, 4, 6, 2, 2, 2, 2)),
    projection=dict(type="SimpleConv3D-v1m1", in_channels=96, out_channels=32),
    renderer=dict(
        type="NeuSModel",
        field=dict(type="SDFField",
                   sdf_decoder=dict(in_dim=32, out_dim=16 + 1, hidden_size=16, n_blocks=5),
                   beta_init=0.3, use_gradient=True, volume_type="default", padding_mode="zeros",
                   share_volume=True),
        collider=dict(type="AABBBoxCollider", near_plane=0.01, bbox=[0.0, 0.0, 0.0, 1.0, 1.0, 1.0]),
        sampler=dict(type="NeuSSampler", initial_sampler="UniformSampler", num_samples=72,
                     num_samples_importance=24, num_upsample_steps=1, train_stratified=True,
                     single_jitter=False),
        loss=dict(sensor_depth_truncation=0.01, weights=dict(depth_loss=10.0))),
    scene_bbox=((-54.0, -54.0, -5.0, 54.0, 54.0, 3.0),),
    grid_shape=((180, 180, 5),),
    grid_size=((0.6, 0.6, 1.6),),
    val_ray_split=8192, pool_type="mean", share_volume=True, render_semantic=False,
    conditions=("nuScenes",), template="[x]", clip_model="ViT-B/16", class_name=CLASSES,
    valid_index=(tuple(range(16)),))

optimizer = dict(type="AdamW", lr=0.0002, weight_decay=0.01)
scheduler = dict(type="OneCycleLR", max_lr=optimizer["lr"], pct_start=0.4, anneal_strategy="cos",
                 div_factor=10.0, final_div_factor=100.0)

data = dict(num_classes=16, ignore_index=-1, names=CLASSES,
            train=dict(type="ConcatDataset",
                       datasets=[dict(type="SyntheticLidarDataset", length=16, point_nsample=512,
                                      loop=1)]))
