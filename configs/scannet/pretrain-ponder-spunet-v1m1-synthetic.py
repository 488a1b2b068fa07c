# PonderV2 indoor pre-training on SYNTHETIC ScanNet-shaped scenes (no dataset in this environment).
# The model section carries the hyper-parameters of the reference's
# configs/scannet/pretrain-ponder-spunet-v1m1-0-base.py; only the data section differs.
_base_ = ["../_base_/default_runtime.py"]

batch_size = 2          # total over all GPUs
num_worker = 2
enable_amp = False      # fp32: the parity configuration
find_unused_parameters = True   # embedding_table / laplace beta / proj_head get no gradient
epoch = 4
eval_epoch = 4

CLASSES = ("wall", "floor", "cabinet", "bed", "chair", "sofa", "table", "door", "window",
           "bookshelf", "picture", "counter", "desk", "curtain", "refridgerator",
           "shower curtain", "toilet", "sink", "bathtub", "otherfurniture")
_mlp = dict(hidden_size=128, points_factor=0.0)

model = dict(
    type="PonderIndoor-v2",
    backbone=dict(type="SpUNet-v1m1", in_channels=6, num_classes=0,
                  channels=(32, 64, 128, 256, 256, 128, 96, 96), layers=(2, 3, 4, 6, 2, 2, 2, 2)),
    projection=dict(type="UNet3D-v1m2", in_channels=96, out_channels=128),
    renderer=dict(
        type="NeuSModel",
        field=dict(type="SDFField",
                   sdf_decoder=dict(in_dim=64, out_dim=65, n_blocks=1, pos_enc=False, **_mlp),
                   rgb_decoder=dict(in_dim=134, out_dim=3, n_blocks=0, pos_enc=False, **_mlp),
                   semantic_decoder=dict(in_dim=131, out_dim=512, n_blocks=0, **_mlp),
                   beta_init=0.3, use_gradient=True, volume_type="default", padding_mode="zeros",
                   share_volume=False, norm_pts=True, norm_padding=0.1),
        collider=dict(type="AABBBoxCollider", near_plane=0.01, bbox=[-0.55] * 3 + [0.55] * 3),
        sampler=dict(type="NeuSSampler", initial_sampler="UniformSampler", num_samples=96,
                     num_samples_importance=36, num_upsample_steps=1, train_stratified=True,
                     single_jitter=False),
        loss=dict(sensor_depth_truncation=0.05, temperature=0.01,
                  weights=dict(eikonal_loss=0.01, free_space_loss=1.0, sdf_loss=10.0,
                               depth_loss=1.0, rgb_loss=10.0, semantic_loss=0.1))),
    mask=None, grid_shape=(128, 128, 32), grid_size=0.02, val_ray_split=10240, ray_nsample=256,
    padding=0.1, pool_type="mean", render_semantic=True, conditions=("ScanNet",),
    template="a photo of a [x]", clip_model="ViT-B/16", class_name=CLASSES,
    valid_index=(tuple(range(20)),), ppt_loss_weight=1.0,
    ppt_criteria=[dict(type="CrossEntropyLoss", loss_weight=1.0, ignore_index=-1)])

optimizer = dict(type="SGD", lr=0.0005 * batch_size / 8, momentum=0.9, weight_decay=0.0001,
                 nesterov=True)
scheduler = dict(type="OneCycleLR", max_lr=optimizer["lr"], pct_start=0.05, anneal_strategy="cos",
                 div_factor=10.0, final_div_factor=10000.0)

data = dict(num_classes=20, ignore_index=-1, names=CLASSES,
            train=dict(type="SyntheticRGBDDataset", length=8, num_views=2, image_hw=(480, 640)))
