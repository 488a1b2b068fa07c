# PonderV2 multi-dataset (PPT) indoor pre-training on SYNTHETIC RGB-D scenes: three "datasets"
# (Structured3D / ScanNet / S3DIS conditions, sampling ratio 4:2:1) through MultiDatasetTrainer and
# the prompt-driven-normalisation backbone SpUNet-v1m3.  Model / optimiser sections carry the
# hyper-parameters of the reference's configs/scannet/pretrain-ponder-ppt-v1m1-0-sc-s3-st-spunet.py;
# only the data section differs (seeded scene generators replace the on-disk datasets).
_base_ = ["../_base_/default_runtime.py"]

batch_size = 2          # total over all GPUs (the reference: 8 per GPU)
num_worker = 2
mix_prob = 0.0
enable_amp = False
find_unused_parameters = True   # the other datasets' BatchNorms get no gradient in a given step
epoch = 4
eval_epoch = 1

train = dict(type="MultiDatasetTrainer")

CLASSES = ("wall", "floor", "cabinet", "bed", "chair", "sofa", "table", "door", "window",
           "bookshelf", "bookcase", "picture", "counter", "desk", "shelves", "curtain", "dresser",
           "pillow", "mirror", "ceiling", "refrigerator", "television", "shower curtain",
           "nightstand", "toilet", "sink", "lamp", "bathtub", "garbagebin", "board", "beam",
           "column", "clutter", "other structure", "other furniture", "other property")
VALID = ((0, 1, 2, 3, 4, 5, 6, 7, 8, 11, 13, 14, 15, 16, 17, 18, 19, 20, 21, 23, 25, 26, 33, 34, 35),
         (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 12, 13, 15, 20, 22, 24, 25, 27, 34),
         (0, 1, 4, 5, 6, 7, 8, 10, 19, 29, 30, 31, 32))
_mlp = dict(hidden_size=128, points_factor=0.0)

model = dict(
    type="PonderIndoor-v2",
    backbone=dict(type="SpUNet-v1m3", in_channels=6, num_classes=0, base_channels=32,
                  context_channels=256, channels=(32, 64, 128, 256, 256, 128, 96, 96),
                  layers=(2, 3, 4, 6, 2, 2, 2, 2), cls_mode=False,
                  conditions=("ScanNet", "S3DIS", "Structured3D"), zero_init=False,
                  norm_decouple=True, norm_adaptive=True, norm_affine=True),
    projection=dict(type="UNet3D-v1m2", in_channels=96, out_channels=128),
    renderer=dict(
        type="NeuSModel",
        field=dict(type="SDFField",
                   sdf_decoder=dict(in_dim=64, out_dim=65, n_blocks=1, **_mlp),
                   rgb_decoder=dict(in_dim=134, out_dim=3, n_blocks=0, **_mlp),
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
    padding=0.1, backbone_out_channels=96, context_channels=256, pool_type="mean",
    render_semantic=True, conditions=("Structured3D", "ScanNet", "S3DIS"),
    template="a photo of a [x]", clip_model="ViT-B/16", class_name=CLASSES, valid_index=VALID,
    ppt_loss_weight=1.0,
    ppt_criteria=[dict(type="CrossEntropyLoss", loss_weight=1.0, ignore_index=-1)])

optimizer = dict(type="SGD", lr=0.0001 * batch_size / 8, momentum=0.9, weight_decay=0.0001,
                 nesterov=True)
scheduler = dict(type="OneCycleLR", max_lr=optimizer["lr"], pct_start=0.05, anneal_strategy="cos",
                 div_factor=10.0, final_div_factor=10000.0)

_scene = dict(type="SyntheticRGBDDataset", num_views=2, image_hw=(480, 640))
data = dict(
    num_classes=20, ignore_index=-1,
    train=dict(type="ConcatDataset", loop=1, datasets=[
        dict(_scene, length=8, base_seed=0, condition="Structured3D", num_classes=25, loop=4),
        dict(_scene, length=8, base_seed=1000, condition="ScanNet", num_classes=20, loop=2),
        dict(_scene, length=8, base_seed=2000, condition="S3DIS", num_classes=13, loop=1)]))
