from .voxelize import GridSample, fnv_hash_vec, ravel_hash_vec  # noqa: F401
from .synthetic import SyntheticRGBDDataset, collate_fn, make_scene  # noqa: F401
from .lidar import (PointRangeFilter, ProjectOnImage, RaySample, SyntheticLidarDataset,  # noqa: F401
                    lidar_collate_fn, make_lidar_scene, make_sweep)
from .dataloader import ConcatDataset, MultiDatasetDataloader  # noqa: F401
