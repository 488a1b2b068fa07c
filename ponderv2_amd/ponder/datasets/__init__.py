from .voxelize import GridSample, fnv_hash_vec, ravel_hash_vec  # noqa: F401
from .synthetic import SyntheticRGBDDataset, collate_fn, make_scene  # noqa: F401
from .lidar import (PointRangeFilter, ProjectOnImage, RaySample, SyntheticLidarDataset,  # noqa: F401
                    lidar_collate_fn, make_lidar_scene, make_sweep)
from .dataloader import ConcatDataset, MultiDatasetDataloader  # noqa: F401
from .transform import TRANSFORMS, Compose  # noqa: F401
from .collate import point_collate_fn  # noqa: F401
from .readers import (NuScenesDataset, S3DISRGBDDataset, ScanNetRGBDDataset,  # noqa: F401
                      Structured3DRGBDDataset)
