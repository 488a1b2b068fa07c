from .voxelize import GridSample, fnv_hash_vec, ravel_hash_vec  # noqa: F401
from .synthetic import SyntheticRGBDDataset, collate_fn, make_scene  # noqa: F401
