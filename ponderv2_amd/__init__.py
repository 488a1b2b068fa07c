"""ponderv2_amd - MI355X-native PonderV2 pre-training hot path.

Sub-modules mirror the import boundaries of the reference:
  ponderv2_amd.spconv.pytorch   <->  spconv.pytorch      (sparse conv runtime)
  ponderv2_amd.smooth_sampler   <->  smooth_sampler      (twice-differentiable trilinear sampler)
  ponderv2_amd.torch_scatter    <->  torch_scatter       (scatter mean/sum)
  ponderv2_amd.ponder           <->  ponder              (registry, config, models, engine)
All device arithmetic goes through libponderv2_hip.so (include/ponderv2_hip.h).
"""
import os as _os

# The dense UNet3D projection runs on MIOpen.  On gfx950 MIOpen's immediate-mode heuristics pick a
# pathological fp32 3-D weight-gradient solver (553 ms per step); the solver search fixes that
# (9 ms) but costs ~100 s on every fresh process.  The search results for the shapes of the
# ScanNet configuration ship in miopen_cache/ (MIOpen user find-db + kernel cache for this
# GPU / MIOpen build) and are picked up in immediate mode.
_ROOT = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
_os.environ.setdefault("MIOPEN_USER_DB_PATH", _os.path.join(_ROOT, "miopen_cache", "db"))
_os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", _os.path.join(_ROOT, "miopen_cache", "cache"))

__version__ = "0.1.0"
