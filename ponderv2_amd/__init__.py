"""ponderv2_amd - MI355X-native PonderV2 pre-training hot path.

Sub-modules mirror the import boundaries of the reference:
  ponderv2_amd.spconv.pytorch   <->  spconv.pytorch      (sparse conv runtime)
  ponderv2_amd.smooth_sampler   <->  smooth_sampler      (twice-differentiable trilinear sampler)
  ponderv2_amd.torch_scatter    <->  torch_scatter       (scatter mean/sum)
  ponderv2_amd.ponder           <->  ponder              (registry, config, models, engine)
All device arithmetic goes through libponderv2_hip.so (include/ponderv2_hip.h).
"""
import os as _os

# The dense UNet3D projection runs on the hand-written kernels of csrc/dense_conv.hip (round 4); the library
# convolutions are only its FALLBACK (evaluation mode, other layer orders, host tensors).  For that
# fallback: on gfx950 MIOpen's immediate-mode heuristics pick a pathological fp32 3-D weight-gradient
# solver (553 ms per step); the solver search fixes that (9 ms) but costs ~100 s on every fresh process,
# so the search results for the shapes of the ScanNet configuration ship in miopen_cache/ (MIOpen user
# find-db + kernel cache for this GPU / MIOpen build) and are picked up in immediate mode.
_ROOT = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
_os.environ.setdefault("MIOPEN_USER_DB_PATH", _os.path.join(_ROOT, "miopen_cache", "db"))
_os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", _os.path.join(_ROOT, "miopen_cache", "cache"))



def limit_hardware_queues_for_process_group():
    """``GPU_MAX_HW_QUEUES=2`` - for the LAUNCHER of a multi-rank job to call (or set) before the ranks'
    HIP runtimes initialise; ``bench.py`` and ``engines/launch.py`` do.  The library itself no longer
    touches the process environment on import; ``PV2_LIMIT_HW_QUEUES=1`` asks for it.
    Why two: with an RCCL communicator in the process the default of four hardware queues costs the step
    +1.6 ms with no reduction issued at all and +2.9 ms with one; two or three queues +0.4 ms; without a
    communicator the count does not matter (profiles/r06_one_rank_pg.txt, profiles/r05_hw_queues.txt).
    Round 6 also brought this program from six HIP streams per rank to four (geometry built on the input
    stream, slab reductions stream-ordered on one communication stream): no change at four queues - the
    interaction is between RCCL and the queues, not the stream count."""
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")


if _os.environ.get("PV2_LIMIT_HW_QUEUES") == "1":
    limit_hardware_queues_for_process_group()

__version__ = "0.1.0"
