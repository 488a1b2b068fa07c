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



def limit_hardware_queues_for_process_group():
    """With an RCCL process group in the process the step runs on six HIP streams (training, input,
    geometry, weight-gradient side stream, communication stream, RCCL's own) over ROCclr's default of FOUR
    hardware queues - and the streams that share a queue serialise: measured with one rank on MI355X
    (profiles/r05_hw_queues.txt) 22.5 - 23.2 ms per step at the default, 44 ms at 6 or 8 queues, 20.1 at 3,
    **19.9 at 2** (without a process group: 19.3 at the default, 19.4 at 2).  ``GPU_MAX_HW_QUEUES`` is read
    when the HIP runtime initialises, so this must run before the first device call; an explicit setting
    in the environment wins."""
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")


# ranks started by a launcher (torch.distributed.run, the driver's multi-GPU bench line) carry WORLD_SIZE
if int(_os.environ.get("WORLD_SIZE", "1") or "1") > 1 or _os.environ.get("PV2_BENCH_FORCE_DIST") == "1":
    limit_hardware_queues_for_process_group()

__version__ = "0.1.0"
