"""ponderv2_amd - MI355X-native PonderV2 pre-training hot path.

Sub-modules mirror the import boundaries of the reference:
  ponderv2_amd.spconv.pytorch   <->  spconv.pytorch      (sparse conv runtime)
  ponderv2_amd.smooth_sampler   <->  smooth_sampler      (twice-differentiable trilinear sampler)
  ponderv2_amd.torch_scatter    <->  torch_scatter       (scatter mean/sum)
  ponderv2_amd.ponder           <->  ponder              (registry, config, models, engine)
All device arithmetic goes through libponderv2_hip.so (include/ponderv2_hip.h).
"""
__version__ = "0.1.0"
