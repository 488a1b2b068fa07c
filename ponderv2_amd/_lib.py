"""ctypes binding of libponderv2_hip.so (the C ABI declared in include/ponderv2_hip.h).

There is no CPU fallback: if the shared library is missing the import of any op fails loudly
with instructions to build it (``python -c "import __graft_entry__ as g; g.build()"``).
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t,
                    c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libponderv2_hip.so")
if os.environ.get("PV2_PROBE_LIB"):   # timing probes only (tools/r06_fake_split.sh): another build of the same sources
    LIB_PATH = os.path.join(_HERE, "lib", os.environ["PV2_PROBE_LIB"])

PAIR_TILE = 32      # PV2_PAIR_TILE
WGRAD_TILE = 512    # PV2_WGRAD_TILE
SCAN_CHUNK = 2048   # PV2_SCAN_CHUNK


class VolumeDesc(Structure):
    _fields_ = [(k, c_int64) for k in ("n", "c", "d", "h", "w", "sn", "sc", "sd", "sh", "sw")]


class OsmPlan(Structure):
    """pv2_osm_plan_t: the mask-grouped row order of one gather table (csrc/sparse_conv_osm.hip)."""
    _fields_ = [("tblp", c_void_p), ("perm", c_void_p), ("tmask", c_void_p), ("n_pad", c_int64),
                ("kflip", c_int32), ("reserved", c_int32)]


class ConvGeom(Structure):
    """pv2_conv_geom: one rulebook as the fused conv + BatchNorm entry points take it."""
    _fields_ = ([("K", c_int32), ("tile_pairs_w", c_int32)]
                + [(k, c_int64) for k in ("n_in", "n_out", "n_tiles", "n_tiles_w", "pos_out_stride",
                                          "pos_in_stride")]
                + [(k, c_void_p) for k in ("pair_in", "pair_out", "kstart", "tile_start",
                                           "tile_start_w", "pos_out", "pos_in")]
                + [("osm_fwd", OsmPlan), ("osm_bwd", OsmPlan), ("zero_row", c_void_p)])


class UnetOp(Structure):
    """pv2_unet_op: one record of the natively executed sparse U-Net (csrc/spunet_exec.hip)."""
    _fields_ = ([(k, c_int32) for k in ("kind", "c_in", "c_out", "relu", "K", "kflip",
                                       "dx_accumulate", "dx_producer")]
                + [(k, c_int64) for k in ("n_in", "n_out", "nbr_stride")]
                + [("geom", POINTER(ConvGeom))]
                + [(k, c_void_p) for k in ("nbr", "x", "residual", "weight", "bn_weight", "bn_bias",
                                           "running_mean", "running_var", "y_conv", "mean_invstd",
                                           "out", "grad_out", "dy", "gsum", "dres", "dx", "dweight",
                                           "weight_t")]
                + [("eps", c_float), ("momentum", c_float), ("dtype", c_int32), ("kflip_t", c_int32),
                   ("nbr_t_stride", c_int64), ("n_tiles16", c_int64)]
                + [(k, c_void_p) for k in ("packed_fwd", "packed_bwd", "nbr_t", "perm", "perm_t",
                                           "tile_start16", "dx_tmp")])


UNET_CONV_BN, UNET_STEM, UNET_CONCAT, UNET_CONV_BN16 = 0, 1, 2, 3


class PointsDesc(Structure):
    _fields_ = [(k, c_int64) for k in ("n_points", "points_per_n", "o_sn", "o_sc", "o_sp")]


_P = c_void_p  # every device pointer travels as void*

# name -> (restype, argtypes); mirrors include/ponderv2_hip.h one to one.
ABI_VERSION = 16

SIGNATURES = {
    "pv2_abi_version": (c_int, []),
    "pv2_last_error": (c_char_p, []),
    "pv2_debug_set_os16_variant": (c_int, [c_int]),
    "pv2_zero_fill": (c_int, [_P, c_int64, _P]),
    "pv2_hash_build": (c_int, [_P, c_int64, _P, _P, c_int64, _P]),
    "pv2_subm_neighbor_table": (c_int, [_P, c_int64, c_int, _P, _P, c_int64, _P, _P]),
    "pv2_downsample_workspace_bytes": (c_size_t, [c_int64]),
    "pv2_downsample_unique": (
        c_int, [_P, c_int64, c_int, POINTER(c_int32), _P, _P, _P, _P, _P, c_size_t, _P]),
    "pv2_downsample_table": (
        c_int, [_P, c_int64, c_int, POINTER(c_int32), _P, _P, _P, c_int64, _P]),
    "pv2_table_count": (c_int, [_P, c_int, c_int64, _P, _P, _P, _P]),
    "pv2_table_compact": (c_int, [_P, c_int, c_int64, _P, _P, _P, _P, _P]),
    "pv2_spconv_forward_tile": (c_int, [c_int, c_int]),
    "pv2_spconv_forward": (
        c_int, [_P, c_int64, c_int, _P, c_int, c_int, _P, _P, _P, _P, c_int, c_int64, c_int64,
                c_int64, _P, c_int64, _P]),
    "pv2_tile_prefix": (c_int, [_P, c_int, POINTER(c_int32), c_int, _P, _P]),
    "pv2_table_invert": (c_int, [_P, c_int, c_int64, c_int64, _P, _P, c_int64, _P]),
    "pv2_table_masks": (c_int, [_P, c_int, c_int64, c_int64, _P, _P, _P]),
    "pv2_spconv_os_forward": (
        c_int, [_P, c_int64, c_int, _P, c_int, c_int, _P, c_int64, _P, c_int, _P, _P, c_int64, _P]),
    "pv2_spconv_forward_wt": (
        c_int, [_P, c_int64, c_int, _P, c_int, c_int, _P, _P, _P, _P, c_int, c_int64, _P, c_int64, _P]),
    "pv2_spconv16_packed_elems": (c_int64, [c_int, c_int, c_int]),
    "pv2_spconv16_pack_weights": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "pv2_spconv16_os_forward": (
        c_int, [_P, c_int64, c_int, _P, c_int, c_int, c_int, _P, c_int64, _P, c_int, _P, _P, c_int64,
                _P]),
    "pv2_spconv16_backward_weight": (
        c_int, [_P, c_int64, c_int, _P, c_int64, c_int, c_int, c_int, _P, _P, _P, _P, c_int, c_int64,
                _P, _P]),
    "pv2_spconv_wgrad_tile": (c_int, [c_int, c_int, c_int64, c_int]),
    "pv2_spconv_backward_weight": (
        c_int, [_P, c_int64, c_int, _P, c_int64, c_int, c_int, _P, _P, _P, _P, c_int, c_int64, _P,
                _P]),
    "pv2_spconv_wgrad_partial_floats": (c_int64, [c_int, c_int, c_int64]),
    "pv2_spconv_backward_weight_det": (
        c_int, [_P, c_int64, c_int, _P, c_int64, c_int, c_int, _P, _P, _P, _P, c_int, c_int64, _P, _P,
                _P]),
    "pv2_pair_positions": (c_int, [_P, _P, _P, c_int, c_int64, c_int64, c_int64, _P, _P, _P]),
    "pv2_spconv_products": (c_int, [_P, c_int, _P, c_int, c_int, c_int, _P, _P, _P, c_int64, _P, _P]),
    "pv2_spconv_reduce_rows": (
        c_int, [_P, _P, c_int64, c_int, c_int, c_int64, _P, _P, _P, _P, POINTER(c_int), _P]),
    "pv2_debug_set_osm": (c_int, [c_int, c_int, c_int, c_int]),
    "pv2_osm_plan_workspace_bytes": (c_size_t, [c_int64]),
    "pv2_osm_plan": (c_int, [_P, c_int, c_int64, c_int64, _P, _P, _P, _P, c_int64, _P, c_size_t, _P]),
    "pv2_spconv_osm": (
        c_int, [_P, c_int, _P, c_int, c_int, c_int, POINTER(OsmPlan), c_int64, _P, _P, _P, _P,
                POINTER(c_int), POINTER(c_int), _P]),
    "pv2_convbn_forward": (
        c_int, [POINTER(ConvGeom), _P, c_int, _P, c_int, _P, _P, _P, c_int, c_float, c_float, _P, _P,
                _P, _P, _P, _P, _P, _P]),
    "pv2_convbn_backward": (
        c_int, [POINTER(ConvGeom), _P, _P, c_int, _P, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                _P, _P, _P, _P]),
    "pv2_unet_forward": (c_int, [POINTER(UnetOp), c_int, _P, _P, _P]),
    "pv2_unet_backward": (c_int, [POINTER(UnetOp), c_int, _P, _P, _P, _P, _P]),
    "pv2_unet_backward_ev": (c_int, [POINTER(UnetOp), c_int, _P, _P, _P, _P, _P, c_int, POINTER(c_int32),
                                     POINTER(c_void_p), POINTER(c_void_p)]),
    "pv2_gemm_nt": (c_int, [_P, c_int64, c_int, _P, c_int, _P, _P, _P]),
    "pv2_gemm_tn": (c_int, [_P, _P, c_int64, c_int, c_int, _P, _P]),
    "pv2_bn_workspace_floats": (c_int64, [c_int]),
    "pv2_bn_forward_mixed": (c_int, [_P, c_int, c_int64, c_int, _P, _P, _P, c_int, c_float, c_float, _P,
                                     _P, _P, _P, _P, c_int, _P]),
    "pv2_bn_backward_mixed": (c_int, [_P, _P, c_int, _P, c_int, _P, _P, c_int64, c_int, _P, _P, _P, _P,
                                      _P]),
    "pv2_bn_forward": (c_int, [_P, c_int64, c_int, _P, _P, _P, c_int, c_float, c_float, _P, _P, _P,
                               _P, _P, _P]),
    "pv2_bn_backward": (c_int, [_P, _P, _P, _P, _P, c_int64, c_int, _P, _P, _P, _P, _P]),
    "pv2_col_sum": (c_int, [_P, c_int64, c_int, _P, _P]),
    "pv2_raymarch_weights_forward": (c_int, [_P, c_int64, c_int, _P, _P, _P]),
    "pv2_raymarch_weights_backward": (c_int, [_P, _P, c_int64, c_int, _P, _P]),
    "pv2_raymarch_accumulate_forward": (c_int, [_P, _P, c_int64, c_int, c_int, _P, _P]),
    "pv2_raymarch_accumulate_backward": (c_int, [_P, _P, _P, c_int64, c_int, c_int, _P, _P, _P]),
    "pv2_neus_head_dims": (c_int, [POINTER(c_int)] * 6),
    "pv2_neus_coarse_sample": (
        c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int64, c_int, c_int, _P, _P,
                c_int, _P, _P, c_int, _P, _P, _P, _P, _P, c_float, _P, _P, _P, _P, _P, _P, _P]),
    "pv2_neus_field_forward": (
        c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int64, c_int] + [_P] * 10
        + [c_int, c_float] + [_P] * 7 + [_P]),
    "pv2_neus_field_backward": (
        c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int64, c_int] + [_P] * 7
        + [c_int, c_float] + [_P] * 9 + [_P] * 9 + [_P]),
    "pv2_ray_setup_record_sizes": (c_int, [POINTER(c_int)] * 2),
    "pv2_unit_cube": (c_int, [_P, _P, c_int, c_int64, c_int, c_float] + [_P] * 9),
    "pv2_ray_gen": (c_int, [_P, c_int, c_int, c_int, c_int, c_int] + [_P] * 13),
    "pv2_surface_loss_workspace_floats": (c_int, []),
    "pv2_surface_loss_forward": (c_int, [_P] * 7 + [c_int64, c_int, c_float, _P, _P, _P, _P, _P]),
    "pv2_surface_loss_backward": (c_int, [_P] * 7 + [c_int64, c_int, c_float, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pv2_ray_loss_info_floats": (c_int, []),
    "pv2_ray_rows_forward": (c_int, [_P, c_int, c_int, c_int, _P, c_int64, c_int, c_int, c_float, c_float,
                                     c_float, _P, _P, _P, _P, _P]),
    "pv2_ray_rows_backward": (c_int, [_P, c_int, c_int, c_int, c_int64, c_int, c_float, c_float, c_float,
                                      _P, _P, _P, _P, _P, _P]),
    "pv2_semantic_ce_forward": (c_int, [_P, _P, _P, _P, c_int64, c_int, c_float, _P, _P]),
    "pv2_ray_loss_finalize": (c_int, [_P, c_int64, c_float, _P, _P, _P]),
    "pv2_semantic_ce_backward": (c_int, [_P, _P, _P, _P, _P, c_int64, c_int, c_float, _P, _P, _P]),
    "pv2_narrow_head_dims": (c_int, [POINTER(c_int)] * 4),
    "pv2_narrow_coarse_sample": (
        c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int64, c_int, c_int, _P, _P,
                c_int, _P, _P, c_int, _P, c_float, c_float, _P, _P, _P, _P, _P, _P, _P]),
    "pv2_narrow_field_forward": (
        c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int64, c_int, _P, c_float, _P,
                _P, _P, _P, _P, _P, _P]),
    "pv2_narrow_backward_slabs": (c_int64, [c_int64, c_int]),
    "pv2_narrow_field_backward": (
        c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int64, c_int, _P, c_float, _P,
                _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pv2_neus_fold_dims": (c_int, [POINTER(c_int)] * 2),
    "pv2_neus_fold_gather": (
        c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, c_int64, c_int, c_int, c_float, _P,
                _P, _P]),
    "pv2_neus_fold_scatter": (
        c_int, [c_int, c_int, c_int, c_int, c_int, _P, _P, _P, c_int64, c_int, c_int, c_float, _P, _P,
                _P, _P, _P]),
    "pv2_neus_coarse_sample_folded": (
        c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, c_int64, c_int, c_int, _P,
                _P, c_int, _P, _P, c_int, _P, _P, _P, _P, _P, c_float, _P, _P, _P, _P, _P, _P, _P]),
    "pv2_neus_field_forward_rows": (
        c_int, [_P, _P, _P, _P, _P, _P, c_int64, c_int] + [_P] * 10 + [c_int, c_float] + [_P] * 7
        + [_P]),
    "pv2_neus_field_backward_rows": (
        c_int, [_P, _P, _P, _P, _P, c_int64, c_int] + [_P] * 7 + [c_int, c_float] + [_P] * 8
        + [_P] * 8 + [_P]),
    "pv2_maxpool3d_cl_forward": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "pv2_maxpool3d_cl_backward": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "pv2_maxpool3d_cl_backward_add": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "pv2_small_inverse": (c_int, [_P, c_int64, c_int, _P, _P]),
    "pv2_gather_rows": (c_int, [_P, _P, c_int64, c_int, _P, _P]),
    "pv2_scatter_add": (c_int, [_P, _P, c_int64, c_int, _P, _P, c_int64, _P]),
    "pv2_scatter_mean_finish": (c_int, [_P, _P, c_int64, c_int, _P]),
    "pv2_scatter_backward": (c_int, [_P, _P, _P, c_int64, c_int, _P, c_int64, _P]),
}
for _sfx in ("f32", "f64"):
    SIGNATURES["pv2_trilinear_forward_" + _sfx] = (
        c_int, [_P, POINTER(VolumeDesc), _P, POINTER(PointsDesc), _P, c_int, c_int, c_int, _P])
    SIGNATURES["pv2_trilinear_backward_" + _sfx] = (
        c_int, [_P, _P, POINTER(VolumeDesc), _P, POINTER(PointsDesc), _P, _P,
                c_int, c_int, c_int, _P])
    SIGNATURES["pv2_trilinear_backward_backward_" + _sfx] = (
        c_int, [_P, _P, _P, POINTER(VolumeDesc), _P, _P, POINTER(PointsDesc), _P, _P, _P,
                c_int, c_int, c_int, _P])
SIGNATURES["pv2_trilinear_forward_16"] = (
    c_int, [_P, c_int, POINTER(VolumeDesc), _P, POINTER(PointsDesc), _P, c_int, c_int, c_int, _P])
SIGNATURES["pv2_trilinear_backward_16"] = (
    c_int, [_P, _P, c_int, POINTER(VolumeDesc), _P, POINTER(PointsDesc), _P, _P,
            c_int, c_int, c_int, _P])
SIGNATURES["pv2_trilinear_backward_backward_16"] = (
    c_int, [_P, _P, _P, c_int, POINTER(VolumeDesc), _P, _P, POINTER(PointsDesc), _P, _P, _P,
            c_int, c_int, c_int, _P])
SIGNATURES["pv2_dconv3_set_one_term"] = (c_int, [c_int])
SIGNATURES["pv2_dconv3_packed_floats"] = (c_int64, [c_int, c_int, c_int])
SIGNATURES["pv2_dconv3_pack_weights"] = (c_int, [_P, c_int, c_int] + [c_int64] * 5 + [c_int, c_int, _P, _P])
SIGNATURES["pv2_dconv3_forward"] = (
    c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, c_int, c_int, _P, _P, _P, _P, _P, c_int, _P, _P, _P])
SIGNATURES["pv2_dconv3_wgrad_partial_floats"] = (c_int64, [c_int] * 7)
SIGNATURES["pv2_dconv3_backward_weight"] = (
    c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, c_int, _P, c_int, _P, _P]
    + [c_int64] * 5 + [_P])
SIGNATURES["pv2_bn_statistics"] = (
    c_int, [_P, c_int64, c_int, _P, _P, c_float, c_float, _P, _P, _P, _P, _P, _P])
SIGNATURES["pv2_bn_statistics_padded"] = (
    c_int, [_P, c_int64, c_int64, c_int, _P, _P, c_float, c_float, _P, _P, _P, _P, _P, _P])
SIGNATURES["pv2_bn_backward_padded"] = (
    c_int, [_P, _P, c_int64, c_int64, c_int, _P, _P, _P, c_int, _P, _P, _P, _P])
SIGNATURES["pv2_cells_tap_table"] = (c_int, [_P, c_int64, c_int, c_int, c_int, _P, _P])
SIGNATURES["pv2_cells_fold_weights"] = (
    c_int, [_P, c_int64, c_int64, c_int64, c_int64, c_int64, c_int, c_int, _P, _P, _P, _P, _P])
SIGNATURES["pv2_cells_expand"] = (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P])
SIGNATURES["pv2_cells_backward_workspace_floats"] = (c_int64, [c_int, c_int, c_int, c_int])
SIGNATURES["pv2_cells_backward_table"] = (
    c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, c_int64, c_int64, c_int64, c_int64, c_int64,
            c_int, _P, _P, _P, _P])
SIGNATURES["pv2_cells_dw_finish"] = (
    c_int, [_P, _P, _P, c_int, c_int, _P, c_int64, c_int64, c_int64, c_int64, c_int64, _P])
SIGNATURES["pv2_voxelize_stage1"] = (
    c_int, [_P, c_int, c_int64, ctypes.c_double, c_int, _P, _P, _P, c_int64, _P, _P, _P, _P, _P, _P])
SIGNATURES["pv2_voxelize_workspace_bytes"] = (c_size_t, [c_int64])
SIGNATURES["pv2_voxelize_stage2"] = (
    c_int, [c_int64, c_int64] + [_P] * 14 + [_P, c_size_t, _P, _P, _P])

_lib = None


def lib():
    """Load (once) and return the ctypes handle; raises if the library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built. Run "
                "`make -C ponderv2_amd/csrc` (or __graft_entry__.build()). There is no CPU "
                "fallback for the ponderv2_amd ops.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if handle.pv2_abi_version() != ABI_VERSION:  # a stale .so with other signatures
            raise RuntimeError(f"{LIB_PATH} has ABI version {handle.pv2_abi_version()}, this package "
                               f"binds version {ABI_VERSION}: rebuild with `make -C ponderv2_amd/csrc`")
        _lib = handle
    return _lib


def check(status, what=""):
    if status != 0:
        msg = lib().pv2_last_error()
        raise RuntimeError(f"libponderv2_hip: {what} failed with status {status}: "
                           f"{msg.decode() if msg else ''}")
