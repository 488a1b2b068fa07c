"""Tensor-level wrappers over the C ABI: rulebook construction, sparse-conv arithmetic, dense
scatter and the trilinear sampler.  Everything here runs on the current HIP stream of a CUDA
(ROCm) device; CPU tensors are rejected - there is deliberately no fallback path.
"""
import ctypes
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch

from . import _lib
from ._lib import PAIR_TILE, SCAN_CHUNK, WGRAD_TILE, PointsDesc, VolumeDesc

import os

FWD_LDS_TILE = 128  # tile of the LDS-staged forward kernel (pv2_spconv_forward_tile)
FWD_LDS_TILE_K = 32  # ... which needs the reduction width to be a multiple of this
# Which forward / grad-input kernels an fp32 conv runs on.
#
# PRODUCT-ROW path (USE_PR, csrc/sparse_conv_pr.hip; the default): stage 1 is the pair-major MFMA
# kernel writing one product row per PAIR with plain stores, stage 2 sums the rows of every output
# voxel in ascending offset order.  No atomics, no zero-fill, bitwise reproducible; needs
# c_in % 32 == 0 and c_out % 4 == 0 and a rulebook of at most 32 offsets.
#   PV2_CONV_PR = "all" (default)  every eligible conv, strided / inverse convs included;
#                 "subm"           the 27-offset submanifold convs only;
#                 "0"              off: the kernels below.
# Where it does not apply (channel counts that are no multiples of 32 / 4, the 125-offset stem) and
# when it is off, USE_OS decides:
#   "auto" (default) / True / PV2_SPCONV_OS=1   the output-stationary kernel over the rulebook's
#                     gather table (no atomics, bitwise reproducible, any channel count);
#   False / PV2_SPCONV_OS=0   the pair-major scatter-add kernels (device-scope fp32 atomics on a
#                     zero-filled output) - the round-1/2 default for submanifold convs, kept as the
#                     A/B baseline.  Rulebooks without a gather table (the dense grid's first layer,
#                     models/ponder/sparse_input.py) always take them.
# So the default selection contains no atomics anywhere in the sparse backbone.
_OS_ENV = os.environ.get("PV2_SPCONV_OS", "auto")
USE_OS = True if _OS_ENV == "1" else False if _OS_ENV == "0" else "auto"
_PR_ENV = os.environ.get("PV2_CONV_PR", "all")
USE_PR = False if _PR_ENV in ("0", "off", "") else ("subm" if _PR_ENV == "subm" else "all")
# Weight gradient in the deterministic two-stage form (partial slabs + ordered reduction) instead of
# fp32 atomics on a zero-filled dW.  PV2_WGRAD_DET=0 restores the atomics.
USE_WGRAD_DET = os.environ.get("PV2_WGRAD_DET", "1") != "0"
# conv -> BatchNorm -> (+shortcut) -> ReLU as one C call per direction (pv2_convbn_*; needs USE_PR).
USE_CONVBN = os.environ.get("PV2_CONVBN", "1") != "0"
PR_MAX_K = 32


def _use_pr(rb, c_in, c_out) -> bool:
    if USE_PR is False or rb.K > PR_MAX_K or c_in % FWD_LDS_TILE_K or c_out % 4 or rb.n_pairs <= 0:
        return False
    if rb.bounded:  # pair counts known on the device only: the product buffer would be worst-case
        return False
    if c_in < FWD_LDS_TILE_K or c_out > 512:
        return False
    return USE_PR == "all" or rb.center_k >= 0 and rb.K > 1


def _use_os(rb) -> bool:
    if rb.nbr is None or USE_OS is False:
        return False
    return True
# MASK-GROUPED OUTPUT-STATIONARY route (round 5, csrc/sparse_conv_osm.hip): one launch per conv and
# direction, sums kept in the MFMA accumulators - no product rows, no row reduce.  The plans (row order,
# permuted gather table, tile masks) are built with the rulebooks; the C side decides per shape
# (pv2::use_osm) unless PV2_CONV_OSM forces it:
#   "1"               plans for every rulebook with at most 31 offsets, every planned conv takes the route;
#   "0" / "auto" (default)  no plans: the product-row route everywhere.  Measured on MI355X
#                     (profiles/r05_spconv_ab.txt): the 27-offset submanifold convs are 1.0 - 4x SLOWER on
#                     this route; the 8-offset strided / inverse convs are up to 1.7x faster per launch
#                     (62 us per step in total), but building their eight plans on the geometry stream
#                     costs more than that - 19.44 ms per step with them against 19.30 without (same box,
#                     two alternating runs each).  The route stays available and tested; it is not the default.
OSM_MODE = os.environ.get("PV2_CONV_OSM", "auto")
if OSM_MODE not in ("0", "1"):
    OSM_MODE = "auto"
OSM_MAX_K = 31


def _want_osm(K: int) -> bool:
    if OSM_MODE == "0" or USE_PR is False or K > OSM_MAX_K or K < 2:
        return False
    return OSM_MODE == "1"


_ZERO_ROWS = {}


def zero_row(device) -> torch.Tensor:
    """4096 zero floats per device: what rows without a neighbour read in the output-stationary conv."""
    z = _ZERO_ROWS.get(device.index)
    if z is None:
        z = _ZERO_ROWS[device.index] = torch.zeros(4096, dtype=torch.float32, device=device)
    return z


class OsmPlanData:
    """Device arrays of one pv2_osm_plan (kept alive by the rulebook) + the struct the C side reads."""

    def __init__(self, perm, tblp, tmask, n_pad, kflip):
        self.perm, self.tblp, self.tmask, self.n_pad, self.kflip = perm, tblp, tmask, n_pad, kflip
        self.struct = _lib.OsmPlan(tblp.data_ptr(), perm.data_ptr(), tmask.data_ptr(), n_pad, kflip, 0)

    def flipped(self):
        """The same table read with mirrored weight offsets (grad-input of a submanifold conv)."""
        return OsmPlanData(self.perm, self.tblp, self.tmask, self.n_pad, 1 - self.kflip)


def build_osm_plan(tbl: torch.Tensor, K: int, n_cols: int, stride: int,
                   n_cols_dev: Optional[torch.Tensor] = None, kflip: int = 0) -> OsmPlanData:
    """pv2_osm_plan on the current stream: rows of the gather table ``tbl`` [K, stride] sorted by their
    offset mask, the table in that order and the per-tile masks."""
    L = _lib.lib()
    dev = tbl.device
    n_pad = (max(n_cols, 1) + 255) // 256 * 256
    perm = torch.empty(max(n_cols, 1), dtype=torch.int32, device=dev)
    tblp = torch.empty(K * n_pad, dtype=torch.int32, device=dev)
    tmask = torch.empty(n_pad // 32, dtype=torch.int32, device=dev)
    ws_bytes = int(L.pv2_osm_plan_workspace_bytes(n_cols))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    _lib.check(L.pv2_osm_plan(_ptr(tbl), K, n_cols, stride, _ptr(n_cols_dev), _ptr(perm), _ptr(tblp),
                              _ptr(tmask), n_pad, _ptr(ws), ws_bytes, _stream(tbl)), "pv2_osm_plan")
    return OsmPlanData(perm, tblp, tmask, n_pad, kflip)


# Run the centre offset of submanifold convs as a separate plain-store pass (no zero-fill, fewer
# atomics) on the scatter-add path.  Measured neutral on MI355X at the ScanNet batch (the second
# launch and its smaller grids cost what the saved fill and atomics gain): off.
USE_CENTER_STORE = False


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(t: torch.Tensor):
    """The current stream of ``t``'s device as a hipStream_t (the raw handle: this is called once
    per kernel launch, a few hundred times per training step)."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(t.device.index if t.device.index is not None
                                           else torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


# element type codes of the entry points that take 16-bit feature matrices (include/ponderv2_hip.h)
DTYPE_CODE = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}
HALF_DTYPES = (torch.bfloat16, torch.float16)


def _zero_fill(t):
    if t.numel():
        _lib.check(_lib.lib().pv2_zero_fill(_ptr(t), t.numel() * t.element_size(), _stream(t)),
                   "pv2_zero_fill")
    return t


def zeros_by_kernel(shape, dtype, device):
    """torch.zeros, but cleared by a kernel launch on the current stream (see pv2_zero_fill).

    Each scatter-add target is cleared by its own launch RIGHT BEFORE the kernel that accumulates
    onto it.  Clearing all ~160 targets of a step from one arena at the start of the step was tried
    (one fill per arena instead of 160 launches) and measured on MI355X: neutral in fp32 (32.9 vs
    33.0 ms per step), 6.5 ms SLOWER under --amp bf16 (31.8 vs 25.3 ms) - the atomics then land on
    lines that have long left the L2 / Infinity Cache instead of lines the fill has just written."""
    return _zero_fill(torch.empty(shape, dtype=dtype, device=device))


_WORKSPACES = {}


def workspace(kind: str, device, floats: int, stream=None) -> torch.Tensor:
    """A reusable fp32 scratch buffer per (kind, device, stream): launches on one stream are
    ordered, so one buffer serves every layer.  ``stream``: a torch stream the buffer is used on
    when that is not the current one (it is then allocated from that stream's pool).  Grown with
    slack on demand."""
    handle = (stream.cuda_stream if stream is not None
              else _raw_stream(device.index) if _raw_stream is not None
              else torch.cuda.current_stream(device).cuda_stream)
    key = (kind, device.index, handle)
    ws = _WORKSPACES.get(key)
    if ws is None or ws.numel() < floats:
        n = max(int(floats * 1.25), 1 << 16)
        if stream is not None:
            with torch.cuda.stream(stream):
                ws = torch.empty(n, dtype=torch.float32, device=device)
        else:
            ws = torch.empty(n, dtype=torch.float32, device=device)
        _WORKSPACES[key] = ws
    return ws


def _require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "ponderv2_amd ops run on MI355X only (got a CPU tensor); there is no CPU fallback")


# --------------------------------------------------------------------------------------------
# Rulebooks
# --------------------------------------------------------------------------------------------
TILE_SIZES = (PAIR_TILE, FWD_LDS_TILE, WGRAD_TILE, 2048)


def device_tile_prefixes(kstart: torch.Tensor, K: int) -> torch.Tensor:
    """int32 [4, K+1] on the device: prefix of ceil(count_k / t) for the four tile sizes the conv
    kernels use, by ONE small launch on the current stream from the device copy of ``kstart`` (no
    host->device copy, which would stall the host behind the queue)."""
    dev = torch.empty((len(TILE_SIZES), K + 1), dtype=torch.int32, device=kstart.device)
    arr = (ctypes.c_int32 * len(TILE_SIZES))(*TILE_SIZES)
    _lib.check(_lib.lib().pv2_tile_prefix(_ptr(kstart), K, arr, len(TILE_SIZES), _ptr(dev),
                                          _stream(kstart)), "pv2_tile_prefix")
    return dev


@dataclass
class Rulebook:
    """Indice pairs of one sparse conv, in canonical (offset k, output row) order."""

    K: int
    n_in: int
    n_out: int
    pair_in: torch.Tensor    # int32 [P]
    pair_out: torch.Tensor   # int32 [P]
    kstart: torch.Tensor     # int32 [K+1] device
    kstart_host: np.ndarray  # int64 [K+1]
    # offset whose pairs hit every output row exactly once (centre tap of a submanifold conv, the
    # single offset of a 1x1 conv); -1 when there is none (strided / inverse convs)
    center_k: int = -1
    _tiles: dict = field(default_factory=dict)
    # output-stationary view: gather table [K, nbr_stride] (input row feeding output row o under
    # offset k, or -1), the row order `perm` that groups rows with equal offset masks, and whether
    # the weight offsets are read mirrored (grad-input of a submanifold conv)
    nbr: Optional[torch.Tensor] = None
    nbr_stride: int = 0
    perm: Optional[torch.Tensor] = None
    kflip: int = 0
    _transposed_os: Optional[tuple] = None   # (nbr, stride, perm, kflip) of the transposed rulebook
    # position tables of the product-row path (pv2_pair_positions): pos_out[k, o] / pos_in[k, i] =
    # index of the pair of offset k with output row o / input row i, or -1; (tensor, row stride)
    _pos: Optional[tuple] = None             # (pos_out, out_stride, pos_in, in_stride)
    # device tile prefixes [4, K+1] launched where the rulebook was BUILT (see device_tile_prefixes)
    _tiles_dev: Optional[torch.Tensor] = None
    bounded: bool = False                    # host-side pair counts are upper bounds (no read-back)
    _geoms: dict = field(default_factory=dict)
    # plans of the mask-grouped output-stationary route: over this rulebook's OUTPUT rows / over its
    # INPUT rows (= the transposed rulebook's outputs: grad-input); None: not planned
    osm: Optional["OsmPlanData"] = None
    osm_t: Optional["OsmPlanData"] = None

    @property
    def n_pairs(self) -> int:
        return int(self.kstart_host[-1])

    TILE_SIZES = TILE_SIZES

    def tiles(self, tile: int):
        """(device prefix int32[K+1], total, host prefix) of ceil(count_k / tile).  The device
        prefixes of all four tile sizes the kernels use come from ONE small launch per rulebook
        (pv2_tile_prefix on the device copy of ``kstart`` - no host->device copy, which would
        stall the host behind the queue); the totals come from the host copy of ``kstart``."""
        if not self._tiles:
            dev = self._tiles_dev if self._tiles_dev is not None else device_tile_prefixes(self.kstart, self.K)
            for i, t in enumerate(self.TILE_SIZES):
                host = np.zeros(self.K + 1, dtype=np.int64)
                np.cumsum((np.diff(self.kstart_host) + t - 1) // t, out=host[1:])
                self._tiles[t] = (dev[i], int(host[-1]), host)
        return self._tiles[tile]

    def __post_init__(self):
        # The device-side tile prefixes are launched HERE, on the stream that builds the rulebook
        # (the caller's, or the geometry side stream, which every consumer is ordered behind) - not
        # lazily at first use: the first user may be a weight gradient running on the backward side
        # stream, and a grad-input kernel on the main stream would then read the table before the
        # side stream has written it (found in round 3: wrong gradients below a 48-channel
        # inverse conv whenever the allocator handed out a dirty block).
        if self._tiles_dev is None and self.kstart is not None and self.kstart.is_cuda:
            self._tiles_dev = device_tile_prefixes(self.kstart, self.K)

    def transposed(self) -> "Rulebook":
        """Same pairs with the roles of input and output swapped (inverse conv / grad-input)."""
        rb = Rulebook(self.K, self.n_out, self.n_in, self.pair_out, self.pair_in, self.kstart,
                      self.kstart_host, self.center_k if self.n_in == self.n_out else -1,
                      _tiles_dev=self._tiles_dev)
        rb._tiles = self._tiles
        rb.bounded = self.bounded
        if self._transposed_os is not None:
            rb.nbr, rb.nbr_stride, rb.perm, rb.kflip = self._transposed_os
            rb._transposed_os = (self.nbr, self.nbr_stride, self.perm, self.kflip)
        rb.osm, rb.osm_t = self.osm_t, self.osm
        if self._pos is not None:
            po, so, pi, si = self._pos
            rb._pos = (pi, si, po, so)
        else:
            rb._pos_source = self   # built on demand on the parent, then mirrored
        return rb

    def positions(self):
        """(pos_out, out_stride, pos_in, in_stride), built on first use (one fill + one scatter
        launch per table; the U-Net's are built a batch ahead with the rulebooks)."""
        if self._pos is None:
            src = getattr(self, "_pos_source", None)
            if src is not None:
                po, so, pi, si = src.positions()
                self._pos = (pi, si, po, so)
            else:
                dev = self.kstart.device
                so, si = max(self.n_out, 1), max(self.n_in, 1)
                pos_out = torch.empty(self.K * so, dtype=torch.int32, device=dev)
                pos_in = torch.empty(self.K * si, dtype=torch.int32, device=dev)
                _lib.check(_lib.lib().pv2_pair_positions(
                    _ptr(self.pair_out), _ptr(self.pair_in), _ptr(self.kstart), self.K,
                    self.n_pairs, so, si, _ptr(pos_out), _ptr(pos_in), _stream(self.kstart)),
                    "pv2_pair_positions")
                self._pos = (pos_out, so, pos_in, si)
        return self._pos

    def geom(self, c_in: int, c_out: int, positions: bool = True):
        """The rulebook as the pv2_conv_geom struct of the fused conv + BatchNorm entry points
        (cached per weight-gradient chunk size; holds raw pointers into this rulebook's tensors).
        ``positions=False``: without the position tables (users that only need the pair lists and
        tile prefixes - the 125-offset stem's weight gradient)."""
        tile_w = int(_lib.lib().pv2_spconv_wgrad_tile(c_in, c_out, self.n_pairs, self.K))
        key = (tile_w, positions)
        g = self._geoms.get(key)
        if g is None:
            pos_out, so, pos_in, si = self.positions() if positions else (None, 0, None, 0)
            ts, n_tiles, _ = self.tiles(FWD_LDS_TILE)
            tsw, n_tiles_w, _ = self.tiles(tile_w)
            none = _lib.OsmPlan(None, None, None, 0, 0, 0)
            g = _lib.ConvGeom(self.K, tile_w, self.n_in, self.n_out, n_tiles, n_tiles_w, so, si,
                              self.pair_in.data_ptr(), self.pair_out.data_ptr(),
                              self.kstart.data_ptr(), ts.data_ptr(), tsw.data_ptr(),
                              pos_out.data_ptr() if positions else None,
                              pos_in.data_ptr() if positions else None,
                              self.osm.struct if (self.osm is not None and positions) else none,
                              self.osm_t.struct if (self.osm_t is not None and positions) else none,
                              zero_row(self.kstart.device).data_ptr())
            self._geoms[key] = g
        return g


def _compact(tbl: torch.Tensor, K: int, n: int, n_rows_dev: Optional[torch.Tensor]):
    """Ordered compaction of a [K, n] table; one host sync to size the pair arrays."""
    L = _lib.lib()
    dev = tbl.device
    nchunks = max(1, (n + SCAN_CHUNK - 1) // SCAN_CHUNK)
    block_sums = torch.empty(K * nchunks, dtype=torch.int32, device=dev)
    kstart = torch.empty(K + 1, dtype=torch.int32, device=dev)
    _lib.check(L.pv2_table_count(_ptr(tbl), K, n, _ptr(n_rows_dev), _ptr(block_sums),
                                 _ptr(kstart), _stream(tbl)), "pv2_table_count")
    kstart_host = kstart.cpu().numpy().astype(np.int64)  # the one sync of a rulebook build
    P = int(kstart_host[-1])
    pair_other = torch.empty(P, dtype=torch.int32, device=dev)
    pair_row = torch.empty(P, dtype=torch.int32, device=dev)
    _lib.check(L.pv2_table_compact(_ptr(tbl), K, n, _ptr(n_rows_dev), _ptr(block_sums),
                                   _ptr(pair_other), _ptr(pair_row), _stream(tbl)),
               "pv2_table_compact")
    return pair_other, pair_row, kstart, kstart_host


def _want_mask_order() -> bool:
    """The row grouping of the output-stationary kernel costs a mask pass and a device sort (~10
    launches) per table; it only pays when that kernel runs the bulk of the convs.  With the
    product-row path on (default) the output-stationary kernel is left with the stem and odd channel
    counts, which take rows in natural order."""
    return USE_PR != "all" or MASK_ORDER


# (the 16-bit training mode runs every conv output-stationary: trainers / bench.py --amp set this)
MASK_ORDER = os.environ.get("PV2_OS_MASK_ORDER", "0") == "1"


def _mask_order(tbl: torch.Tensor, K: int, n_cols: int, stride: int) -> Optional[torch.Tensor]:
    """Row order for the output-stationary kernel: rows sorted (stably) by the bit mask of their
    present offsets.  None (natural order) for windows wider than 63 offsets and tiny tables."""
    if K > 63 or n_cols < 64 or not _want_mask_order():
        return None
    mask = torch.empty(n_cols, dtype=torch.int64, device=tbl.device)
    _lib.check(_lib.lib().pv2_table_masks(_ptr(tbl), K, n_cols, stride, None, _ptr(mask),
                                          _stream(tbl)), "pv2_table_masks")
    return torch.argsort(mask, stable=True).to(torch.int32)


_RAMPS = {}


def build_subm_rulebook(coords: torch.Tensor, ksize: int) -> Rulebook:
    """coords int32 [N,4] (b,x,y,z) -> rulebook of a submanifold conv with an odd cubic kernel."""
    _require_device(coords)
    assert coords.dtype == torch.int32 and coords.dim() == 2 and coords.shape[1] == 4
    coords = coords.contiguous()
    n = coords.shape[0]
    dev = coords.device
    K = ksize ** 3
    if ksize == 1:
        # (rows 0..n-1 of one cached ramp per device, and [0, n] as ramp[:2] * n: one launch per
        # identity rulebook instead of three - four of them per forward pass)
        ramp = _RAMPS.get(dev)
        if ramp is None or ramp.numel() < max(n, 2):
            ramp = _RAMPS[dev] = torch.arange(max(2 * n, 1 << 16), dtype=torch.int32, device=dev)
        ar = ramp[:n]
        kh = np.array([0, n], dtype=np.int64)
        kstart = ramp[:2] * n
        rb = Rulebook(1, n, n, ar, ar, kstart, kh, center_k=0)
        rb.nbr, rb.nbr_stride = ar, n
        rb._transposed_os = (ar, n, None, 0)
        return rb
    L = _lib.lib()
    tsize = 1 << max(4, int(2 * max(n, 1) - 1).bit_length())
    keys = torch.empty(tsize, dtype=torch.int64, device=dev)
    vals = torch.empty(tsize, dtype=torch.int32, device=dev)
    _lib.check(L.pv2_hash_build(_ptr(coords), n, _ptr(keys), _ptr(vals), tsize, _stream(coords)),
               "pv2_hash_build")
    nbr = torch.empty(K * max(n, 1), dtype=torch.int32, device=dev)
    _lib.check(L.pv2_subm_neighbor_table(_ptr(coords), n, ksize, _ptr(keys), _ptr(vals), tsize,
                                         _ptr(nbr), _stream(coords)), "pv2_subm_neighbor_table")
    pair_in, pair_out, kstart, kstart_host = _compact(nbr, K, n, None)
    rb = Rulebook(K, n, n, pair_in, pair_out, kstart, kstart_host, center_k=K // 2)
    if n > 0:
        # the neighbour table is its own transpose up to mirroring the offsets (coordinates are
        # unique): grad-input reads the same table with the weight offsets flipped
        # (the mask sort is only worth its launches when the submanifold convs run output-stationary)
        rb.nbr, rb.nbr_stride = nbr, n
        rb.perm = _mask_order(nbr, K, n, n) if USE_OS is True else None
        rb._transposed_os = (nbr, n, rb.perm, 1)
        if _want_osm(K):
            rb.osm = build_osm_plan(nbr, K, n, n)
            rb.osm_t = rb.osm.flipped()
    return rb


def rulebook_from_table(tbl: torch.Tensor, K: int, n_in: int, n_out: int,
                        n_rows_dev: Optional[torch.Tensor] = None, bounded: bool = False) -> Rulebook:
    """tbl int32 [K, n_in]: the output row input row i feeds under offset k, or -1 -> rulebook in
    canonical (offset, input row) order.  Used for convolutions whose geometry is arithmetic (the
    dense grid's first layer, models/ponder/sparse_input.py) rather than hashed.

    ``bounded``: no device->host read.  The pair arrays are sized for the worst case (every row
    pairs under every offset), the host-side counts are that upper bound - they only size launch
    grids; the kernels take the true counts from the device copy of ``kstart`` and workgroups past
    them return at once.  ``n_rows_dev`` (int32 [1] on the device): rows past it are padding."""
    _require_device(tbl)
    assert tbl.dtype == torch.int32 and tbl.shape == (K, n_in)
    flat = tbl.contiguous().reshape(-1)
    if not bounded:
        pair_out, pair_in, kstart, kstart_host = _compact(flat, K, n_in, n_rows_dev)
        return Rulebook(K, n_in, n_out, pair_in, pair_out, kstart, kstart_host)
    L = _lib.lib()
    dev = tbl.device
    nchunks = max(1, (n_in + SCAN_CHUNK - 1) // SCAN_CHUNK)
    block_sums = torch.empty(K * nchunks, dtype=torch.int32, device=dev)
    kstart = torch.empty(K + 1, dtype=torch.int32, device=dev)
    _lib.check(L.pv2_table_count(_ptr(flat), K, n_in, _ptr(n_rows_dev), _ptr(block_sums),
                                 _ptr(kstart), _stream(tbl)), "pv2_table_count")
    pair_out = torch.empty(K * n_in, dtype=torch.int32, device=dev)
    pair_in = torch.empty(K * n_in, dtype=torch.int32, device=dev)
    _lib.check(L.pv2_table_compact(_ptr(flat), K, n_in, _ptr(n_rows_dev), _ptr(block_sums),
                                   _ptr(pair_out), _ptr(pair_in), _stream(tbl)), "pv2_table_compact")
    bound = np.arange(K + 1, dtype=np.int64) * n_in
    rb = Rulebook(K, n_in, n_out, pair_in, pair_out, kstart, bound)
    rb.bounded = True
    return rb


def build_downsample_rulebook(coords: torch.Tensor, stride: int, out_shape: List[int]):
    """Strided conv with kernel == stride, no padding.  Returns (rulebook, out_coords int32 [M,4]);
    out_coords are sorted by (b,x,y,z)."""
    _require_device(coords)
    assert coords.dtype == torch.int32 and coords.dim() == 2 and coords.shape[1] == 4
    coords = coords.contiguous()
    n = coords.shape[0]
    dev = coords.device
    L = _lib.lib()
    K = stride ** 3
    shape_c = (ctypes.c_int32 * 3)(*[int(s) for s in out_shape])
    keys_a = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
    keys_b = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
    out_coords = torch.empty((max(n, 1), 4), dtype=torch.int32, device=dev)
    n_out_dev = torch.zeros(1, dtype=torch.int32, device=dev)
    ws_bytes = int(L.pv2_downsample_workspace_bytes(n))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    st = _stream(coords)
    _lib.check(L.pv2_downsample_unique(_ptr(coords), n, stride, shape_c, _ptr(keys_a),
                                       _ptr(keys_b), _ptr(out_coords), _ptr(n_out_dev), _ptr(ws),
                                       ws_bytes, st), "pv2_downsample_unique")
    tbl = torch.empty(K * max(n, 1), dtype=torch.int32, device=dev)
    _lib.check(L.pv2_downsample_table(_ptr(coords), n, stride, shape_c, _ptr(keys_b),
                                      _ptr(n_out_dev), _ptr(tbl), n, st), "pv2_downsample_table")
    pair_in, pair_out, kstart, kstart_host = _compact(tbl, K, n, n_out_dev)
    n_out = int(n_out_dev.item())
    rb = Rulebook(K, n, n_out, pair_in, pair_out, kstart, kstart_host)
    if n > 0 and n_out > 0:
        # forward gathers the children of every output voxel (tbl, row stride n); the transposed
        # use (grad-input, inverse conv) gathers the one parent of every input voxel
        parent = torch.empty(K * n, dtype=torch.int32, device=dev)
        _lib.check(L.pv2_table_invert(_ptr(tbl), K, n_out, n, None, _ptr(parent), n, st),
                   "pv2_table_invert")
        rb.nbr, rb.nbr_stride, rb.perm = tbl, n, _mask_order(tbl, K, n_out, n)
        rb._transposed_os = (parent, n, _mask_order(parent, K, n, n), 0)
        if _want_osm(K):
            rb.osm = build_osm_plan(tbl, K, n_out, n)
            rb.osm_t = build_osm_plan(parent, K, n, n)
    return rb, out_coords[:n_out]


def prepare_unet_geometry(indices: torch.Tensor, spatial_shape, n_levels: int = 4,
                          stem_ksize: int = 5, stem_key: str = "stem") -> dict:
    """Every rulebook a SpUNet forward needs, built in ONE pass with ONE device->host read
    (``_launch_unet_geometry`` + a blocking copy + ``_finish_unet_geometry``; the non-blocking form
    is ``prefetch_unet_geometry``)."""
    launched = _launch_unet_geometry(indices, spatial_shape, n_levels, stem_ksize, stem_key)
    if launched is None:
        return {}
    state, counts = launched
    return _finish_unet_geometry(state, counts.cpu().numpy())


class PendingGeometry:
    """Rulebooks in flight on a side stream (``prefetch_unet_geometry``)."""

    def __init__(self, state, counts_host, event, stream, n_rows):
        self._state, self._host, self._event, self._stream = state, counts_host, event, stream
        self.n_rows = n_rows
        self._result = None

    def result(self) -> dict:
        """The ``indice_dict``; the host waits for the side stream's copy only (long done when
        the geometry was launched a step ahead), the current stream for its kernels."""
        if self._result is None:
            if self._state is None:
                self._result = {}
            else:
                self._event.synchronize()
                cur = torch.cuda.current_stream(self._state["dev"])
                cur.wait_event(self._event)
                self._result = _finish_unet_geometry(self._state, self._host.numpy())
                _record_stream(self._result, cur)   # allocated on the side stream, used on this one
                self._state = None
        return self._result


_SKIP_RECORD = os.environ.get("PV2_DEBUG_NO_RECORD_STREAM") == "1"   # timing experiments only: UNSAFE


def _record_stream(obj, stream, _seen=None):
    if _SKIP_RECORD:
        return
    seen = set() if _seen is None else _seen
    if id(obj) in seen:
        return
    seen.add(id(obj))
    if torch.is_tensor(obj):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            _record_stream(v, stream, seen)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _record_stream(v, stream, seen)
    elif isinstance(obj, Rulebook):
        # (element by element: a TEMPORARY list here would enter `seen` by its id, be freed on return, and
        # the next rulebook's temporary could be handed the same address - and be skipped whole, leaving
        # its tensors unrecorded: a use-after-free on the training stream, found in round 5 as a memory
        # fault of bench.py once the plans added more such temporaries)
        for v in vars(obj).values():
            _record_stream(v, stream, seen)
    elif isinstance(obj, OsmPlanData):
        for v in (obj.perm, obj.tblp, obj.tmask):
            _record_stream(v, stream, seen)


_GEOMETRY_STREAMS = {}


GEOMETRY_ON_INPUT_STREAM = os.environ.get("PV2_GEOMETRY_ON_INPUT", "1") != "0"


def _is_input_stream(dev, stream) -> bool:
    from .ponder.datasets import voxelize

    s = voxelize._INPUT_STREAMS.get(dev.index if dev.index is not None else torch.cuda.current_device())
    return s is not None and s == stream


def prefetch_unet_geometry(indices: torch.Tensor, spatial_shape, n_levels: int = 4,
                           stem_ksize: int = 5, stem_key: str = "stem") -> PendingGeometry:
    """``prepare_unet_geometry`` without stalling the host: the tables are built on a side stream
    (after whatever the current stream has queued so far - so launch it BEFORE the training step
    it should overlap with, i.e. one batch ahead) and their counts travel to pinned host memory
    asynchronously.  The geometry depends on the batch's coordinates only, never on the model: this
    is input-pipeline work, the counterpart of the reference's dataloader workers."""
    _require_device(indices)
    dev = indices.device
    cur = torch.cuda.current_stream(dev)
    if GEOMETRY_ON_INPUT_STREAM and _is_input_stream(dev, cur):
        # Round 6: called from the batch's staging (datasets.voxelize.input_stream) the tables are built ON
        # that stream - it is a side stream already, one batch ahead of the training stream.  A stream of
        # their own was one more HIP stream per rank (six with a process group, over four hardware queues:
        # the streams that share a queue serialise, +3 ms per step measured in round 5).
        launched = _launch_unet_geometry(indices, spatial_shape, n_levels, stem_ksize, stem_key)
        if launched is None:
            return PendingGeometry(None, None, None, cur, 0)
        state, counts = launched
        host = torch.empty(counts.shape, dtype=counts.dtype, pin_memory=True)
        host.copy_(counts, non_blocking=True)
        event = torch.cuda.Event()
        event.record(cur)
        return PendingGeometry(state, host, event, cur, indices.shape[0])
    side = _GEOMETRY_STREAMS.get(dev.index)
    if side is None:
        # HIGH priority by default: ~190 tiny dependent kernels that otherwise wait for a gap between
        # the training stream's kernels, one gap each - the tables of the next batch then arrive
        # 5-6 ms into its own step and the host stalls on their counts (round 4, tools/profile_host.py)
        prio = int(os.environ.get("PV2_GEOMETRY_PRIORITY", "-1"))
        side = _GEOMETRY_STREAMS[dev.index] = torch.cuda.Stream(device=dev, priority=prio)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        indices.record_stream(side)
        launched = _launch_unet_geometry(indices, spatial_shape, n_levels, stem_ksize, stem_key)
        if launched is None:
            return PendingGeometry(None, None, None, side, 0)
        state, counts = launched
        host = torch.empty(counts.shape, dtype=counts.dtype, pin_memory=True)
        host.copy_(counts, non_blocking=True)
        event = torch.cuda.Event()
        event.record(side)
    return PendingGeometry(state, host, event, side, indices.shape[0])


def _launch_unet_geometry(indices: torch.Tensor, spatial_shape, n_levels: int = 4,
                          stem_ksize: int = 5, stem_key: str = "stem"):
    """Launches every table build of a SpUNet forward; returns (state, device tensor of all counts)
    or None for an empty input.

    The lazy builders above read a count back per rulebook (to size its pair arrays) and the
    strided convs another one (the number of output voxels): ~14 blocking reads per forward, each
    of which stalls the host until the GPU has drained.  Here the four strided levels are chained
    on capacity-sized coordinate arrays (rows past the device-side count are padding with batch
    index -1, which the kernels skip), every table is compacted into worst-case-sized pair arrays,
    and all ``kstart`` prefixes and voxel counts come back in a single copy at the end.  Returns an
    ``indice_dict`` for ``SparseConvTensor``: ``stem`` (k5) and ``subm0..n`` (k3) submanifold
    entries, ``spconv1..n`` strided entries (with their output indices / shapes).  Identical
    rulebooks, bit for bit, to the lazy builders (tests/test_gpu_kernels.py)."""
    _require_device(indices)
    assert indices.dtype == torch.int32 and indices.dim() == 2 and indices.shape[1] == 4
    L = _lib.lib()
    dev = indices.device
    st = _stream(indices)
    cap = indices.shape[0]
    if cap == 0:
        return None
    coords = [indices.contiguous()]
    n_dev = [None]                      # device-side valid count of each level (None: all rows)
    shapes = [[int(v) for v in spatial_shape]]
    downs, readback = [], []

    tiles_dev = {}

    def compact(tbl, K, n_rows_dev, out_cap):
        nchunks = max(1, (cap + SCAN_CHUNK - 1) // SCAN_CHUNK)
        block_sums = torch.empty(K * nchunks, dtype=torch.int32, device=dev)
        kstart = torch.empty(K + 1, dtype=torch.int32, device=dev)
        _lib.check(L.pv2_table_count(_ptr(tbl), K, cap, _ptr(n_rows_dev), _ptr(block_sums),
                                     _ptr(kstart), st), "pv2_table_count")
        other = torch.empty(out_cap, dtype=torch.int32, device=dev)
        row = torch.empty(out_cap, dtype=torch.int32, device=dev)
        _lib.check(L.pv2_table_compact(_ptr(tbl), K, cap, _ptr(n_rows_dev), _ptr(block_sums),
                                       _ptr(other), _ptr(row), st), "pv2_table_compact")
        readback.append(kstart)
        tiles_dev[id(kstart)] = device_tile_prefixes(kstart, K)   # on this (the geometry) stream
        return other, row, kstart

    def order(tbl, K, n_cols_dev):
        if not _want_mask_order():
            return None
        mask = torch.empty(cap, dtype=torch.int64, device=dev)
        _lib.check(L.pv2_table_masks(_ptr(tbl), K, cap, cap, _ptr(n_cols_dev), _ptr(mask), st),
                   "pv2_table_masks")
        return torch.argsort(mask, stable=True).to(torch.int32)

    def positions(pair_in, pair_out, kstart, K, bound):
        """Position tables of the product-row conv path, capacity-sized (row stride ``cap``)."""
        if USE_PR is False or K > PR_MAX_K:
            return None
        pos_out = torch.empty(K * cap, dtype=torch.int32, device=dev)
        pos_in = torch.empty(K * cap, dtype=torch.int32, device=dev)
        _lib.check(L.pv2_pair_positions(_ptr(pair_out), _ptr(pair_in), _ptr(kstart), K, bound, cap,
                                        cap, _ptr(pos_out), _ptr(pos_in), st), "pv2_pair_positions")
        return (pos_out, cap, pos_in, cap)

    ws_bytes = int(L.pv2_downsample_workspace_bytes(cap))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    for l in range(1, n_levels + 1):
        out_shape = [(s - 2) // 2 + 1 for s in shapes[-1]]
        shape_c = (ctypes.c_int32 * 3)(*out_shape)
        keys_a = torch.empty(cap, dtype=torch.int64, device=dev)
        keys_b = torch.empty(cap, dtype=torch.int64, device=dev)
        out_coords = torch.empty((cap, 4), dtype=torch.int32, device=dev)
        n_out_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(L.pv2_downsample_unique(_ptr(coords[-1]), cap, 2, shape_c, _ptr(keys_a),
                                           _ptr(keys_b), _ptr(out_coords), _ptr(n_out_dev), _ptr(ws),
                                           ws_bytes, st), "pv2_downsample_unique")
        tbl = torch.empty(8 * cap, dtype=torch.int32, device=dev)
        _lib.check(L.pv2_downsample_table(_ptr(coords[-1]), cap, 2, shape_c, _ptr(keys_b),
                                          _ptr(n_out_dev), _ptr(tbl), cap, st), "pv2_downsample_table")
        pair_in, pair_out, kstart = compact(tbl, 8, n_out_dev, cap)
        parent = torch.empty(8 * cap, dtype=torch.int32, device=dev)
        _lib.check(L.pv2_table_invert(_ptr(tbl), 8, cap, cap, _ptr(n_out_dev), _ptr(parent), cap, st),
                   "pv2_table_invert")
        downs.append(dict(pair_in=pair_in, pair_out=pair_out, kstart=kstart, tbl=tbl, parent=parent,
                          perm=order(tbl, 8, n_out_dev), perm_t=order(parent, 8, n_dev[-1]),
                          out_shape=out_shape, pos=positions(pair_in, pair_out, kstart, 8, cap),
                          osm=build_osm_plan(tbl, 8, cap, cap, n_out_dev) if _want_osm(8) else None,
                          osm_t=build_osm_plan(parent, 8, cap, cap, n_dev[-1]) if _want_osm(8) else None))
        coords.append(out_coords)
        n_dev.append(n_out_dev)
        shapes.append(out_shape)
    subms = []
    tsize = 1 << max(4, int(2 * cap - 1).bit_length())
    for level in range(n_levels + 1):
        keys = torch.empty(tsize, dtype=torch.int64, device=dev)
        vals = torch.empty(tsize, dtype=torch.int32, device=dev)
        _lib.check(L.pv2_hash_build(_ptr(coords[level]), cap, _ptr(keys), _ptr(vals), tsize, st),
                   "pv2_hash_build")
        for ksize, key in ([(stem_ksize, stem_key)] if level == 0 else []) + [(3, f"subm{level}")]:
            K = ksize ** 3
            nbr = torch.empty(K * cap, dtype=torch.int32, device=dev)
            _lib.check(L.pv2_subm_neighbor_table(_ptr(coords[level]), cap, ksize, _ptr(keys),
                                                 _ptr(vals), tsize, _ptr(nbr), st),
                       "pv2_subm_neighbor_table")
            # a voxel pairs with at most its K window cells, and only valid rows pair at all; the
            # level-0 count is exact, coarser levels are bounded by the level above
            pair_in, pair_out, kstart = compact(nbr, K, None, K * cap if level == 0 else 27 * cap)
            subms.append(dict(key=key, ksize=ksize, level=level, K=K, nbr=nbr, pair_in=pair_in,
                              pair_out=pair_out, kstart=kstart,
                              pos=positions(pair_in, pair_out, kstart, K,
                                            K * cap if level == 0 else 27 * cap),
                              perm=order(nbr, K, n_dev[level]) if (USE_OS is True and K <= 63) else None,
                              osm=build_osm_plan(nbr, K, cap, cap, n_dev[level]) if _want_osm(K) else None))
    for d in downs + subms:
        d["tiles_dev"] = tiles_dev[id(d["kstart"])]
    state = dict(cap=cap, n_levels=n_levels, coords=coords, shapes=shapes, downs=downs, subms=subms,
                 readback=readback, dev=dev)
    return state, torch.cat([t.reshape(-1) for t in readback + n_dev[1:]])


def _finish_unet_geometry(state, host) -> dict:
    """Rulebooks from the launched tables and the host copy of their counts."""
    cap, n_levels, coords, shapes = state["cap"], state["n_levels"], state["coords"], state["shapes"]
    downs, subms, readback, dev = state["downs"], state["subms"], state["readback"], state["dev"]
    st = _stream(coords[0])  # (the caller's stream: the tables may have been built on another)
    L = _lib.lib()
    host = np.asarray(host).astype(np.int64)
    pos = 0
    hosts = []
    for t in readback:
        hosts.append(host[pos:pos + t.numel()])
        pos += t.numel()
    n_lvl = [cap] + [int(v) for v in host[pos:pos + n_levels]]
    out = {}
    for l, d in enumerate(downs, start=1):
        kh = hosts[l - 1]
        P = int(kh[-1])
        rb = Rulebook(8, n_lvl[l - 1], n_lvl[l], d["pair_in"][:P], d["pair_out"][:P], d["kstart"], kh,
                      _tiles_dev=d["tiles_dev"])
        if n_lvl[l] > 0:
            rb.nbr, rb.nbr_stride, rb.perm = d["tbl"], cap, d["perm"]
            rb._transposed_os = (d["parent"], cap, d["perm_t"], 0)
            if d.get("pos") is not None:
                rb._pos = d["pos"]
            rb.osm, rb.osm_t = d.get("osm"), d.get("osm_t")
        out[f"spconv{l}"] = dict(kind="down", ksize=2, rulebook=rb, in_indices=coords[l - 1][:n_lvl[l - 1]],
                                 in_spatial_shape=shapes[l - 1], out_indices=coords[l][:n_lvl[l]],
                                 out_shape=d["out_shape"], prebuilt=True)
    for d, kh in zip(subms, hosts[n_levels:]):
        n = n_lvl[d["level"]]
        P = int(kh[-1])
        rb = Rulebook(d["K"], n, n, d["pair_in"][:P], d["pair_out"][:P], d["kstart"], kh,
                      center_k=d["K"] // 2, _tiles_dev=d["tiles_dev"])
        if n > 0:
            rb.nbr, rb.nbr_stride, rb.perm = d["nbr"], cap, d["perm"]
            rb._transposed_os = (d["nbr"], cap, d["perm"], 1)
            if d.get("pos") is not None:
                rb._pos = d["pos"]
            if d.get("osm") is not None:
                rb.osm = d["osm"]
                rb.osm_t = rb.osm.flipped()
        out[d["key"]] = dict(kind="subm", ksize=d["ksize"], n=n, rulebook=rb)
    return out


# --------------------------------------------------------------------------------------------
# Sparse conv arithmetic
# --------------------------------------------------------------------------------------------
_FWD_TILE = {}


def _forward_tile(c_in: int, c_out: int) -> int:
    t = _FWD_TILE.get((c_in, c_out))
    if t is None:
        t = _FWD_TILE[(c_in, c_out)] = int(_lib.lib().pv2_spconv_forward_tile(c_in, c_out))
    return t


def _pr_conv(feats, weight, rb, c_in, c_out, reduction_major, pair_in, pos, pos_stride, n_rows,
             bias=None, addend=None, bn_partial=None):
    """The two stages of the product-row conv (csrc/sparse_conv_pr.hip): prod[p] = W[k(p)] .
    feats[pair_in[p]], then out[o] = (addend[o] + bias) + sum_k prod[pos[k, o]] in ascending k."""
    L = _lib.lib()
    dev = feats.device
    tile_start, n_tiles, _ = rb.tiles(FWD_LDS_TILE)
    prod = workspace("prod", dev, rb.n_pairs * c_out)
    st = _stream(feats)
    _lib.check(L.pv2_spconv_products(_ptr(feats), c_in, _ptr(weight), rb.K, c_out,
                                     int(reduction_major), _ptr(pair_in), _ptr(rb.kstart),
                                     _ptr(tile_start), n_tiles, _ptr(prod), st), "pv2_spconv_products")
    out = torch.empty((n_rows, c_out), dtype=torch.float32, device=dev)
    blocks = ctypes.c_int(0)
    _lib.check(L.pv2_spconv_reduce_rows(_ptr(prod), _ptr(pos), pos_stride, rb.K, c_out, n_rows,
                                        _ptr(bias), _ptr(addend), _ptr(out), _ptr(bn_partial),
                                        ctypes.byref(blocks), st), "pv2_spconv_reduce_rows")
    return out, blocks.value


def spconv_osm(feats: torch.Tensor, weight: torch.Tensor, rb: Rulebook, transposed: bool = False,
               addend: Optional[torch.Tensor] = None, stats: bool = False):
    """The mask-grouped output-stationary conv (pv2_spconv_osm) of ``feats`` over ``rb``'s plan.
    ``transposed``: grad-input - ``feats`` is grad_out [n_out, c_out], ``weight`` the FORWARD weight
    [c_out, K, c_in] read in place, the result [n_in, c_in].  ``stats``: also (partial, blocks,
    rows_per_block), the per-block BatchNorm moments."""
    _require_device(feats, weight)
    plan = rb.osm_t if transposed else rb.osm
    assert plan is not None, "rulebook carries no output-stationary plan"
    feats, weight = feats.contiguous(), weight.contiguous()
    c_out, Kw, c_in = weight.shape
    c_red, c_cols, n_rows = (c_out, c_in, rb.n_in) if transposed else (c_in, c_out, rb.n_out)
    assert feats.shape[1] == c_red and Kw == rb.K
    out = torch.empty((n_rows, c_cols), dtype=torch.float32, device=feats.device)
    partial = torch.empty(int(_lib.lib().pv2_bn_workspace_floats(c_cols)), dtype=torch.float32,
                          device=feats.device) if stats else None
    blocks, rpb = ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(_lib.lib().pv2_spconv_osm(
        _ptr(feats), c_red, _ptr(weight), rb.K, c_cols, int(transposed), ctypes.byref(plan.struct),
        n_rows, _ptr(zero_row(feats.device)), _ptr(addend), _ptr(out), _ptr(partial),
        ctypes.byref(blocks), ctypes.byref(rpb), _stream(feats)), "pv2_spconv_osm")
    if stats:
        return out, partial, blocks.value, rpb.value
    return out


def spconv_forward(feats: torch.Tensor, weight_okc: torch.Tensor, rb: Rulebook,
                   out: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[pair_out] += W[k] . feats[pair_in].  weight_okc: fp32 [c_out, K, c_in] contiguous.
    Without a tensor to accumulate onto, the product-row path (default) or the output-stationary
    kernel computes ``bias + conv`` and writes every element once; ``out`` given: it is added to
    (product-row path: read as the addend, the result is a new tensor)."""
    _require_device(feats, weight_okc)
    assert feats.dtype == torch.float32 and weight_okc.dtype == torch.float32
    feats = feats.contiguous()
    weight_okc = weight_okc.contiguous()
    c_out, K, c_in = weight_okc.shape
    assert K == rb.K and feats.shape == (rb.n_in, c_in), (weight_okc.shape, feats.shape, rb.K, rb.n_in)
    L = _lib.lib()
    if out is None and _use_pr(rb, c_in, c_out) and rb.n_out > 0:
        pos_out, so, _, _ = rb.positions()
        res, _ = _pr_conv(feats, weight_okc, rb, c_in, c_out, False, rb.pair_in, pos_out, so,
                          rb.n_out, bias=bias)
        return res
    if out is None and _use_os(rb):
        out = torch.empty((rb.n_out, c_out), dtype=torch.float32, device=feats.device)
        _lib.check(L.pv2_spconv_os_forward(
            _ptr(feats), rb.n_in, c_in, _ptr(weight_okc), K, c_out, _ptr(rb.nbr), rb.nbr_stride,
            _ptr(rb.perm), rb.kflip, _ptr(bias), _ptr(out), rb.n_out, _stream(feats)),
            "pv2_spconv_os_forward")
        return out
    assert bias is None or out is None, "bias is fused only by the atomic-free paths"
    tile = _forward_tile(c_in, c_out)
    tile_start, n_tiles, tile_host = rb.tiles(tile)
    c_lo = c_hi = 0
    if out is None:
        if USE_CENTER_STORE and tile == FWD_LDS_TILE and rb.center_k >= 0 and rb.n_out > 0:
            # the centre offset initialises every output row with plain stores (no zero-fill)
            c_lo, c_hi = int(tile_host[rb.center_k]), int(tile_host[rb.center_k + 1])
            out = torch.empty((rb.n_out, c_out), dtype=torch.float32, device=feats.device)
        else:
            out = zeros_by_kernel((rb.n_out, c_out), torch.float32, feats.device)
    _lib.check(L.pv2_spconv_forward(
        _ptr(feats), rb.n_in, c_in, _ptr(weight_okc), K, c_out, _ptr(rb.pair_in),
        _ptr(rb.pair_out), _ptr(rb.kstart), _ptr(tile_start), tile, n_tiles, c_lo, c_hi, _ptr(out),
        rb.n_out, _stream(feats)), "pv2_spconv_forward")
    return out if bias is None else out + bias


def spconv_backward_weight(feats: torch.Tensor, grad_out: torch.Tensor, rb: Rulebook,
                           c_out: int, tile: Optional[int] = None) -> torch.Tensor:
    """dW [c_out, K, c_in] for out = conv(feats, W): the deterministic two-stage form (partial slabs
    + ordered reduction, no atomics, nothing to clear) where the channel counts allow, else fp32
    atomics on a zero-filled dW."""
    _require_device(feats, grad_out)
    feats = feats.contiguous()
    grad_out = grad_out.contiguous()
    c_in = feats.shape[1]
    assert grad_out.shape == (rb.n_out, c_out) and feats.shape[0] == rb.n_in
    L = _lib.lib()
    if tile is None:
        tile = L.pv2_spconv_wgrad_tile(c_in, c_out, rb.n_pairs, rb.K)
    tile_start, n_tiles, _ = rb.tiles(tile)
    if USE_WGRAD_DET and c_in % 4 == 0 and c_out % 4 == 0 and os.environ.get("PV2_SPCONV_GENERIC") != "1":
        dw = torch.empty((c_out, rb.K, c_in), dtype=torch.float32, device=feats.device)
        part = workspace("wgrad", feats.device,
                         int(L.pv2_spconv_wgrad_partial_floats(c_in, c_out, n_tiles)))
        _lib.check(L.pv2_spconv_backward_weight_det(
            _ptr(feats), rb.n_in, c_in, _ptr(grad_out), rb.n_out, c_out, rb.K, _ptr(rb.pair_in),
            _ptr(rb.pair_out), _ptr(rb.kstart), _ptr(tile_start), tile, n_tiles, _ptr(part), _ptr(dw),
            _stream(feats)), "pv2_spconv_backward_weight_det")
        return dw
    dw = zeros_by_kernel((c_out, rb.K, c_in), torch.float32, feats.device)
    _lib.check(L.pv2_spconv_backward_weight(
        _ptr(feats), rb.n_in, c_in, _ptr(grad_out), rb.n_out, c_out, rb.K, _ptr(rb.pair_in),
        _ptr(rb.pair_out), _ptr(rb.kstart), _ptr(tile_start), tile, n_tiles, _ptr(dw),
        _stream(feats)), "pv2_spconv_backward_weight")
    return dw


def spconv_grad_input(grad_out: torch.Tensor, weight_okc: torch.Tensor, rb: Rulebook,
                      addend: Optional[torch.Tensor] = None) -> torch.Tensor:
    """d loss / d feats of ``out = conv(feats, W)`` (+ ``addend``, product-row path only): the conv
    of ``grad_out`` over the transposed rulebook with W^T.  The product-row and scatter-add paths
    read the forward weight [c_out, K, c_in] as it is stored (reduction-major for this pass) - no
    transposed copy per layer per step; the output-stationary kernels read 16-byte pieces along
    their reduction axis and take the materialised transpose."""
    rbt = rb.transposed()
    c_out, K, c_in = weight_okc.shape
    if _use_pr(rbt, c_out, c_in) and rbt.n_out > 0:
        _, _, pos_in, si = rb.positions()
        res, _ = _pr_conv(grad_out.contiguous(), weight_okc.contiguous(), rb, c_out, c_in, True,
                          rb.pair_out, pos_in, si, rb.n_in, addend=addend)
        return res
    assert addend is None
    scatter = not _use_os(rbt)
    if scatter and c_out % FWD_LDS_TILE_K == 0 and c_in % 4 == 0 and rbt.n_pairs > 0:
        grad_out = grad_out.contiguous()
        weight_okc = weight_okc.contiguous()
        L = _lib.lib()
        tile_start, n_tiles, _ = rbt.tiles(FWD_LDS_TILE)
        out = zeros_by_kernel((rbt.n_out, c_in), torch.float32, grad_out.device)
        _lib.check(L.pv2_spconv_forward_wt(
            _ptr(grad_out), rbt.n_in, c_out, _ptr(weight_okc), K, c_in, _ptr(rbt.pair_in),
            _ptr(rbt.pair_out), _ptr(rbt.kstart), _ptr(tile_start), FWD_LDS_TILE, n_tiles, _ptr(out),
            rbt.n_out, _stream(grad_out)), "pv2_spconv_forward_wt")
        return out
    return spconv_forward(grad_out, weight_okc.permute(2, 1, 0).contiguous(), rbt)


# --------------------------------------------------------------------------------------------
# 16-bit sparse conv (bf16 / fp16 features, fp32 master weights and accumulation)
# --------------------------------------------------------------------------------------------
def packed_weights(weight_okc: torch.Tensor, dtype: torch.dtype, cache: Optional[dict] = None):
    """(forward, grad-input) copies of an fp32 master weight [c_out, K, c_in] in ``dtype`` and MFMA
    fragment order (pv2_spconv16_pack_weights).  ``cache`` (a dict owned by the module that owns
    the parameter) keeps them until the parameter changes - its version counter moves with every
    in-place update - so they are packed once per optimiser step."""
    stamp = (weight_okc._version, weight_okc.data_ptr(), tuple(weight_okc.shape), dtype)
    if cache is not None and cache.get("stamp") == stamp:
        return cache["fwd"], cache["bwd"]
    L = _lib.lib()
    w = weight_okc.detach().contiguous()
    c_out, K, c_in = w.shape
    fwd = torch.empty(int(L.pv2_spconv16_packed_elems(c_out, K, c_in)), dtype=dtype, device=w.device)
    bwd = torch.empty(int(L.pv2_spconv16_packed_elems(c_in, K, c_out)), dtype=dtype, device=w.device)
    _lib.check(L.pv2_spconv16_pack_weights(_ptr(w), c_out, K, c_in, DTYPE_CODE[dtype], _ptr(fwd),
                                           _ptr(bwd), _stream(w)), "pv2_spconv16_pack_weights")
    if cache is not None:
        cache.update(stamp=stamp, fwd=fwd, bwd=bwd)
    return fwd, bwd


def spconv16_supported(feats: torch.Tensor, weight_okc: torch.Tensor, rb: Rulebook) -> bool:
    """The 16-bit kernels cover 16-bit device features with channel counts that are multiples of
    8 on a rulebook that carries its gather tables (every rulebook the builders above make)."""
    return (feats.is_cuda and feats.dtype in HALF_DTYPES and weight_okc.dtype == torch.float32
            and weight_okc.shape[0] % 8 == 0 and weight_okc.shape[2] % 8 == 0
            and rb.nbr is not None and rb._transposed_os is not None)


def spconv16_forward(feats, packed, K, c_out, nbr, nbr_stride, perm, kflip, n_out,
                     bias: Optional[torch.Tensor] = None, n_pairs: int = 0) -> torch.Tensor:
    """Output-stationary conv of 16-bit ``feats`` [n_in, c_in] with a packed weight.  (``n_pairs``:
    the rulebook's pair count, for flop accounting by instrumentation only.)"""
    _require_device(feats, packed)
    feats = feats.contiguous()
    out = torch.empty((n_out, c_out), dtype=feats.dtype, device=feats.device)
    _lib.check(_lib.lib().pv2_spconv16_os_forward(
        _ptr(feats), feats.shape[0], feats.shape[1], _ptr(packed), K, c_out, DTYPE_CODE[feats.dtype],
        _ptr(nbr), nbr_stride, _ptr(perm), kflip, _ptr(bias), _ptr(out), n_out, _stream(feats)),
        "pv2_spconv16_os_forward")
    return out


def spconv16_backward_weight(feats, grad_out, rb: Rulebook, c_out: int) -> torch.Tensor:
    """fp32 dW [c_out, K, c_in] from 16-bit features and output gradients."""
    _require_device(feats, grad_out)
    feats, grad_out = feats.contiguous(), grad_out.contiguous()
    assert feats.dtype == grad_out.dtype and feats.dtype in HALF_DTYPES
    c_in = feats.shape[1]
    dw = zeros_by_kernel((c_out, rb.K, c_in), torch.float32, feats.device)
    tile_start, n_tiles, _ = rb.tiles(WGRAD_TILE)
    _lib.check(_lib.lib().pv2_spconv16_backward_weight(
        _ptr(feats), rb.n_in, c_in, _ptr(grad_out), rb.n_out, c_out, DTYPE_CODE[feats.dtype], rb.K,
        _ptr(rb.pair_in), _ptr(rb.pair_out), _ptr(rb.kstart), _ptr(tile_start), WGRAD_TILE, n_tiles,
        _ptr(dw), _stream(feats)), "pv2_spconv16_backward_weight")
    return dw


def _weight_grad(fn, weight_view, *reads):
    """``fn()`` (a weight-gradient launch reading ``reads``) on the backward side stream when that
    is on and safe for this leaf (sidestream.py), on the current stream otherwise.  ``reads`` holds
    EVERYTHING the launch touches until the streams join - the rulebook (its pair lists) included:
    autograd drops a node's context as soon as the node has run, long before the side stream has."""
    from . import sidestream

    if sidestream.active(reads[0]) and sidestream.safe_leaf(weight_view):
        return sidestream.fork(fn, reads)
    return fn()


class SparseConv16Function(torch.autograd.Function):
    """The sparse conv of ``SparseConvFunction`` on 16-bit features: fp32 master weight in, 16-bit
    features out, fp32 weight gradient; forward and grad-input on the output-stationary kernel."""

    @staticmethod
    def forward(ctx, feats, weight_okc, rb: Rulebook, bias, cache=None):
        c_out, K, c_in = weight_okc.shape
        assert K == rb.K and feats.shape == (rb.n_in, c_in), (weight_okc.shape, feats.shape, rb.K, rb.n_in)
        fwd, ctx.packed_bwd = packed_weights(weight_okc, feats.dtype, cache)
        ctx.rb = rb
        ctx.has_bias = bias is not None
        ctx.save_for_backward(feats, weight_okc)
        return spconv16_forward(feats, fwd, K, c_out, rb.nbr, rb.nbr_stride, rb.perm, rb.kflip,
                                rb.n_out, bias, n_pairs=rb.n_pairs)

    @staticmethod
    def backward(ctx, grad_out):
        feats, weight_okc = ctx.saved_tensors
        rb = ctx.rb
        c_out, K, c_in = weight_okc.shape
        grad_out = grad_out.contiguous()
        g_feats = g_w = g_b = None
        if ctx.needs_input_grad[1]:   # first: it leaves for the side stream (sidestream.py)
            g_w = _weight_grad(lambda: spconv16_backward_weight(feats, grad_out, rb, c_out),
                               weight_okc, grad_out, feats, rb)
        if ctx.needs_input_grad[0]:
            nbr, stride, perm, kflip = rb._transposed_os
            g_feats = spconv16_forward(grad_out, ctx.packed_bwd, K, c_in, nbr, stride, perm, kflip,
                                       rb.n_in, n_pairs=rb.n_pairs)
        if ctx.has_bias and ctx.needs_input_grad[3]:
            g_b = grad_out.float().sum(0)
        return g_feats, g_w, None, g_b, None


class SparseConvFunction(torch.autograd.Function):
    """Differentiable sparse conv on a fixed rulebook (features and weight get gradients)."""

    @staticmethod
    def forward(ctx, feats, weight_okc, rb: Rulebook):
        ctx.rb = rb
        ctx.save_for_backward(feats, weight_okc)
        return spconv_forward(feats, weight_okc, rb)

    @staticmethod
    def backward(ctx, grad_out):
        feats, weight_okc = ctx.saved_tensors
        rb = ctx.rb
        grad_out = grad_out.contiguous()
        g_feats = g_w = None
        if ctx.needs_input_grad[1]:   # first: it leaves for the side stream (sidestream.py)
            g_w = _weight_grad(lambda: spconv_backward_weight(feats, grad_out, rb, weight_okc.shape[0]),
                               weight_okc, grad_out, feats, rb)
        if ctx.needs_input_grad[0]:
            g_feats = spconv_grad_input(grad_out, weight_okc, rb)
        return g_feats, g_w, None


class SparseConvIntoFunction(torch.autograd.Function):
    """``init + conv(feats)``, accumulated in place into ``init`` (which must be a fresh tensor:
    it is returned as the result).  Lets a caller seed the output with something other than zeros
    without paying a second pass over it."""

    @staticmethod
    def forward(ctx, feats, weight_okc, rb: Rulebook, init):
        assert init.is_contiguous() and init.shape == (rb.n_out, weight_okc.shape[0])
        ctx.rb = rb
        ctx.save_for_backward(feats, weight_okc)
        spconv_forward(feats, weight_okc, rb, out=init)
        ctx.mark_dirty(init)
        return init

    @staticmethod
    def backward(ctx, grad_out):
        feats, weight_okc = ctx.saved_tensors
        rb = ctx.rb
        grad_out = grad_out.contiguous()
        g_feats = g_w = None
        # (on the current stream: the one caller, sparse_input.conv3d_on_cells, uses its weight a
        # second time for the constant part - autograd ADDS the two weight gradients on the main
        # stream the moment both exist, so this one must be complete there, not in flight)
        if ctx.needs_input_grad[1]:
            g_w = spconv_backward_weight(feats, grad_out, rb, weight_okc.shape[0])
        if ctx.needs_input_grad[0]:
            g_feats = spconv_grad_input(grad_out, weight_okc, rb)
        return g_feats, g_w, None, grad_out


# --------------------------------------------------------------------------------------------
# Dense scatter (to_dense)
# --------------------------------------------------------------------------------------------
class ScatterRowsFunction(torch.autograd.Function):
    """out[index[i]] (+)= src[i], optionally averaged over the rows landing in each bin; IN PLACE on
    ``out`` like ``torch_scatter.scatter(..., out=out)`` (whose sum / mean are ``out.scatter_add_``
    underneath).  ``out`` is the FIRST input on purpose: the reference passes a VIEW
    (``fea_grid[i] = scatter(feat, idx, out=fea_grid[i])``, ponder_indoor_base.py:214) and autograd
    rebases an in-place op on a view through ``CopySlices``, which routes the gradient of input 0 -
    and only input 0 - back into the base.  With ``src`` first, the row gradients would have been
    copied over the grid gradient's slice (a size-mismatch error, or silently wrong numbers)."""

    @staticmethod
    def forward(ctx, out, src, index, mean: bool):
        _require_device(src, index, out)
        src = src.contiguous()
        index = index.reshape(-1).contiguous()
        assert src.dtype == torch.float32 and index.dtype == torch.int64
        assert out.is_contiguous() and out.dim() == 2 and out.shape[1] == src.shape[1]
        m, c = src.shape
        g = out.shape[0]
        L = _lib.lib()
        count = torch.zeros(g, dtype=torch.float32, device=src.device) if mean else None
        _lib.check(L.pv2_scatter_add(_ptr(src), _ptr(index), m, c, _ptr(out), _ptr(count), g,
                                     _stream(src)), "pv2_scatter_add")
        if mean:
            _lib.check(L.pv2_scatter_mean_finish(_ptr(out), _ptr(count), g, c, _stream(src)),
                       "pv2_scatter_mean_finish")
        ctx.save_for_backward(index, count if mean else index)
        ctx.mean = mean
        ctx.shape = (m, c, g)
        ctx.mark_dirty(out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        index, count = ctx.saved_tensors
        m, c, g = ctx.shape
        grad_out = grad_out.contiguous()
        dsrc = None
        if ctx.needs_input_grad[1]:
            dsrc = torch.empty((m, c), dtype=torch.float32, device=grad_out.device)
            _lib.check(_lib.lib().pv2_scatter_backward(
                _ptr(grad_out), _ptr(index), _ptr(count) if ctx.mean else None, m, c, _ptr(dsrc), g,
                _stream(grad_out)), "pv2_scatter_backward")
        dout = None
        if ctx.needs_input_grad[0]:   # what was in ``out`` before: summed as is / divided like the rows
            dout = grad_out / count.clamp(min=1.0)[:, None] if ctx.mean else grad_out
        return dout, dsrc, None, None


# --------------------------------------------------------------------------------------------
# Trilinear sampler
# --------------------------------------------------------------------------------------------
_PADDING = {"zeros": 0, "border": 1, "reflection": 2}


def _vol_desc(x: torch.Tensor) -> VolumeDesc:
    n, c, d, h, w = x.shape
    sn, sc, sd, sh, sw = x.stride()
    return VolumeDesc(n, c, d, h, w, sn, sc, sd, sh, sw)


def _dense_like(x: torch.Tensor) -> bool:
    """True when x's storage is a permutation of a dense block (no overlap, no holes)."""
    return x.is_contiguous() or x.is_contiguous(memory_format=torch.channels_last_3d)


def _out_layout(n, c, do, ho, wo, like: torch.Tensor, channels_last: bool):
    """Allocate an (N,C,Do,Ho,Wo) tensor; channels-last storage keeps a point's channels adjacent
    so the kernels' per-point lane sweep is contiguous."""
    if channels_last:
        buf = torch.empty((n, do, ho, wo, c), dtype=like.dtype, device=like.device)
        return buf.permute(0, 4, 1, 2, 3)
    return torch.empty((n, c, do, ho, wo), dtype=like.dtype, device=like.device)


def _pts_desc(out_like: torch.Tensor) -> PointsDesc:
    n, c, do, ho, wo = out_like.shape
    sn, sc, sd, sh, sw = out_like.stride()
    ppn = do * ho * wo
    # point q = (d*Ho + h)*Wo + w must be addressable with a single stride
    assert _single_point_stride(out_like), "output-shaped tensor needs a uniform point stride"
    return PointsDesc(n * ppn, ppn, sn, sc, sw)


def _single_point_stride(t: torch.Tensor) -> bool:
    n, c, do, ho, wo = t.shape
    sn, sc, sd, sh, sw = t.stride()
    ok_h = ho == 1 or sh == sw * wo
    ok_d = do == 1 or sd == sw * wo * ho
    return ok_h and ok_d


def _fn(name: str, dtype: torch.dtype):
    if dtype == torch.float32:
        return getattr(_lib.lib(), name + "_f32")
    if dtype == torch.float64:
        return getattr(_lib.lib(), name + "_f64")
    raise TypeError(f"trilinear sampler supports float64/float32/float16/bfloat16, got {dtype}")


def _same_dtype(what, ref, *tensors):
    for t in tensors:
        if t is not None and t.dtype != ref.dtype:
            raise TypeError(f"{what}: expected every tensor in {ref.dtype}, got {t.dtype} "
                            "(the reference's extension checks scalar types the same way)")


def trilinear_forward(inp, grid, padding_mode="zeros", align_corners=True, smooth=False):
    _require_device(inp, grid)
    assert inp.dim() == 5 and grid.dim() == 5 and grid.shape[-1] == 3 and grid.shape[0] == inp.shape[0]
    if not _dense_like(inp):
        inp = inp.contiguous()
    grid = grid.contiguous()
    n, c = inp.shape[:2]
    _, do, ho, wo, _ = grid.shape
    out = _out_layout(n, c, do, ho, wo, inp, channels_last=True)
    vd, pd = _vol_desc(inp), _pts_desc(out)
    _same_dtype("trilinear_forward", inp, grid)
    if inp.dtype in HALF_DTYPES:   # 16-bit storage, fp32 arithmetic (smooth_sampler_kernel.cu:630)
        _lib.check(_lib.lib().pv2_trilinear_forward_16(
            _ptr(inp), DTYPE_CODE[inp.dtype], ctypes.byref(vd), _ptr(grid), ctypes.byref(pd),
            _ptr(out), _PADDING[padding_mode], int(align_corners), int(smooth), _stream(inp)),
            "pv2_trilinear_forward_16")
        return out
    _lib.check(_fn("pv2_trilinear_forward", inp.dtype)(
        _ptr(inp), ctypes.byref(vd), _ptr(grid), ctypes.byref(pd), _ptr(out),
        _PADDING[padding_mode], int(align_corners), int(smooth), _stream(inp)),
        "pv2_trilinear_forward")
    return out


def trilinear_backward(grad_out, inp, grid, padding_mode, align_corners, smooth, need_input_grad):
    _require_device(grad_out, inp, grid)
    if not _dense_like(inp):
        inp = inp.contiguous()
    grid = grid.contiguous()
    if not _single_point_stride(grad_out):
        grad_out = grad_out.contiguous()
    grad_grid = torch.empty_like(grid)
    _same_dtype("trilinear_backward", inp, grid, grad_out)
    half = inp.dtype in HALF_DTYPES
    # preserves inp's strides; the 16-bit path accumulates the volume gradient in fp32
    grad_in = torch.zeros_like(inp, dtype=torch.float32 if half else None) if need_input_grad else None
    if grad_in is not None:
        assert grad_in.stride() == inp.stride()
    vd, pd = _vol_desc(inp), _pts_desc(grad_out)
    if half:
        _lib.check(_lib.lib().pv2_trilinear_backward_16(
            _ptr(grad_out), _ptr(inp), DTYPE_CODE[inp.dtype], ctypes.byref(vd), _ptr(grid),
            ctypes.byref(pd), _ptr(grad_in), _ptr(grad_grid), _PADDING[padding_mode],
            int(align_corners), int(smooth), _stream(inp)), "pv2_trilinear_backward_16")
        return (None if grad_in is None else grad_in.to(inp.dtype)), grad_grid
    _lib.check(_fn("pv2_trilinear_backward", inp.dtype)(
        _ptr(grad_out), _ptr(inp), ctypes.byref(vd), _ptr(grid), ctypes.byref(pd), _ptr(grad_in),
        _ptr(grad_grid), _PADDING[padding_mode], int(align_corners), int(smooth), _stream(inp)),
        "pv2_trilinear_backward")
    return grad_in, grad_grid


def trilinear_backward_backward(g_ginput, g_ggrid, inp, grid, grad_out, padding_mode,
                                align_corners, smooth, need_input_grad):
    _require_device(g_ggrid, inp, grid, grad_out)
    if not _dense_like(inp):
        inp = inp.contiguous()
    grid = grid.contiguous()
    g_ggrid = g_ggrid.contiguous()
    if not _single_point_stride(grad_out):
        grad_out = grad_out.contiguous()
    if g_ginput is not None and g_ginput.stride() != inp.stride():
        tmp = torch.empty_like(inp)
        tmp.copy_(g_ginput)
        g_ginput = tmp
    _same_dtype("trilinear_backward_backward", inp, grid, grad_out, g_ggrid, g_ginput)
    half = inp.dtype in HALF_DTYPES
    grad_in2 = torch.zeros_like(inp, dtype=torch.float32 if half else None) if need_input_grad else None
    grad_grid2 = torch.empty_like(grid)
    gg_out = torch.empty_like(grad_out)  # same strides as grad_out (dense permutation)
    assert gg_out.stride() == grad_out.stride()
    vd, pd = _vol_desc(inp), _pts_desc(grad_out)
    if half:
        _lib.check(_lib.lib().pv2_trilinear_backward_backward_16(
            _ptr(g_ginput), _ptr(g_ggrid), _ptr(inp), DTYPE_CODE[inp.dtype], ctypes.byref(vd),
            _ptr(grid), _ptr(grad_out), ctypes.byref(pd), _ptr(grad_in2), _ptr(grad_grid2),
            _ptr(gg_out), _PADDING[padding_mode], int(align_corners), int(smooth), _stream(inp)),
            "pv2_trilinear_backward_backward_16")
        return (None if grad_in2 is None else grad_in2.to(inp.dtype)), grad_grid2, gg_out
    _lib.check(_fn("pv2_trilinear_backward_backward", inp.dtype)(
        _ptr(g_ginput), _ptr(g_ggrid), _ptr(inp), ctypes.byref(vd), _ptr(grid), _ptr(grad_out),
        ctypes.byref(pd), _ptr(grad_in2), _ptr(grad_grid2), _ptr(gg_out),
        _PADDING[padding_mode], int(align_corners), int(smooth), _stream(inp)),
        "pv2_trilinear_backward_backward")
    return grad_in2, grad_grid2, gg_out
