"""Ray set-up of the indoor model on csrc/ray_setup.hip: ``PonderIndoor.to_unit_cube`` +
``ray_sample`` (ponder/models/ponder/ponder_indoor_base.py:344-470 of the reference) as four launches
plus the library's random pixel choice, instead of ~140 small batched torch launches (2 ms of host
time per step).  Same arithmetic in the same order (fp32 without contraction, the slab test in
double); the torch statement in ponder_indoor_base.py stays as the route for every other input (host
tensors, other dtypes) and as what tests compare this one with."""
import ctypes
import os

import torch

from . import _lib
from .kernels import _ptr, _stream

ENABLED = os.environ.get("PV2_FUSED_RAY_SETUP", "1") != "0"
CALLS = 0
_REC = None


def _records():
    global _REC
    if _REC is None:
        a, b = ctypes.c_int(), ctypes.c_int()
        _lib.check(_lib.lib().pv2_ray_setup_record_sizes(ctypes.byref(a), ctypes.byref(b)),
                   "pv2_ray_setup_record_sizes")
        _REC = (a.value, b.value)
    return _REC


def usable(model, data_dict):
    if not ENABLED:
        return False
    need = ("coord", "offset", "rgb", "depth", "intrinsic", "extrinsic", "depth_scale")
    if any(k not in data_dict for k in need):
        return False
    f32 = lambda t: t.is_cuda and t.dtype == torch.float32
    d = data_dict
    if not (all(f32(d[k]) for k in ("coord", "rgb", "depth", "intrinsic", "extrinsic", "depth_scale"))
            and d["offset"].is_cuda and d["depth"].dim() == 4 and d["rgb"].dim() == 5):
        return False
    B, V = d["depth"].shape[:2]
    if not (1 <= B <= 64 and B * V <= 256 and d["offset"].numel() == B and d["extrinsic"].shape[:2] == (B, V)):
        return False
    if model.render_semantic and ("semantic" not in d or d["semantic"].dtype != torch.int64):
        return False
    return True


@torch.no_grad()
def prepare_ray(model, data_dict):
    """-> (ray_dict, data_dict), as ``PonderIndoor.prepare_ray``."""
    global CALLS
    CALLS += 1
    lib = _lib.lib()
    d = data_dict
    dev = d["coord"].device
    coords = d["coord"].contiguous()
    offset = d["offset"].to(torch.int64).contiguous()
    depths = d["depth"].contiguous()
    colors = d["rgb"].contiguous()
    B, V, H, W = depths.shape
    n = model.ray_nsample
    intr = d["intrinsic"]
    if intr.dim() == 3:
        intr = intr[:, None].expand(B, V, *intr.shape[-2:])
    kmat = intr[..., :3, :3].contiguous()
    extr_in = d["extrinsic"].contiguous()
    dscale = d["depth_scale"].contiguous()
    n_scene, n_view = _records()
    new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    ws, scene, view = new(B, 6), new(B, n_scene), new(B * V, n_view)
    extr_out, coords_out = new(B, V, 4, 4), new(*coords.shape)
    st = _stream(coords)
    _lib.check(lib.pv2_unit_cube(_ptr(coords), _ptr(offset), B, coords.shape[0], V, -0.5, _ptr(dscale),
                                 _ptr(extr_in), _ptr(kmat), _ptr(ws), _ptr(scene), _ptr(extr_out),
                                 _ptr(view), _ptr(coords_out), st), "pv2_unit_cube")
    d["extrinsic"] = extr_out
    d["depth_scale"] = scene[:, 5]
    d["pc_scale"] = scene[:, 4]
    d["bbox"] = scene[:, 6:12].reshape(B, 2, 3)
    d["coord"] = coords_out
    d["ray_setup_records"] = (scene, view)   # (scene / view records of csrc/ray_setup.hip, kept with the batch)

    pix = d.get("ray_pixels")   # optional (B,V,n,2) [y,x] from the caller
    if pix is None:
        flat = model._choose_pixels(depths > 0, n)
    else:
        flat = (pix[..., 0].long() * W + pix[..., 1].long()).to(dev)
    flat = flat.contiguous()
    sem_in = table = None
    if model.render_semantic:
        table = model._semantic_table(d, dev)
        sem_in = d["semantic"].contiguous()
    R = B * V * n
    ray_o, ray_d, rgb, depth = new(B, V * n, 3), new(B, V * n, 3), new(R, 3), new(R, 1)
    sem_row = torch.empty(R, dtype=torch.int64, device=dev) if sem_in is not None else None
    lo = (ctypes.c_double * 3)(*[float(v) for v in model.bounds[0]])
    hi = (ctypes.c_double * 3)(*[float(v) for v in model.bounds[1]])
    _lib.check(lib.pv2_ray_gen(_ptr(flat), B, V, n, H, W, _ptr(view), _ptr(scene), _ptr(colors),
                               _ptr(depths), _ptr(sem_in), lo, hi, _ptr(ray_o), _ptr(ray_d), _ptr(rgb),
                               _ptr(depth), _ptr(sem_row), st), "pv2_ray_gen")
    ray_dict = dict(ray_o=ray_o, ray_d=ray_d, rgb=rgb, depth=depth)
    if sem_row is not None:
        ray_dict["semantic"] = table[sem_row].float()
    return ray_dict, d
