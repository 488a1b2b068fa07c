"""SpUNet's forward and backward pass as one native call each (csrc/spunet_exec.hip).

``SpUNetBase.forward`` of the reference (ponder/models/sparse_unet/spconv_unet_v1m1_base.py:242-278)
walks ~60 conv + BatchNorm modules in Python; every one of them costs a Python call, an autograd
node and ~5 small launches per direction, and on MI355X the training step was HOST-bound on exactly
that (the host needed ~27 ms to enqueue a 28.9 ms step, ~10 ms of it inside the backbone).  Here the
walk happens once per forward to build a flat plan - one ``pv2_unet_op`` record per unit or skip
concatenation, with pointers into two arenas (activations; gradients) - and the launches happen in
C++ (``pv2_unet_forward`` / ``pv2_unet_backward``), ~4 us each.  One autograd node stands for the
whole backbone: its inputs are the stem's (padded) input features and every unit's weight and
BatchNorm affine pair, its output the backbone's output features.

Same kernels, same order, same numbers as the per-unit path (convbn.py) up to the order in which the
gradients of an activation with several consumers are added - here inside the grad-input row reduce
(its addend), there by autograd's add kernels.  Anything the plan cannot express takes the modular
path: eval mode, units the product-row path does not cover (channel counts that are no multiples of 32).

Round 6: the reduced-precision training mode (the reference's ``enable_amp = True``,
configs/scannet/pretrain-ponder-spunet-v1m1-0-base.py:12; precision.py) runs through the same executor:
under ``precision.sparse_dtype()`` every unit behind the stem is a ``UNET_CONV_BN16`` record - the 16-bit
output-stationary conv of csrc/sparse_conv16.hip on packed weights, the mixed-type BatchNorm, 16-bit
activation and gradient arenas; statistics, master weights and every parameter gradient stay fp32.  The
module-by-module walk of that mode was HOST-bound (18.4 - 21.2 ms per step at bs = 2, slower than fp32).
"""
import ctypes
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, kernels as K, precision, rownorm, sidestream
from ._lib import UNET_CONCAT, UNET_CONV_BN, UNET_CONV_BN16, UNET_STEM, UnetOp

ENABLED = os.environ.get("PV2_NATIVE_UNET", "1") != "0"
NATIVE16 = os.environ.get("PV2_NATIVE_UNET16", "1") != "0"   # 0: the 16-bit mode walks the modules (A / B)
# A gradient reducer that wants the parameter-gradient arena from INSIDE the backward node
# (ponder/utils/grad_sync.py FlatGradSync(overlap=True).attach()): an object with ``wants(tensors)``,
# ``slab_elems`` and ``_on_arena(arena, members, slabs, events)``.
GRAD_SLAB_HOOK = None
_EVENT_POOL = {}
CALLS = 0     # forward passes that ran natively (tests assert the path is the one being measured)
_ALIGN = 64   # floats: every carved buffer starts on a 256-byte boundary


class _Arena:
    """Bump allocator over one device tensor (sizes in floats are collected first)."""

    def __init__(self):
        self.size = 0
        self.base = 0
        self.tensor = None

    def reserve(self, floats):
        off = self.size
        self.size += (int(floats) + _ALIGN - 1) // _ALIGN * _ALIGN
        return off

    def allocate(self, device):
        self.tensor = torch.empty(max(self.size, 1), dtype=torch.float32, device=device)
        self.base = self.tensor.data_ptr()

    def ptr(self, off):
        return self.base + 4 * off

    def view(self, off, *shape):
        n = 1
        for s in shape:
            n *= s
        return self.tensor[off:off + n].view(*shape)

    def views(self, specs):
        """Views for many (offset, numel) pieces at once - ONE split call instead of two small
        tensor ops per piece (177 pieces per backward pass).  ``specs`` in ascending offset order."""
        sizes, pick, pos = [], [], 0
        for off, n in specs:
            if off > pos:
                sizes.append(off - pos)
            pick.append(len(sizes))
            sizes.append(n)
            pos = off + n
        if pos < self.tensor.numel():
            sizes.append(self.tensor.numel() - pos)
        parts = self.tensor.split_with_sizes(sizes)
        return [parts[i] for i in pick]


class _Unit:
    __slots__ = ("kind", "conv", "bn", "rb", "geom", "geom_ptr", "c_in", "c_out", "relu", "src", "dst",
                 "res", "w_index", "affine", "n_in", "n_out", "y_off", "mi_off", "dy_off", "gsum_off",
                 "dw_off", "acc_dx", "acc_res", "K", "dx_producer", "affine_in_arena")


class Plan:
    """The flat description of one forward pass: units, activation table, arenas."""

    def __init__(self):
        self.units = []
        self.acts = []        # (rows, channels) per activation id; id 0 = the stem's input
        self.tensors = []     # autograd inputs after the features: per unit weight, bn weight, bn bias
        self.fwd = _Arena()
        self.act_off = []
        self.ops = None
        self.dtype = None     # torch.bfloat16 / torch.float16: the 16-bit mode (None: fp32)
        self.keep = []        # tensors the ops point into (packed weights)

    def act(self, rows, channels):
        self.acts.append((rows, channels))
        return len(self.acts) - 1


def _bn_and_affine(norm, condition, context):
    """(nn.BatchNorm1d, weight, bias) of a plain BatchNorm1d or a PDBatchNorm (whose effective
    affine pair for this forward comes from ``SpUNetBase._prepare_modulation``); None when the
    layer's affine pair is not available in that form."""
    if isinstance(norm, nn.BatchNorm1d):
        return (norm, norm.weight, norm.bias) if norm.affine else None
    bn = norm.bns[norm.conditions.index(condition)] if norm.decouple else norm.bn
    if not norm.adaptive:
        return (bn, bn.weight, bn.bias) if bn.affine else None
    pairs = getattr(context, "pairs", None)
    if pairs is None or norm not in pairs:
        return None
    weight, bias = pairs[norm]
    return bn, weight.float().contiguous(), bias.float().contiguous()


def _plannable_conv(conv, bn, rb):
    return (conv.bias is None and K._use_pr(rb, conv.in_channels, conv.out_channels)
            and conv.out_channels % 32 == 0 and rb.n_out > 1 and rb.n_in > 0 and bn.training
            and bn.momentum is not None and type(bn) is nn.BatchNorm1d)


def _plannable_conv16(conv, bn, rb):
    """What kernels.spconv16_supported asks of a 16-bit conv, and a trainable plain BatchNorm1d behind it."""
    return (conv.bias is None and conv.in_channels % 8 == 0 and conv.out_channels % 8 == 0
            and rb.nbr is not None and rb._transposed_os is not None and rb.n_out > 1 and rb.n_in > 0
            and bn.training and bn.momentum is not None and type(bn) is nn.BatchNorm1d)


def _is_conv(u):
    return u.kind in (UNET_CONV_BN, UNET_CONV_BN16)


def build_plan(model, x, condition=None, context=None, dtype=None):
    """Plan of ``model`` (SpUNet-v1m1 / -v1m3 layout) on the sparse tensor ``x`` whose
    ``indice_dict`` holds the prebuilt geometry; None when a unit is outside what the executor covers.
    ``dtype``: torch.bfloat16 / torch.float16 - the units behind the stem on 16-bit activations."""
    from .spconv import pytorch as spconv

    geo = x.indice_dict
    if not geo or "stem" not in geo:
        return None
    plan = Plan()
    plan.dtype = dtype
    conv_kind = UNET_CONV_BN if dtype is None else UNET_CONV_BN16
    n0 = x.indices.shape[0]
    level_rows = [n0] + [geo[f"spconv{l}"]["rulebook"].n_out for l in range(1, model.num_stages + 1)]
    level_idx = [x.indices] + [geo[f"spconv{l}"]["out_indices"] for l in range(1, model.num_stages + 1)]
    pointwise = {}

    def rulebook_1x1(level):
        if level not in pointwise:
            pointwise[level] = K.build_subm_rulebook(level_idx[level], 1)
        return pointwise[level]

    def add_unit(conv, norm, rb, src, res=None, relu=True, kind=None, c_in=None):
        kind = conv_kind if kind is None else kind
        resolved = _bn_and_affine(norm, condition, context)
        if resolved is None:
            return None
        bn, bn_weight, bn_bias = resolved
        u = _Unit()
        u.kind, u.conv, u.bn, u.rb, u.relu, u.src, u.res = kind, conv, bn, rb, relu, src, res
        u.c_in = conv.in_channels if c_in is None else c_in
        u.c_out, u.K = conv.out_channels, rb.K
        u.n_in, u.n_out = rb.n_in, rb.n_out
        if kind == UNET_CONV_BN and not _plannable_conv(conv, bn, rb):
            return None
        if kind == UNET_CONV_BN16 and not _plannable_conv16(conv, bn, rb):
            return None
        if kind == UNET_STEM and not (bn.training and bn.momentum is not None and type(bn) is nn.BatchNorm1d):
            return None
        u.affine = (bn_weight, bn_bias)
        u.acc_dx = u.acc_res = False
        u.dx_producer = 0
        u.dst = plan.act(u.n_out, u.c_out)
        plan.units.append(u)
        return u.dst

    def block(b, src, rb, level):
        # conv1 -> bn1 -> relu ; conv2 -> bn2 (+ shortcut) -> relu     (BasicBlock.forward :70-83)
        y = add_unit(b.conv1, b.bn1, rb, src)
        if y is None:
            return None
        proj_conv = getattr(b, "proj_conv", None)
        if proj_conv is None and len(b.proj) > 1:
            proj_conv, proj_norm = b.proj[0], b.proj[1]
        elif proj_conv is not None:
            proj_norm = b.proj_norm
        if proj_conv is not None:
            short = add_unit(proj_conv, proj_norm, rulebook_1x1(level), src, relu=False)
            if short is None:
                return None
        else:
            short = src
        return add_unit(b.conv2, b.bn2, rb, y, res=short)

    def conv_norm(m):
        """(conv, norm) of a conv + norm + ReLU stage: SparseSequential (v1m1) or _ConvNormReLU (v1m3)."""
        return (m.conv, m.bn) if hasattr(m, "conv") else (m[0], m[1])

    # stem: 125 offsets, 6 -> 8 zero-padded input channels; output-stationary conv + BatchNorm + ReLU
    conv, norm = conv_norm(model.conv_input)
    stem_rb = geo["stem"]["rulebook"]
    if conv.bias is not None or stem_rb.nbr is None or conv.out_channels % 8:
        return None
    c_pad = conv.in_channels + (-conv.in_channels % 8)
    src = plan.act(n0, c_pad)
    cur = add_unit(conv, norm, stem_rb, src, kind=UNET_STEM, c_in=c_pad)
    if cur is None:
        return None
    skips = [cur]
    for s in range(model.num_stages):
        conv, norm = conv_norm(model.down[s])
        cur = add_unit(conv, norm, geo[f"spconv{s + 1}"]["rulebook"], cur)
        if cur is None:
            return None
        rb = geo[f"subm{s + 1}"]["rulebook"]
        for b in model.enc[s]:
            cur = block(b, cur, rb, s + 1)
            if cur is None:
                return None
        skips.append(cur)
    cur = skips.pop(-1)
    for s in reversed(range(model.num_stages)):
        conv, norm = conv_norm(model.up[s])
        entry = geo[f"spconv{s + 1}"]
        rbt = entry.get("rulebook_t")
        if rbt is None:
            rbt = entry["rulebook_t"] = entry["rulebook"].transposed()
        up = add_unit(conv, norm, rbt, cur)
        if up is None:
            return None
        skip = skips.pop(-1)
        cat = _Unit()
        cat.kind, cat.src, cat.res = UNET_CONCAT, up, skip
        cat.c_in, cat.c_out = plan.acts[up][1], plan.acts[skip][1]
        cat.n_in = cat.n_out = level_rows[s]
        cat.relu, cat.K, cat.acc_dx, cat.acc_res = 0, 0, False, False
        cat.dst = plan.act(level_rows[s], cat.c_in + cat.c_out)
        plan.units.append(cat)
        cur = cat.dst
        rb = geo["subm0" if s == 0 else f"subm{s}"]["rulebook"]
        for b in model.dec[s]:
            cur = block(b, cur, rb, s)
            if cur is None:
                return None
    plan.out_act = cur
    # gradient bookkeeping: records run last to first; the first gradient of an activation is written,
    # later ones are added.  BatchNorm's shortcut gradient (dres) has no accumulating form and is
    # produced before the same unit's grad-input: it must be the first writer of its target.
    seen = {plan.out_act}
    for u in reversed(plan.units):
        if u.kind == UNET_CONCAT:
            u.acc_res = u.res in seen
            if u.src in seen:
                return None
            seen.update((u.res, u.src))
            continue
        if u.res is not None:
            if u.res in seen:
                return None
            seen.add(u.res)
        if _is_conv(u):
            u.acc_dx = u.src in seen
            seen.add(u.src)
    # The LAST gradient an activation receives in backward order comes from its first consumer in forward
    # order.  When that consumer is a conv unit reading it as its input, the BatchNorm backward sums of
    # the unit that PRODUCED the activation ride in that conv's grad-input row reduce (csrc/
    # sparse_conv_pr.hip STATS == 2): the producer then skips the statistics pass of its BatchNorm backward.
    producer = {u.dst: i for i, u in enumerate(plan.units) if u.kind != UNET_CONCAT}
    first_consumer = {}
    for i, u in enumerate(plan.units):
        for a in (u.src, u.res):
            if a is not None and a not in first_consumer:
                first_consumer[a] = i
    for a, i in first_consumer.items():
        u = plan.units[i]
        if u.kind == UNET_CONV_BN and u.src == a and u.res != a and a in producer and a != plan.out_act:
            u.dx_producer = producer[a] + 1
    return plan


def supported(model, x) -> bool:
    return (ENABLED and K.USE_PR == "all" and K.USE_CONVBN and x.features.is_cuda and model.training
            and not getattr(model, "cls_mode", False) and x.features.dtype == torch.float32
            and (precision.sparse_dtype() is None or NATIVE16)
            and x.indices.shape[0] > 1 and torch.is_grad_enabled())


def run(model, x, condition=None, context=None):
    """The backbone's output features for the sparse input ``x``, or None when the executor does not
    cover this model / input (the caller then walks the modules)."""
    if not supported(model, x):
        return None
    plan = build_plan(model, x, condition, context, dtype=precision.sparse_dtype())
    if plan is None:
        return None
    feats = x.features
    pad = plan.acts[0][1] - feats.shape[1]
    stem = plan.units[0]
    w_stem = stem.conv.weight.reshape(stem.c_out, -1, stem.conv.in_channels)
    if pad:
        feats = F.pad(feats, (0, pad))
        w_stem = F.pad(w_stem, (0, pad))
    tensors = []
    for u in plan.units:
        if u.kind == UNET_CONCAT:
            continue
        # (the parameter itself, [c_out, k, k, k, c_in] contiguous = [c_out, K, c_in] in memory: no view
        # op per unit and step in front of the node, no view-backward node behind it)
        w = w_stem if u is stem else u.conv.weight
        u.w_index = len(tensors)
        tensors += [w, u.affine[0], u.affine[1]]
        if u.bn.track_running_stats and u.bn.num_batches_tracked is not None:
            rownorm._bump_batches_tracked(u.bn)
    global CALLS
    CALLS += 1
    # (in the 16-bit mode the result is 16-bit, as on the modular walk: its consumers widen it)
    return SpUNetFunction.apply(feats.contiguous(), plan, *tensors)


def _floats(plan, rows, ch, half):
    """Arena floats of a [rows, ch] matrix: fp32, or 16-bit elements when ``half`` in the 16-bit mode."""
    n = rows * ch
    return (n + 1) // 2 if (half and plan.dtype is not None) else n


def _fill_forward(plan, feats, tensors):
    dev = feats.device
    arena = plan.fwd
    half = plan.dtype is not None
    code = K.DTYPE_CODE[plan.dtype] if half else 0
    plan.act_off = [None] * len(plan.acts)
    for u in plan.units:
        if u.kind != UNET_CONCAT:
            # (the stem's conv output stays fp32; everything else is in the plan's element type)
            u.y_off = arena.reserve(_floats(plan, u.n_out, u.c_out, u.kind != UNET_STEM))
            u.mi_off = arena.reserve(2 * u.c_out)
        plan.act_off[u.dst] = arena.reserve(_floats(plan, plan.acts[u.dst][0], plan.acts[u.dst][1], True))
    arena.allocate(dev)
    ops = (UnetOp * len(plan.units))()
    act_ptr = [None if off is None else arena.ptr(off) for off in plan.act_off]
    act_ptr[0] = feats.data_ptr()
    max_prod, max_c = 1, 8
    for op, u in zip(ops, plan.units):
        op.kind, op.c_in, op.c_out, op.relu = u.kind, u.c_in, u.c_out, int(bool(u.relu))
        op.n_in, op.n_out = u.n_in, u.n_out
        op.dtype = code
        op.x = act_ptr[u.src]
        op.residual = act_ptr[u.res] if u.res is not None else None
        op.out = act_ptr[u.dst]
        if u.kind == UNET_CONCAT:
            op.dx_accumulate = int(u.acc_res)
            continue
        w, bw, bb = tensors[u.w_index:u.w_index + 3]
        assert w.is_contiguous() and bw.is_contiguous() and bb.is_contiguous()
        op.K = u.rb.K
        u.geom = u.rb.geom(u.c_in, u.c_out, positions=u.kind == UNET_CONV_BN)
        u.geom_ptr = ctypes.pointer(u.geom)
        op.geom = u.geom_ptr
        op.weight, op.bn_weight, op.bn_bias = w.data_ptr(), bw.data_ptr(), bb.data_ptr()
        bn = u.bn
        if bn.track_running_stats:
            op.running_mean, op.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
        op.eps, op.momentum = float(bn.eps), float(bn.momentum)
        op.y_conv, op.mean_invstd = arena.ptr(u.y_off), arena.ptr(u.mi_off)
        max_c = max(max_c, u.c_out, u.c_in)
        if u.kind == UNET_STEM:
            op.nbr, op.nbr_stride, op.kflip = u.rb.nbr.data_ptr(), u.rb.nbr_stride, u.rb.kflip
            n_tiles_w = u.geom.n_tiles_w
            max_prod = max(max_prod, int(_lib.lib().pv2_spconv_wgrad_partial_floats(
                u.c_in, u.c_out, n_tiles_w)))
        elif u.kind == UNET_CONV_BN16:
            # packed 16-bit copies of the fp32 master weight, once per optimiser step (cached on the conv
            # module, shared with the modular walk: kernels.packed_weights)
            cache = u.conv.__dict__.setdefault("_pv2_packed", {})
            fwd, bwd = K.packed_weights(w.detach().reshape(u.c_out, u.rb.K, u.c_in), plan.dtype, cache)
            plan.keep += [fwd, bwd]
            op.packed_fwd, op.packed_bwd = fwd.data_ptr(), bwd.data_ptr()
            rb = u.rb
            op.nbr, op.nbr_stride, op.kflip = rb.nbr.data_ptr(), rb.nbr_stride, rb.kflip
            op.perm = rb.perm.data_ptr() if rb.perm is not None else None
            nbr_t, stride_t, perm_t, kflip_t = rb._transposed_os
            op.nbr_t, op.nbr_t_stride, op.kflip_t = nbr_t.data_ptr(), stride_t, kflip_t
            op.perm_t = perm_t.data_ptr() if perm_t is not None else None
            ts16, n16, _ = rb.tiles(_lib.WGRAD_TILE)
            op.tile_start16, op.n_tiles16 = ts16.data_ptr(), n16
            op.dx_accumulate = int(u.acc_dx)
        else:
            op.dx_accumulate = int(u.acc_dx)
            op.dx_producer = u.dx_producer
            max_prod = max(max_prod, u.rb.n_pairs * max(u.c_in, u.c_out))
    plan.ops = ops
    plan.max_prod, plan.max_c = max_prod, max_c
    return ops


def _gradient_slabs(plan, tensors, hook, parena):
    """Slabs of the parameter-gradient arena in the order the backward completes them (the executor walks
    the units last to first, the arena is laid out first to last): ([(parameter, offset, numel)],
    [(lo, hi)], [unit index whose completion finishes the slab]) - or None when the reducer does not want
    this plan.  The stem (unit 0; its weight travels zero-padded, so its gradient is not a parameter's
    ``.grad``) stays outside the slabs.  Members are the LEAF tensors only: a BatchNorm affine pair that is
    computed (SpUNet-v1m3's prompt-driven normalisation, spconv_unet_v1m3_pdnorm.py:23-72: the modulated
    pair is a slice of a product) has its gradient in a second arena that no slab covers - autograd still
    reads it, and the parameters behind it are reduced with the rest after the backward."""
    convs = [(i, u) for i, u in enumerate(plan.units) if _is_conv(u)]
    if len(convs) < 2:
        return None
    members = []
    for _, u in convs:
        w, bw, bb = tensors[u.w_index:u.w_index + 3]
        if not w.is_leaf:
            return None
        if u.affine_in_arena:
            members += [(bb, u.gsum_off, u.c_out), (bw, u.gsum_off + u.c_out, u.c_out)]
        members.append((w, u.dw_off, w.numel()))
    if not hook.wants([t for t, _, _ in members]):
        return None
    end = parena.size
    spans, units = [], []
    hi = end
    for pos in range(len(convs) - 1, -1, -1):
        i, u = convs[pos]
        lo = u.gsum_off if u.affine_in_arena else u.dw_off
        if hi - lo >= hook.slab_elems or pos == 0:
            spans.append((lo, hi))
            units.append(i)
            hi = lo
    return members, spans, units


def _slab_events(dev, n):
    """n reusable (training stream, side stream) event pairs of this device, created once (a torch event
    has no handle before its first record)."""
    pool = _EVENT_POOL.setdefault(dev.index, [])
    while len(pool) < n:
        pair = (torch.cuda.Event(), torch.cuda.Event())
        for e in pair:
            e.record(torch.cuda.current_stream(dev))
        pool.append(pair)
    return pool[:n]


class SpUNetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, plan, *tensors):
        dev = feats.device
        ops = _fill_forward(plan, feats, tensors)
        prod = K.workspace("prod", dev, plan.max_prod)
        stats = rownorm._workspace(dev, plan.max_c)
        _lib.check(_lib.lib().pv2_unet_forward(ops, len(ops), K._ptr(prod), K._ptr(stats),
                                               K._stream(feats)), "pv2_unet_forward")
        ctx.plan = plan
        ctx.feats = feats
        ctx.save_for_backward(*tensors)
        rows, ch = plan.acts[plan.out_act]
        if plan.dtype is None:
            return plan.fwd.view(plan.act_off[plan.out_act], rows, ch)
        off = plan.act_off[plan.out_act]
        return plan.fwd.tensor[off:off + (rows * ch + 1) // 2].view(plan.dtype)[:rows * ch].view(rows, ch)

    @staticmethod
    def backward(ctx, grad_out):
        plan = ctx.plan
        tensors = ctx.saved_tensors
        dev = grad_out.device
        grad_out = grad_out.contiguous()
        arena = _Arena()
        g_off = [None] * len(plan.acts)
        want_dx = ctx.needs_input_grad[0]   # (a learnable mask token was written into the input features)
        half = plan.dtype is not None
        if half:
            assert grad_out.dtype == plan.dtype, (grad_out.dtype, plan.dtype)
        for a, (rows, ch) in enumerate(plan.acts):
            if (a != 0 or want_dx) and a != plan.out_act:
                g_off[a] = arena.reserve(_floats(plan, rows, ch, a != 0))
        # parameter gradients live in their OWN (small) arena: ``AccumulateGrad`` keeps the returned
        # views as ``param.grad`` until the next ``zero_grad`` - carved from the activation-gradient
        # arena they would pin its ~GB through the following forward
        parena = _Arena()
        narena = _Arena()   # gradients of COMPUTED affine pairs (never reduced in place, see _gradient_slabs)
        tmp_floats = 0
        for u in plan.units:
            if u.kind == UNET_CONCAT:
                continue
            u.dy_off = arena.reserve(_floats(plan, u.n_out, u.c_out, u.kind != UNET_STEM))
            bw, bb = tensors[u.w_index + 1], tensors[u.w_index + 2]
            u.affine_in_arena = bw.is_leaf and bb.is_leaf
            u.gsum_off = (parena if u.affine_in_arena else narena).reserve(2 * u.c_out)
            u.dw_off = parena.reserve(u.c_out * u.rb.K * u.c_in)
            if u.kind == UNET_CONV_BN16 and u.acc_dx:
                tmp_floats = max(tmp_floats, _floats(plan, u.n_in, u.c_in, True))
        tmp_off = arena.reserve(tmp_floats) if tmp_floats else None
        arena.allocate(dev)
        parena.allocate(dev)
        narena.allocate(dev)
        g_ptr = [None if off is None else arena.ptr(off) for off in g_off]
        g_ptr[plan.out_act] = grad_out.data_ptr()
        ops = plan.ops
        part_floats = 1
        for op, u in zip(ops, plan.units):
            op.grad_out = g_ptr[u.dst]
            op.dx = g_ptr[u.src]
            op.dres = g_ptr[u.res] if u.res is not None else None
            if u.kind == UNET_CONCAT:
                continue
            op.dy, op.dweight = arena.ptr(u.dy_off), parena.ptr(u.dw_off)
            op.gsum = (parena if u.affine_in_arena else narena).ptr(u.gsum_off)
            if u.kind == UNET_CONV_BN:
                part_floats = max(part_floats, int(_lib.lib().pv2_spconv_wgrad_partial_floats(
                    u.c_in, u.c_out, u.geom.n_tiles_w)))
            elif u.kind == UNET_CONV_BN16:
                op.dx_tmp = arena.ptr(tmp_off) if (u.acc_dx and tmp_off is not None) else None
            elif want_dx:   # the stem's grad-input pass reads the weight as [c_in, K, c_out]
                w_t = tensors[u.w_index].detach().permute(2, 1, 0).contiguous()
                op.weight_t = w_t.data_ptr()
        # weight gradients on the backward side stream when every one of them is only stored
        weights = [tensors[u.w_index] for u in plan.units if _is_conv(u)]
        side = None
        if sidestream.active(grad_out) and all(sidestream.safe_leaf(w) for w in weights):
            # (the plan too: its rulebooks own the pair lists / tile prefixes the side stream's kernels
            # read - released with ``ctx.plan = None`` below while those kernels were still queued,
            # the arrays could be handed to the next allocation on this stream BEFORE the join:
            # a memory fault once in a few runs of the full-size fixtures)
            side = sidestream.native_fork(dev, (plan.fwd.tensor, arena.tensor, parena.tensor, ctx.feats,
                                                grad_out, plan))
            part = K.workspace("wgrad", dev, part_floats, stream=side)
        else:
            part = K.workspace("wgrad", dev, part_floats)
        prod = K.workspace("prod", dev, plan.max_prod)
        stats = rownorm._workspace(dev, plan.max_c)
        hook, slabs = GRAD_SLAB_HOOK, None
        if hook is not None:
            slabs = _gradient_slabs(plan, tensors, hook, parena)
        if slabs is None:
            _lib.check(_lib.lib().pv2_unet_backward(
                ops, len(ops), K._ptr(prod), K._ptr(stats), K._ptr(part), K._stream(grad_out),
                ctypes.c_void_p(side.cuda_stream) if side is not None else None), "pv2_unet_backward")
        else:
            members, spans, units = slabs
            events = _slab_events(dev, len(spans))
            unit_arr = (ctypes.c_int32 * len(units))(*units)
            ev_main = (ctypes.c_void_p * len(units))(*[e[0].cuda_event for e in events])
            ev_side = (ctypes.c_void_p * len(units))(*[e[1].cuda_event for e in events])
            _lib.check(_lib.lib().pv2_unet_backward_ev(
                ops, len(ops), K._ptr(prod), K._ptr(stats), K._ptr(part), K._stream(grad_out),
                ctypes.c_void_p(side.cuda_stream) if side is not None else None, len(units), unit_arr,
                ev_main, ev_side), "pv2_unet_backward_ev")
            hook._on_arena(parena.tensor, members, spans, events)
        grads = [None] * len(tensors)
        convs = [u for u in plan.units if u.kind != UNET_CONCAT]
        for ar, pick in ((parena, True), (narena, False)):
            specs, owners = [], []
            for u in convs:   # (reserved in this order per unit: gsum, dweight)
                if u.affine_in_arena == pick:
                    specs += [(u.gsum_off, u.c_out), (u.gsum_off + u.c_out, u.c_out)]
                    owners += [(u, 2), (u, 1)]        # d bn bias = sum g, d bn weight = sum g * xhat
                if pick:
                    specs.append((u.dw_off, tensors[u.w_index].numel()))
                    owners.append((u, 0))
            if not specs:
                continue
            for (u, slot), piece in zip(owners, ar.views(specs)):
                i = u.w_index
                grads[i + slot] = piece.view(tensors[i].shape) if slot == 0 else piece
        ctx.plan = None
        g_feats = arena.view(g_off[0], *plan.acts[0]) if want_dx else None
        return (g_feats, None) + tuple(grads)
