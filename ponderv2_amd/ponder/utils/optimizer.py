"""OPTIMIZERS / SCHEDULERS registries (ponder/utils/optimizer.py:13-56, scheduler.py:12-148):
the entries the pre-training configs name - SGD, Adam, AdamW; OneCycleLR, MultiStepLR, CosineAnnealingLR."""
import logging

import torch

from .registry import Registry

OPTIMIZERS = Registry("optimizers")
SCHEDULERS = Registry("schedulers")
for _o in (torch.optim.SGD, torch.optim.Adam, torch.optim.AdamW):
    OPTIMIZERS.register_module(module=_o, name=_o.__name__)


def _prefer_fused(cfg, params):
    """``fused=True`` for the torch optimizers that have a single-kernel multi-tensor step, when
    every parameter lives on the GPU and the config does not choose an implementation itself.
    Same update rule (momentum / nesterov / weight decay as configured); one pass over parameter,
    gradient and momentum buffer instead of the ~14 passes and ~15 launches of the for-each form
    (SGD with nesterov over SpUNet + UNet3D's 39 M parameters: 0.9 -> ~0.3 ms per step on MI355X).
    PV2_FUSED_OPTIMIZER=0 keeps torch's default."""
    import inspect
    import os

    if os.environ.get("PV2_FUSED_OPTIMIZER", "1") == "0" or "fused" in cfg or "foreach" in cfg:
        return cfg
    cls = OPTIMIZERS.get(cfg.get("type")) if isinstance(cfg.get("type"), str) else cfg.get("type")
    if cls is None or "fused" not in inspect.signature(cls.__init__).parameters:
        return cfg
    params = list(params)
    if params and all(p.is_cuda and torch.is_floating_point(p) for p in params):
        cfg = dict(cfg, fused=True)
    return cfg


def _ready_for_fused(optimizer):
    """torch's fused SGD step takes the momentum buffers of a group as ONE list and fails when some
    exist and some do not - which happens whenever parameters receive their first gradient at
    different steps (the multi-dataset model trains one condition's normalisation layers per
    step).  With dampening 0 a zero buffer gives the same first update as the lazily created one
    (buf = momentum * 0 + grad), so the buffers are created up front; a group with dampening falls
    back to the for-each step."""
    if not (isinstance(optimizer, torch.optim.SGD) and optimizer.defaults.get("fused")):
        return optimizer
    for group in optimizer.param_groups:
        if not group.get("fused"):
            continue
        if group.get("dampening", 0) != 0:
            group["fused"], group["foreach"] = False, True
            continue
        if group.get("momentum", 0) != 0:
            for p in group["params"]:
                if optimizer.state[p].get("momentum_buffer") is None:
                    optimizer.state[p]["momentum_buffer"] = torch.zeros_like(p)
    if not getattr(optimizer, "_pv2_reready_hook", False):
        # load_state_dict REPLACES the state: a checkpoint written by the for-each step (or by the
        # reference), or one in which some parameters were never stepped (the multi-dataset model's
        # per-condition norms), brings back the mixed None / tensor buffer list the fused step
        # fails on - so the buffers are completed again after every load
        optimizer.register_load_state_dict_post_hook(lambda opt: _ready_for_fused(opt) and None)
        optimizer._pv2_reready_hook = True
    return optimizer


def _lean_fused_sgd_step(optimizer):
    """Replace the per-step Python of torch's fused SGD (``_init_group`` walks every parameter,
    re-checks its state and regroups the lists by device and dtype on every call: ~1.3 ms of host
    time per step for this model's ~240 tensors, on a step that is host-bound) by a direct call of
    the same fused multi-tensor kernel (``torch._fused_sgd_``) on lists that are built once.  Same
    update rule, same state (the momentum buffers are the ones in ``optimizer.state``, so
    checkpoints are unchanged); hyper-parameters are read from the group every step (the OneCycle
    schedule moves ``lr`` and ``momentum``).  Parameters without a gradient are skipped, as torch
    does.  Falls back to torch's own step when called with a closure."""
    if not (isinstance(optimizer, torch.optim.SGD) and optimizer.defaults.get("fused")
            and hasattr(torch, "_fused_sgd_")):
        return optimizer
    torch_step = optimizer.step
    cache = {}

    def lists(group_index, group):
        key = (group_index, len(group["params"]))
        if key not in cache:
            ps = list(group["params"])
            cache[key] = (ps, [optimizer.state[p].get("momentum_buffer") for p in ps])
        return cache[key]

    @torch.no_grad()
    def step(self, closure=None):   # (bound below: lr schedulers wrap ``optimizer.step.__func__``)
        if closure is not None:
            return torch_step(closure)
        grad_scale = getattr(optimizer, "grad_scale", None)
        found_inf = getattr(optimizer, "found_inf", None)
        for gi, group in enumerate(optimizer.param_groups):
            if not group.get("fused") or group.get("dampening", 0) != 0:
                return torch_step()
            ps, bufs = lists(gi, group)
            if group["momentum"] != 0 and any(b is None for b in bufs):
                cache.clear()           # (state replaced by load_state_dict: rebuild next call)
                return torch_step()
            sel = [i for i, p in enumerate(ps) if p.grad is not None]
            if not sel:
                continue
            full = len(sel) == len(ps)
            params = ps if full else [ps[i] for i in sel]
            optimizer._pv2_lean_steps += 1
            torch._fused_sgd_(params, [p.grad for p in params],
                              [] if group["momentum"] == 0 else (bufs if full else [bufs[i] for i in sel]),
                              weight_decay=group["weight_decay"], momentum=group["momentum"],
                              lr=group["lr"], dampening=0.0, nesterov=group["nesterov"],
                              maximize=group.get("maximize", False), is_first_step=False,
                              grad_scale=grad_scale, found_inf=found_inf)
        return None

    import types

    optimizer.register_load_state_dict_post_hook(lambda opt: cache.clear())
    optimizer.step = types.MethodType(step, optimizer)
    optimizer._pv2_lean_steps = 0   # group updates that took the short route (tests read it)
    return optimizer


def _lean_fused_adam_step(optimizer):
    """The same for torch's fused Adam / AdamW (the nuScenes configuration): after torch's own first step
    has created the state, every later step calls ``torch._fused_adam(w)_`` on cached lists - parameters,
    moments and the device-side step counters of the parameters that HAVE state - instead of re-walking
    ~240 parameters, their state dicts and a device / dtype regrouping in Python (~1.5 ms per step on a
    host-bound step).  Same kernel, same state, same checkpoints; lr and betas are read from the group
    every step (OneCycle moves both).  Any deviation - a closure, amsgrad, a tensor lr, a parameter whose
    gradient appears or disappears, a reloaded state - goes back to torch's step (and re-arms)."""
    if not (isinstance(optimizer, (torch.optim.Adam, torch.optim.AdamW)) and optimizer.defaults.get("fused")
            and hasattr(torch, "_fused_adamw_") and hasattr(torch, "_fused_adam_")):
        return optimizer
    torch_step = optimizer.step
    cache = {}

    def lists(gi, group):
        if gi not in cache:
            ps = [p for p in group["params"] if p.grad is not None]
            st = [optimizer.state.get(p) for p in ps]
            if not ps or any((not s_) or "exp_avg" not in s_ or not torch.is_tensor(s_.get("step"))
                             or not s_["step"].is_cuda for s_ in st):
                return None
            cache[gi] = (ps, [s_["exp_avg"] for s_ in st], [s_["exp_avg_sq"] for s_ in st],
                         [s_["step"] for s_ in st], len(group["params"]),
                         [p for p in group["params"] if p.grad is None])
        return cache[gi]

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None or getattr(optimizer, "grad_scale", None) is not None \
                or getattr(optimizer, "found_inf", None) is not None:
            cache.clear()
            return torch_step(closure)
        plans = []
        for gi, group in enumerate(optimizer.param_groups):
            plan = lists(gi, group) if (group.get("fused") and not group.get("amsgrad")
                                        and not group.get("differentiable")
                                        and not torch.is_tensor(group["lr"])) else None
            if plan is None or plan[4] != len(group["params"]) \
                    or any(p.grad is None for p in plan[0]) or any(p.grad is not None for p in plan[5]):
                cache.clear()
                return torch_step()
            plans.append(plan)
        for group, (ps, m1, m2, steps, _, _) in zip(optimizer.param_groups, plans):
            beta1, beta2 = group["betas"]
            decoupled = isinstance(optimizer, torch.optim.AdamW) or group.get("decoupled_weight_decay", False)
            optimizer._pv2_lean_steps += 1
            torch._foreach_add_(steps, 1)
            (torch._fused_adamw_ if decoupled else torch._fused_adam_)(
                ps, [p.grad for p in ps], m1, m2, [], steps, amsgrad=False, lr=group["lr"],
                beta1=float(beta1), beta2=float(beta2), weight_decay=group["weight_decay"],
                eps=group["eps"], maximize=group.get("maximize", False), grad_scale=None, found_inf=None)
        return None

    import types

    optimizer.register_load_state_dict_post_hook(lambda opt: cache.clear())
    optimizer.step = types.MethodType(step, optimizer)
    optimizer._pv2_lean_steps = 0
    return optimizer


def _lean(optimizer):
    return _lean_fused_adam_step(_lean_fused_sgd_step(_ready_for_fused(optimizer)))


def build_optimizer(cfg, model, param_dicts=None):
    """``param_dicts=[dict(keyword=..., lr=..., momentum=..., weight_decay=...)]`` puts the
    parameters whose name contains ``keyword`` into their own group with those ABSOLUTE settings
    (ponder/utils/optimizer.py:21-56; e.g. ``dict(keyword="modulation", lr=0.005)`` in the reference's
    multi-dataset configs); group 0 keeps ``cfg.lr``.  ``lr_scale`` (a multiple of ``cfg.lr``) is
    accepted as an extra spelling."""
    cfg = dict(cfg)
    cfg = _prefer_fused(cfg, model.parameters())
    if param_dicts is None:
        cfg["params"] = model.parameters()
        return _lean(OPTIMIZERS.build(cfg=cfg))
    groups, names = [dict(params=[], lr=cfg["lr"])], [[]]
    for d in param_dicts:
        g = dict(params=[])
        for key in ("lr", "momentum", "weight_decay"):
            if key in d:
                g[key] = d[key]
        if "lr_scale" in d and "lr" not in d:
            g["lr"] = cfg["lr"] * d["lr_scale"]
        groups.append(g)
        names.append([])
    for name, p in model.named_parameters():
        for i, d in enumerate(param_dicts):
            if d["keyword"] in name:
                groups[i + 1]["params"].append(p)
                names[i + 1].append(name)
                break
        else:
            groups[0]["params"].append(p)
            names[0].append(name)
    log = logging.getLogger("ponder")
    for i, g in enumerate(groups):
        settings = "".join(f" {k}: {v};" for k, v in g.items() if k != "params")
        log.info(f"Params Group {i + 1} -{settings} Params: {names[i]}.")
    cfg["params"] = groups
    return _lean(OPTIMIZERS.build(cfg=cfg))


@SCHEDULERS.register_module()
class OneCycleLR(torch.optim.lr_scheduler.OneCycleLR):
    def __init__(self, optimizer, total_steps, **kwargs):
        super().__init__(optimizer=optimizer, total_steps=total_steps, **kwargs)


@SCHEDULERS.register_module()
class MultiStepLR(torch.optim.lr_scheduler.MultiStepLR):
    def __init__(self, optimizer, milestones, total_steps, gamma=0.1, **kwargs):
        super().__init__(optimizer, milestones=[int(m * total_steps) for m in milestones],
                         gamma=gamma, **kwargs)


@SCHEDULERS.register_module()
class CosineAnnealingLR(torch.optim.lr_scheduler.CosineAnnealingLR):
    def __init__(self, optimizer, total_steps, eta_min=0, **kwargs):
        super().__init__(optimizer, T_max=total_steps, eta_min=eta_min, **kwargs)


def build_scheduler(cfg, optimizer):
    cfg = dict(cfg)
    cfg["optimizer"] = optimizer
    return SCHEDULERS.build(cfg=cfg)
