"""OPTIMIZERS / SCHEDULERS registries (ponder/utils/optimizer.py:13-56, scheduler.py:12-148):
the entries the pre-training configs name - SGD, Adam, AdamW; OneCycleLR, MultiStepLR, CosineAnnealingLR."""
import torch

from .registry import Registry

OPTIMIZERS = Registry("optimizers")
SCHEDULERS = Registry("schedulers")
for _o in (torch.optim.SGD, torch.optim.Adam, torch.optim.AdamW):
    OPTIMIZERS.register_module(module=_o, name=_o.__name__)


def build_optimizer(cfg, model, param_dicts=None):
    """``param_dicts=[dict(keyword=..., lr_scale=...)]`` gives matching parameters a scaled lr."""
    if not param_dicts:
        params = model.parameters()
    else:
        groups = [dict(params=[])] + [dict(params=[], lr=cfg["lr"] * d["lr_scale"]) for d in param_dicts]
        for name, p in model.named_parameters():
            for i, d in enumerate(param_dicts):
                if d["keyword"] in name:
                    groups[i + 1]["params"].append(p)
                    break
            else:
                groups[0]["params"].append(p)
        params = groups
    cfg = dict(cfg)
    cfg["params"] = params
    return OPTIMIZERS.build(cfg=cfg)


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
