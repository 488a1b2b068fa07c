"""Loss registry with the single criterion the pre-training path touches (PPT's CrossEntropyLoss,
ponder/models/losses/misc.py:15-40; builder ponder/models/losses/builder.py:13-31)."""
import torch
import torch.nn as nn

from ...utils.registry import Registry

LOSSES = Registry("losses")


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    def __init__(self, weight=None, size_average=None, reduce=None, reduction="mean",
                 label_smoothing=0.0, loss_weight=1.0, ignore_index=-1):
        super().__init__()
        weight = torch.tensor(weight) if weight is not None else None
        self.loss_weight = loss_weight
        self.loss = nn.CrossEntropyLoss(weight=weight, size_average=size_average,
                                        ignore_index=ignore_index, reduce=reduce,
                                        reduction=reduction, label_smoothing=label_smoothing)

    def forward(self, pred, target):
        return self.loss(pred, target) * self.loss_weight


class Criteria:
    def __init__(self, cfg=None):
        self.criteria = [LOSSES.build(cfg=c) for c in (cfg or [])]

    def __call__(self, pred, target):
        if not self.criteria:
            return pred  # loss computed inside the model
        return sum(c(pred, target) for c in self.criteria)


def build_criteria(cfg):
    return Criteria(cfg)
