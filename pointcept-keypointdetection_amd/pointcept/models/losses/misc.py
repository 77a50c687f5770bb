"""CrossEntropyLoss entry of the LOSSES registry (reference: pointcept/models/losses/misc.py)."""
import torch
import torch.nn as nn

from .builder import LOSSES


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    def __init__(self, weight=None, size_average=None, reduce=None, reduction="mean", label_smoothing=0.0,
                 loss_weight=1.0, ignore_index=-1):
        super().__init__()
        weight = torch.tensor(weight).cuda() if weight is not None else None
        self.loss_weight = loss_weight
        self.loss = nn.CrossEntropyLoss(weight=weight, size_average=size_average, ignore_index=ignore_index,
                                        reduce=reduce, reduction=reduction, label_smoothing=label_smoothing)

    def forward(self, pred, target):
        return self.loss(pred, target) * self.loss_weight
