"""Criteria builder (reference: pointcept/models/losses/builder.py). Losses are plain torch, out of the
MI355X hot path; kept so DefaultSegmentorV2 configs build unchanged."""
from pointcept.utils.registry import Registry

LOSSES = Registry("losses")


class Criteria(object):
    def __init__(self, cfg=None):
        self.cfg = cfg if cfg is not None else []
        self.criteria = [LOSSES.build(cfg=loss_cfg) for loss_cfg in self.cfg]

    def __call__(self, pred, target):
        if len(self.criteria) == 0:
            return pred  # loss computed inside the model
        loss = 0
        for c in self.criteria:
            loss += c(pred, target)
        return loss


def build_criteria(cfg):
    return Criteria(cfg)
