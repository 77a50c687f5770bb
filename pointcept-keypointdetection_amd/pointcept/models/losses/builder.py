"""Loss registry + the `Criteria` aggregate the segmentor wrappers call (interface of the reference's
pointcept/models/losses/builder.py).  Losses are plain torch and sit outside the MI355X hot path
(SURVEY.md section 2a row 9); the registry exists so DefaultSegmentorV2 configs build unchanged."""
from functools import reduce

from pointcept.utils.registry import Registry

LOSSES = Registry("losses")


class Criteria:
    """Sum of the configured losses; with no loss configured the prediction passes through (the model computes
    its own loss, as OffsetKeypointPTv3 does)."""

    def __init__(self, cfg=None):
        self.cfg = list(cfg) if cfg else []
        self.criteria = [LOSSES.build(cfg=c) for c in self.cfg]

    def __call__(self, pred, target):
        if not self.criteria:
            return pred
        return reduce(lambda acc, fn: acc + fn(pred, target), self.criteria, 0)


def build_criteria(cfg):
    return Criteria(cfg)
