"""Segmentation wrapper around the MI355X PTv3 backbones: registry name "DefaultSegmentorV2".

Contract of the reference class (pointcept/models/default.py:41-95), which the semseg configs build
(`configs/scannet/semseg-pt-v3m1-0-base.py`, `configs/pigseg/semseg-ptv3-v1m1-0-base.py`): constructor keywords
`num_classes, backbone_out_channels, backbone, criteria, freeze_backbone`; state_dict = `seg_head.{weight,bias}` +
`backbone.*`; `forward(input_dict, return_point=False)` returns `loss` in training, `loss` + `seg_logits` when the
batch carries "segment" in eval, `seg_logits` alone otherwise (+ `point` on request).  The head is one ptv3_gemm.
"""
import torch
import torch.nn as nn

from pointcept.models.losses import build_criteria
from pointcept.models.utils.structure import Point
from pointcept.models.utils.hip_layers import Linear
from .builder import MODELS, build_model


def _full_resolution(point):
    """Point-order features at the input resolution.  A decoder-less backbone (enc_mode) hands back its deepest
    level: every level's features are broadcast to its parent through `pooling_inverse` and appended to the
    parent's channels until the unpooled root is reached (default.py:70-75)."""
    if not isinstance(point, Point):
        return point, point                      # legacy backbones return the feature matrix itself
    levels = [point]
    while "pooling_parent" in levels[-1].keys():
        child = levels[-1]
        if "pooling_inverse" not in child.keys():
            raise AssertionError("pooling_parent without pooling_inverse")
        levels.append(child["pooling_parent"])
    for child, parent in zip(levels[:-1], levels[1:]):
        inverse = child.pop("pooling_inverse")
        child.pop("pooling_parent")
        parent.feat = torch.cat((parent.feat, child.feat.index_select(0, inverse)), dim=1)
    root = levels[-1]
    return root, root.feat


@MODELS.register_module()
class DefaultSegmentor(nn.Module):
    """default.py:14-37: a backbone that returns per-point logits itself (Swin3D-v1m1 is built this way,
    configs/s3dis/semseg-swin3d-v1m1-0-small.py:9-31)."""

    def __init__(self, backbone=None, criteria=None):
        super().__init__()
        self.backbone = build_model(backbone)
        self.criteria = build_criteria(criteria)

    def forward(self, input_dict):
        if "condition" in input_dict.keys():
            input_dict["condition"] = input_dict["condition"][0]
        seg_logits = self.backbone(input_dict)
        if self.training:
            return dict(loss=self.criteria(seg_logits, input_dict["segment"]))
        if "segment" in input_dict.keys():
            return dict(loss=self.criteria(seg_logits, input_dict["segment"]), seg_logits=seg_logits)
        return dict(seg_logits=seg_logits)


@MODELS.register_module()
class DefaultSegmentorV2(nn.Module):
    def __init__(self, num_classes, backbone_out_channels, backbone=None, criteria=None, freeze_backbone=False):
        super().__init__()
        self.seg_head = Linear(backbone_out_channels, num_classes) if num_classes > 0 else nn.Identity()
        self.backbone = build_model(backbone)
        self.criteria = build_criteria(criteria)
        self.freeze_backbone = bool(freeze_backbone)
        if self.freeze_backbone:
            self.backbone.requires_grad_(False)

    def forward(self, input_dict, return_point=False):
        # the collated dict goes to the backbone as it is (the backbone wraps it in its own Point, as the reference's
        # does with the Point it is handed): the native executor then derives `batch` from `offset` on its geometry
        # stream instead of receiving one produced on the caller's stream a moment ago
        point, feat = _full_resolution(self.backbone(input_dict))
        seg_logits = self.seg_head(feat.contiguous()).float()
        result = {"point": point} if return_point else {}
        labelled = "segment" in input_dict.keys()
        if self.training or labelled:
            result["loss"] = self.criteria(seg_logits, input_dict["segment"])
        if not self.training:
            result["seg_logits"] = seg_logits
        return result
