"""DefaultSegmentorV2 on the MI355X PTv3 backbone (reference: pointcept/models/default.py:40-95)."""
import torch
import torch.nn as nn

from pointcept.models.losses import build_criteria
from pointcept.models.utils.structure import Point
from pointcept.models.utils.hip_layers import Linear
from .builder import MODELS, build_model


@MODELS.register_module()
class DefaultSegmentorV2(nn.Module):
    def __init__(self, num_classes, backbone_out_channels, backbone=None, criteria=None, freeze_backbone=False):
        super().__init__()
        self.seg_head = Linear(backbone_out_channels, num_classes) if num_classes > 0 else nn.Identity()
        self.backbone = build_model(backbone)
        self.criteria = build_criteria(criteria)
        self.freeze_backbone = freeze_backbone
        if self.freeze_backbone:
            for p in self.backbone.parameters():
                p.requires_grad = False

    def forward(self, input_dict, return_point=False):
        point = Point(input_dict)
        point = self.backbone(point)
        if isinstance(point, Point):
            while "pooling_parent" in point.keys():  # enc_mode backbones: concatenate back up (default.py:70-75)
                assert "pooling_inverse" in point.keys()
                parent = point.pop("pooling_parent")
                inverse = point.pop("pooling_inverse")
                parent.feat = torch.cat([parent.feat, point.feat[inverse]], dim=-1)
                point = parent
            feat = point.feat
        else:
            feat = point
        seg_logits = self.seg_head(feat.contiguous()).float()
        return_dict = dict()
        if return_point:
            return_dict["point"] = point
        if self.training:
            return_dict["loss"] = self.criteria(seg_logits, input_dict["segment"])
        elif "segment" in input_dict.keys():
            return_dict["loss"] = self.criteria(seg_logits, input_dict["segment"])
            return_dict["seg_logits"] = seg_logits
        else:
            return_dict["seg_logits"] = seg_logits
        return return_dict
