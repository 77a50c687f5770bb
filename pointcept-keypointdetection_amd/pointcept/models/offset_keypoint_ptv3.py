"""OffsetKeypointPTv3: dense per-point keypoint-offset head on the MI355X PTv3 backbone.

Counterpart of the reference's pointcept/models/offset_keypoint_ptv3.py:6-107: same constructor
(backbone_conf, num_keypoints=6, hidden_dim=256), same `head.{0,1,3}` parameters, same output dict.
Head = Linear -> BatchNorm1d -> ReLU -> Linear, run as two HIP GEMMs (BN + ReLU in the first epilogue).
Every scalar entry of the result is a 0-d tensor (the reference stores python floats for two of them,
which its own InformationWriter hook cannot `.item()`).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ptv3_hip import ops
from pointcept.models.builder import MODELS, build_model
from pointcept.models.utils.hip_layers import Linear, BatchNorm1d, ReLU, _no_training, check_sync_batchnorm


@MODELS.register_module()
class OffsetKeypointPTv3(nn.Module):
    def __init__(self, backbone_conf, num_keypoints=6, hidden_dim=256):
        super().__init__()
        self.backbone = build_model(backbone_conf)
        in_channels = backbone_conf["dec_channels"][0]
        self.num_keypoints = num_keypoints
        output_dim = num_keypoints * 4  # (dx, dy, dz, mask logit) per keypoint
        self.head = nn.Sequential(
            Linear(in_channels, hidden_dim),
            BatchNorm1d(hidden_dim),
            ReLU(inplace=True),
            Linear(hidden_dim, output_dim),
        )
        self.reg_criterion = nn.L1Loss(reduction="none")
        self.cls_criterion = nn.BCEWithLogitsLoss(reduction="none")

    def forward(self, data_dict):
        check_sync_batchnorm(self)
        point_output = self.backbone(data_dict, _head=None if self.training else self.head)
        if "_head_out" in point_output.keys():   # head ran inside the native executor
            pred_flat = point_output.pop("_head_out")
        elif self.training:
            # Linear -> BatchNorm1d (batch statistics) + ReLU -> Linear, each a taped HIP Function
            hidden = self.head[1](self.head[0](point_output.feat), act=ops.ACT_RELU)
            pred_flat = self.head[3](hidden).float()
        else:
            feat = point_output.feat
            scale, shift = self.head[1].folded()
            h0, h3 = self.head[0], self.head[3]
            if ops.mlp2_fusable(h0.in_features, h0.out_features, h3.out_features, feat.dtype):
                w2 = h3._cache.get(("mlp2", feat.dtype), [h3.weight], lambda: ops.mlp2_weight2(h3.weight, feat.dtype))
                pred_flat = ops.mlp2(feat, h0.weight_for(feat.dtype), h0.bias_f32(), scale, shift, ops.ACT_RELU, w2,
                                     h3.bias_f32(), h3.out_features)
            else:
                hidden = h0(feat, bn_scale=scale, bn_shift=shift, act=ops.ACT_RELU)
                pred_flat = h3(hidden).float()
        pred = pred_flat.view(-1, self.num_keypoints, 4)

        result_dict = {}
        if "target" in data_dict:
            # masked L1 + BCE (:50-87): a handful of reductions over (N, K, 4) - loss plumbing on torch
            target = data_dict["target"]
            offset_gt, mask_gt = target[..., :3], target[..., 3]
            offset_pred, mask_logits = pred[..., :3], pred[..., 3]
            cls_loss = self.cls_criterion(mask_logits, mask_gt).mean()
            valid_mask_exp = (mask_gt > 0.5).float().unsqueeze(-1)
            raw_reg_loss = self.reg_criterion(offset_pred, offset_gt)
            reg_loss = (raw_reg_loss * valid_mask_exp).sum() / (valid_mask_exp.sum() * 3 + 1e-6)
            result_dict["loss"] = cls_loss + reg_loss * 2.0
            if self.training:
                with torch.no_grad():
                    result_dict["train/cls_loss"] = cls_loss.detach()
                    result_dict["train/reg_loss"] = reg_loss.detach()
                    result_dict["train/offset_l1_err"] = (
                        (torch.abs(offset_pred - offset_gt) * valid_mask_exp).sum()
                        / (valid_mask_exp.sum() * 3 + 1e-6))
        if not self.training:
            final_pred = pred.clone()
            final_pred[..., 3] = torch.sigmoid(pred[..., 3])
            result_dict["pred"] = final_pred
        return result_dict
