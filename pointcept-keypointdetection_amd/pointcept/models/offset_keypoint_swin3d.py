"""Keypoint-offset head on the Swin3D backbone: registry name "OffsetKeypointSwin3D".

Contract of the reference class (pointcept/models/offset_keypoint_swin3d.py:5-130): constructor keywords
`backbone_conf, num_keypoints, hidden_dim`; state_dict = `backbone.*` + `head.{0,1,3}.*`; `coord_feat` is built from
`feat` (or `coord` + `feat` when the stem expects three more channels, :38-56) when the batch does not carry one; eval
returns `pred` (N, K, 4) with the score column through a sigmoid (:125-128) and, when the batch carries `target`, the
loss of :73-88 (BCE on the score + 2 x masked L1 on the offsets); training also returns the curves the reference's
InformationWriter logs (:92-124): train/cls_loss, train/reg_loss, train/offset_l1_err, train/mean_dist and
train/kp{i}_dist (per-keypoint mean distance of the valid points, in scene units when the batch carries `scale`) -
as detached 0-d DEVICE tensors (the reference calls .item() on each: 4 + K host syncs per step).
"""
import torch
import torch.nn as nn

from ptv3_hip import ops
from pointcept.models.utils.hip_layers import Linear, BatchNorm1d
from .builder import MODELS, build_model


@MODELS.register_module()
class OffsetKeypointSwin3D(nn.Module):
    def __init__(self, backbone_conf, num_keypoints=6, hidden_dim=256):
        super().__init__()
        self.backbone = build_model(backbone_conf)
        in_channels = backbone_conf["channels"][0] if "channels" in backbone_conf else 96
        self.num_keypoints = num_keypoints
        self.head = nn.Sequential(Linear(in_channels, hidden_dim), BatchNorm1d(hidden_dim), nn.ReLU(inplace=True),
                                  Linear(hidden_dim, num_keypoints * 4))
        self.reg_criterion = nn.L1Loss(reduction="none")
        self.cls_criterion = nn.BCEWithLogitsLoss(reduction="none")

    def forward(self, data_dict):
        if "coord_feat" not in data_dict:
            coord, feat = data_dict["coord"], data_dict["feat"]
            expected = self.backbone.stem_layer.conv_layers[0].in_channels
            data_dict["coord_feat"] = torch.cat([coord, feat], dim=1) if expected == feat.shape[1] + 3 else feat
        feat = self.backbone(data_dict)
        x = self.head[1](self.head[0](feat.contiguous()), act=ops.ACT_RELU)
        pred = self.head[3](x).float().view(-1, self.num_keypoints, 4)
        result = {}
        if "target" in data_dict:
            target = data_dict["target"]
            mask_gt = target[..., 3]
            cls_loss = self.cls_criterion(pred[..., 3], mask_gt).mean()
            valid = (mask_gt > 0.5).float().unsqueeze(-1)
            reg = (self.reg_criterion(pred[..., :3], target[..., :3]) * valid).sum() / (valid.sum() * 3 + 1e-6)
            result["loss"] = cls_loss + reg * 2.0
            if self.training:
                with torch.no_grad():
                    result["train/cls_loss"] = cls_loss.detach()
                    result["train/reg_loss"] = reg.detach()
                    result["train/offset_l1_err"] = ((torch.abs(pred[..., :3] - target[..., :3]) * valid).sum()
                                                     / (valid.sum() * 3 + 1e-6))
                    dist = torch.norm(pred[..., :3] - target[..., :3], p=2, dim=-1)          # (N, K)
                    if "scale" in data_dict and "offset" in data_dict:
                        offset = data_dict["offset"]
                        b = torch.zeros(int(dist.shape[0]), dtype=torch.long, device=dist.device)
                        if len(offset) > 1:
                            b[offset[:-1].long()] = 1
                        dist = dist * data_dict["scale"].view(-1)[torch.cumsum(b, dim=0)].unsqueeze(-1)
                    vm = (mask_gt > 0.5).float()
                    vsum = vm.sum(dim=0)
                    kp = (dist * vm).sum(dim=0) / vsum.clamp(min=1e-6)
                    kp = torch.where(vsum == 0, torch.zeros_like(kp), kp)
                    result["train/mean_dist"] = kp.mean()
                    for i in range(self.num_keypoints):
                        result[f"train/kp{i}_dist"] = kp[i]
        if not self.training:
            final = pred.clone()
            final[..., 3] = torch.sigmoid(pred[..., 3])
            result["pred"] = final
        return result
