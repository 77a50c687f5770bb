"""Oracle (test infrastructure): the reference's per-sample / per-keypoint loops, restated with torch-CPU.

  * evaluator_totals      engines/hooks/offset_keypoint_evaluator.py:46-92
  * infer_keypoints       tools/infer_offset.py:555-597 (agg_method "argmax" | "weighted")
"""
import torch


def evaluator_totals(pred, target, coord, offset, scale, num_kps):
    total_dist, total_samples = 0.0, 0
    per_kp, per_cnt = [0.0] * num_kps, [0] * num_kps
    for b in range(len(offset)):
        start = 0 if b == 0 else int(offset[b - 1])
        end = int(offset[b])
        b_coord, b_pred, b_target = coord[start:end], pred[start:end], target[start:end]
        b_scale = scale[b].item() if scale is not None else 1.0
        b_sum, b_valid = 0.0, 0
        for k in range(num_kps):
            valid_idx = torch.nonzero(b_target[:, k, 3] > 0).squeeze(-1)
            if len(valid_idx) == 0:
                continue
            gt_kp = (b_coord[valid_idx] + b_target[valid_idx, k, :3]).mean(dim=0)
            best = torch.argmax(b_pred[:, k, 3])
            pred_kp = b_coord[best] + b_pred[best, k, :3]
            d = (torch.norm(pred_kp - gt_kp, p=2) * b_scale).item()
            per_kp[k] += d
            per_cnt[k] += 1
            b_sum += d
            b_valid += 1
        if b_valid > 0:
            total_dist += b_sum / b_valid
            total_samples += 1
    return [total_dist, float(total_samples)] + per_kp + [float(c) for c in per_cnt]


def infer_keypoints(pred, target, coord, offset, scale, centroid, num_kps, agg_method="argmax", mask_thresh=0.5):
    B = len(offset)
    pred_kps = torch.zeros(B, num_kps, 3)
    target_kps = torch.full((B, num_kps, 3), float("nan"))
    for b in range(B):
        start = 0 if b == 0 else int(offset[b - 1])
        end = int(offset[b])
        s_coord = coord[start:end]
        s_pred_mask, s_pred_off = pred[start:end, :, 3], pred[start:end, :, :3]
        s_target_mask, s_target_off = target[start:end, :, 3], target[start:end, :, :3]
        s_scale, s_centroid = scale[b], centroid[b]
        s_true = s_coord * s_scale + s_centroid
        for k in range(num_kps):
            probs = s_pred_mask[:, k]
            if len(probs) == 0:
                continue
            if agg_method == "argmax":
                i = torch.argmax(probs)
                kp = s_true[i] + s_pred_off[i, k] * s_scale
            else:
                valid = probs > mask_thresh
                if torch.any(valid):
                    cand = s_true[valid] + s_pred_off[valid, k] * s_scale
                    w = probs[valid] / torch.sum(probs[valid])
                    kp = torch.sum(cand * w.unsqueeze(1), dim=0)
                else:
                    i = torch.argmax(probs)
                    kp = s_true[i] + s_pred_off[i, k] * s_scale
            pred_kps[b, k] = kp
            vi = torch.where(s_target_mask[:, k] > 0.5)[0]
            if len(vi) > 0:
                target_kps[b, k] = s_true[vi[0]] + s_target_off[vi[0], k] * s_scale
    return pred_kps, target_kps
