"""OffsetKeypointEvaluator on the HIP path: same hook name, constructor and reported quantities as the reference
(pointcept/engines/hooks/offset_keypoint_evaluator.py:10-123), but the per-sample x per-keypoint python loops
(:46-92, one .item() sync per keypoint) are two launches of ptv3_keypoint_aggregate per batch and the running
totals stay on the device until the epoch ends."""
import torch
import torch.distributed as dist

import pointcept.utils.comm as comm
from pointcept.engines.hooks.builder import HOOKS
from pointcept.engines.hooks import HookBase
from ptv3_hip import ops


def evaluate_batch(pred, target, coord, offset, scale=None):
    """Totals of one batch as a device vector [sum over samples of the mean keypoint distance, #samples with a
    valid keypoint, per-keypoint distance sums (K), per-keypoint sample counts (K)] (reference :46-92)."""
    pred, target = pred.float().contiguous(), target.float().contiguous()
    coord, offset = coord.float().contiguous(), offset.long().contiguous()
    pred_kp, _ = ops.keypoint_aggregate(coord, pred, offset, ops.KP_ARGMAX)        # :62-69 (normalised frame)
    gt_kp, cnt = ops.keypoint_aggregate(coord, target, offset, ops.KP_GT_MEAN)     # :49-60
    valid = cnt > 0
    dist_bk = torch.linalg.vector_norm(pred_kp - gt_kp, dim=-1)                    # (B, K) tiny: torch plumbing
    if scale is not None:
        dist_bk = dist_bk * scale.float().view(-1, 1)
    dist_bk = torch.where(valid, dist_bk, torch.zeros_like(dist_bk))
    nvalid = valid.sum(1)
    has = nvalid > 0
    sample_mean = dist_bk.sum(1) / nvalid.clamp(min=1)
    return torch.cat([sample_mean[has].sum().view(1), has.sum().float().view(1), dist_bk.sum(0),
                      valid.sum(0).float()])


@HOOKS.register_module()
class OffsetKeypointEvaluator(HookBase):
    def __init__(self, num_keypoints=6):
        self.num_keypoints = num_keypoints

    def after_epoch(self):
        if self.trainer.val_loader is not None:
            self.eval()

    def eval(self):
        self.trainer.model.eval()
        self.trainer.logger.info(">>>>>>>>>>>>>>>> Start Offset-based Evaluation >>>>>>>>>>>>>>>>")
        K = self.num_keypoints
        totals = None
        with torch.no_grad():
            for data_dict in self.trainer.val_loader:
                for key in data_dict.keys():
                    if isinstance(data_dict[key], torch.Tensor):
                        data_dict[key] = data_dict[key].cuda(non_blocking=True)
                pred = self.trainer.model(data_dict)["pred"]
                t = evaluate_batch(pred, data_dict["target"], data_dict["coord"], data_dict["offset"],
                                   data_dict.get("scale", None))
                totals = t if totals is None else totals + t
        if totals is None:
            totals = torch.zeros(2 + 2 * K, device="cuda")
        if comm.get_world_size() > 1:
            dist.all_reduce(totals)
        v = totals.tolist()   # the one host sync of the evaluation
        total_dist, total_samples = v[0], v[1]
        per_kp, per_cnt = v[2:2 + K], v[2 + K:2 + 2 * K]
        mean_dist = total_dist / (total_samples + 1e-6)
        log = self.trainer.logger.info
        log(f"Val Result: Mean Distance (Over all active keypoints and samples) = {mean_dist:.4f}")
        for k in range(K):
            k_mean = per_kp[k] / (per_cnt[k] + 1e-6)
            log(f"  Keypoint {k} Mean Distance: {k_mean:.4f} (Valid Samples Evaluated: {int(per_cnt[k])})")
            if self.trainer.writer is not None:
                self.trainer.writer.add_scalar(f"val/KP_{k}_MeanDist", k_mean, self.trainer.epoch + 1)
        if self.trainer.writer is not None:
            self.trainer.writer.add_scalar("val/MeanDist", mean_dist, self.trainer.epoch + 1)
        # negative: SaveBest keeps the larger value (reference :121-123)
        self.trainer.comm_info["current_metric_value"] = -mean_dist
        self.trainer.comm_info["current_metric_name"] = "mean_dist"
