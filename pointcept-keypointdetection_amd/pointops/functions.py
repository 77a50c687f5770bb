"""Python wrappers with the reference's call surface (libs/pointops/functions/{query,grouping,
interpolation,utils}.py) over the HIP entry points in pointops._C."""
import torch
from torch.autograd import Function

from ._C import (knn_query_cuda, grouping_forward_cuda, grouping_backward_cuda, interpolation_forward_cuda,
                 interpolation_backward_cuda)


class KNNQuery(Function):
    @staticmethod
    def forward(ctx, nsample, xyz, offset, new_xyz=None, new_offset=None):
        """xyz (n,3), new_xyz (m,3), offset (b), new_offset (b) -> idx (m,nsample) int32 (-1 pad), dist (m,nsample)."""
        if new_xyz is None or new_offset is None:
            new_xyz = xyz
            new_offset = offset
        assert xyz.is_contiguous() and new_xyz.is_contiguous()
        m = new_xyz.shape[0]
        idx = torch.zeros((m, nsample), dtype=torch.int, device=xyz.device)
        dist2 = torch.zeros((m, nsample), dtype=torch.float, device=xyz.device)
        knn_query_cuda(m, nsample, xyz, new_xyz, offset.int().contiguous(), new_offset.int().contiguous(), idx, dist2)
        return idx, torch.sqrt(dist2)


knn_query = KNNQuery.apply


class Grouping(Function):
    @staticmethod
    def forward(ctx, input, idx):
        assert input.is_contiguous() and idx.is_contiguous()
        m, nsample, n, c = idx.shape[0], idx.shape[1], input.shape[0], input.shape[1]
        output = torch.zeros((m, nsample, c), dtype=torch.float, device=input.device)
        grouping_forward_cuda(m, nsample, c, input, idx, output)
        ctx.n = n
        ctx.save_for_backward(idx)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        n = ctx.n
        (idx,) = ctx.saved_tensors
        m, nsample, c = grad_output.shape
        grad_input = torch.zeros((n, c), dtype=torch.float, device=idx.device)
        grouping_backward_cuda(m, nsample, c, grad_output.contiguous(), idx, grad_input)
        return grad_input, None


grouping2 = Grouping.apply


def grouping(idx, feat, xyz, new_xyz=None, with_xyz=False):
    """functions/grouping.py:41-63: -1 indices read an appended zero row."""
    if new_xyz is None:
        new_xyz = xyz
    assert xyz.is_contiguous() and feat.is_contiguous()
    m, nsample, c = idx.shape[0], idx.shape[1], feat.shape[1]
    grouped_feat = Grouping.apply(feat, idx.contiguous()) if not feat.requires_grad else _group_torch(feat, idx)
    if with_xyz:
        assert new_xyz.is_contiguous()
        mask = torch.sign(idx + 1)
        grouped_xyz = Grouping.apply(xyz, idx.contiguous()) - new_xyz.unsqueeze(1)
        grouped_xyz = torch.einsum("n s c, n s -> n s c", grouped_xyz, mask.to(grouped_xyz.dtype))
        return torch.cat((grouped_xyz, grouped_feat), -1)
    return grouped_feat


def _group_torch(feat, idx):
    m, nsample, c = idx.shape[0], idx.shape[1], feat.shape[1]
    feat = torch.cat([feat, torch.zeros([1, c], device=feat.device, dtype=feat.dtype)], dim=0)
    return feat[idx.view(-1).long(), :].view(m, nsample, c)


class Interpolation(Function):
    @staticmethod
    def forward(ctx, xyz, new_xyz, input, offset, new_offset, k=3):
        assert xyz.is_contiguous() and new_xyz.is_contiguous() and input.is_contiguous()
        idx, dist = knn_query(k, xyz, offset, new_xyz, new_offset)
        dist_recip = 1.0 / (dist + 1e-8)
        norm = torch.sum(dist_recip, dim=1, keepdim=True)
        weight = (dist_recip / norm).contiguous()
        n, c, m = new_xyz.shape[0], input.shape[1], input.shape[0]
        output = torch.zeros((n, c), dtype=torch.float, device=xyz.device)
        interpolation_forward_cuda(n, c, k, input, idx, weight, output)
        ctx.m, ctx.k = m, k
        ctx.save_for_backward(idx, weight)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        m, k = ctx.m, ctx.k
        idx, weight = ctx.saved_tensors
        n, c = grad_output.shape
        grad_input = torch.zeros((m, c), dtype=torch.float, device=idx.device)
        interpolation_backward_cuda(n, c, k, grad_output.contiguous(), idx, weight, grad_input)
        return None, None, grad_input, None, None, None


interpolation2 = Interpolation.apply


def interpolation(xyz, new_xyz, feat, offset, new_offset, k=3):
    """functions/interpolation.py:8-25 (no autograd through the op in the reference either)."""
    return Interpolation.apply(xyz, new_xyz, feat.contiguous(), offset, new_offset, k)


def knn_query_and_group(feat, xyz, offset=None, new_xyz=None, new_offset=None, idx=None, nsample=None,
                        with_xyz=False):
    if idx is None:
        assert nsample is not None
        idx, _ = knn_query(nsample, xyz, offset, new_xyz, new_offset)
    return grouping(idx, feat, xyz, new_xyz, with_xyz), idx


def offset2batch(offset):
    counts = torch.diff(offset.long(), prepend=torch.zeros(1, dtype=torch.long, device=offset.device))
    return torch.arange(len(offset), device=offset.device).repeat_interleave(counts).long()


def batch2offset(batch):
    return torch.cumsum(batch.bincount(), dim=0).int()
