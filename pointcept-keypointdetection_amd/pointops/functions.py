"""The `pointops` call surface on top of the HIP entry points in `pointops._C`.

What callers of the reference's libs/pointops rely on (functions/__init__.py:1-14; call sites: engines/test.py:939-945,
hooks/evaluator.py:569-575, the PTv2 / Sonata model files):

    idx, dist = knn_query(nsample, xyz, offset, new_xyz=None, new_offset=None)   idx int32 (-1 = no neighbour), dist = sqrt(d2)
    grouping(idx, feat, xyz, new_xyz=None, with_xyz=False)       (m, nsample, [3 +] c); -1 gathers a zero row
    grouping2(feat, idx)                                          the bare gather
    interpolation(xyz, new_xyz, feat, offset, new_offset, k=3)   inverse-distance blend of the k nearest rows
    interpolation2(...)                                           same arguments
    knn_query_and_group(...), offset2batch, batch2offset

Both spellings of grouping / interpolation run the same HIP kernels here (the reference keeps a torch-indexing and
a CUDA variant of each); gradients with respect to the features flow through the matching backward kernels.
"""
import torch

from . import _C


def _i32(t):
    return t if t.dtype == torch.int32 and t.is_contiguous() else t.to(torch.int32).contiguous()


# nsample = 1 over at least this many (query, candidate) pairs goes through the cell grid: the result is identical
# (ptv3_knn_query_cells) and the label-mapping calls of the tester / evaluator (engines/test.py:939-945,
# hooks/evaluator.py:569-575: 0.2-1M queries against ~100k voxels) stop costing a scan of the scene per query
_CELL_GRID_PAIRS = 1 << 30


@torch.no_grad()
def knn_query(nsample, xyz, offset, new_xyz=None, new_offset=None, cell=None):
    """k nearest candidates of every query inside its own scene (libs/pointops/functions/query.py:7-24).
    xyz (n, 3) candidates / offset (b) cumulative ends; new_xyz (m, 3) queries / new_offset (b); queries default to
    the candidates themselves.  Rows are ascending in distance; scenes with fewer than nsample candidates pad with
    idx -1, dist sqrt(1e10).
    `cell` (not in the reference): edge of a uniform grid to search through instead of scanning the scene - same
    distances, ties resolved by index (see ptv3_knn_query_cells); used by the Swin3D down / upsampling, whose
    candidates are voxels of known size."""
    if new_xyz is None or new_offset is None:
        new_xyz, new_offset = xyz, offset
    if not (xyz.is_contiguous() and new_xyz.is_contiguous()):
        raise AssertionError("knn_query: xyz / new_xyz must be contiguous")
    m = new_xyz.shape[0]
    from ptv3_hip import ops
    if cell is not None or (nsample == 1 and m * xyz.shape[0] >= _CELL_GRID_PAIRS):
        idx, dist2 = ops.knn_query_cells(nsample, xyz.float(), _i32(offset), new_xyz.float(), _i32(new_offset), cell)
        return idx, dist2.sqrt_()
    ops._check_offsets("knn_query", offset, xyz.shape[0], new_offset, m)   # the kernel trusts the scene ends
    idx = torch.zeros((m, nsample), dtype=torch.int32, device=xyz.device)
    dist2 = torch.zeros((m, nsample), dtype=torch.float32, device=xyz.device)
    _C.knn_query_cuda(m, nsample, xyz, new_xyz, _i32(offset), _i32(new_offset), idx, dist2)
    return idx, dist2.sqrt_()


class _RowGather(torch.autograd.Function):
    """out[i, s, :] = table[idx[i, s], :] (zero row for idx -1); backward = scatter-add of the row gradients."""

    @staticmethod
    def forward(ctx, table, idx):
        if not (table.is_contiguous() and idx.is_contiguous()):
            raise AssertionError("grouping: input / idx must be contiguous")
        m, nsample = idx.shape
        rows, width = table.shape
        out = torch.zeros((m, nsample, width), dtype=torch.float32, device=table.device)
        _C.grouping_forward_cuda(m, nsample, width, table, idx, out)
        ctx.rows = rows
        ctx.save_for_backward(idx)
        return out

    @staticmethod
    def backward(ctx, grad):
        (idx,) = ctx.saved_tensors
        m, nsample, width = grad.shape
        acc = torch.zeros((ctx.rows, width), dtype=torch.float32, device=grad.device)
        _C.grouping_backward_cuda(m, nsample, width, grad.contiguous(), idx, acc)
        return acc, None


def grouping2(input, idx):
    return _RowGather.apply(input, idx)


def grouping(idx, feat, xyz, new_xyz=None, with_xyz=False):
    """Neighbourhood features (libs/pointops/functions/grouping.py:41-63).  With `with_xyz` the neighbour offsets
    relative to the query come first; a missing neighbour (-1) contributes zeros to both parts."""
    if not (xyz.is_contiguous() and feat.is_contiguous()):
        raise AssertionError("grouping: xyz / feat must be contiguous")
    centre = xyz if new_xyz is None else new_xyz
    idx = idx.contiguous()
    parts = [_RowGather.apply(feat, idx)]
    if with_xyz:
        if not centre.is_contiguous():
            raise AssertionError("grouping: new_xyz must be contiguous")
        present = (idx >= 0).to(torch.float32).unsqueeze(-1)
        parts.insert(0, (_RowGather.apply(xyz, idx) - centre.unsqueeze(1)) * present)
    return parts[0] if len(parts) == 1 else torch.cat(parts, dim=-1)


class _KnnBlend(torch.autograd.Function):
    """out[i] = sum_j weight[i, j] * table[idx[i, j]]; the gradient reaches `table` only (weights come out of a
    no-grad neighbour search, as in the reference: libs/pointops/functions/interpolation.py:28-61)."""

    @staticmethod
    def forward(ctx, table, idx, weight):
        n, k = idx.shape
        width = table.shape[1]
        out = torch.zeros((n, width), dtype=torch.float32, device=table.device)
        _C.interpolation_forward_cuda(n, width, k, table, idx, weight, out)
        ctx.rows = table.shape[0]
        ctx.save_for_backward(idx, weight)
        return out

    @staticmethod
    def backward(ctx, grad):
        idx, weight = ctx.saved_tensors
        n, width = grad.shape
        acc = torch.zeros((ctx.rows, width), dtype=torch.float32, device=grad.device)
        _C.interpolation_backward_cuda(n, width, idx.shape[1], grad.contiguous(), idx, weight, acc)
        return acc, None, None


def interpolation(xyz, new_xyz, feat, offset, new_offset, k=3, cell=None):
    """Features of `xyz` carried to `new_xyz` by normalised 1 / (distance + 1e-8) weights over the k nearest
    (libs/pointops/functions/interpolation.py:8-25).  `cell`: as in knn_query."""
    if not (xyz.is_contiguous() and new_xyz.is_contiguous() and feat.is_contiguous()):
        raise AssertionError("interpolation: xyz / new_xyz / feat must be contiguous")
    idx, dist = knn_query(k, xyz, offset, new_xyz, new_offset, cell=cell)
    w = torch.reciprocal(dist + 1e-8)
    w = (w / w.sum(dim=1, keepdim=True)).contiguous()
    return _KnnBlend.apply(feat, idx, w)


interpolation2 = interpolation


def knn_query_and_group(feat, xyz, offset=None, new_xyz=None, new_offset=None, idx=None, nsample=None,
                        with_xyz=False):
    """libs/pointops/functions/utils.py:5-18: search (unless `idx` is given), then grouping()."""
    if idx is None:
        if nsample is None:
            raise AssertionError("knn_query_and_group: nsample is required when idx is not given")
        idx, _ = knn_query(nsample, xyz, offset, new_xyz, new_offset)
    return grouping(idx, feat, xyz, new_xyz, with_xyz), idx


def offset2batch(offset):
    """cumulative scene ends -> scene id per point (libs/pointops/functions/utils.py offset2batch)"""
    ends = offset.long()
    sizes = torch.diff(ends, prepend=ends.new_zeros(1))
    return torch.repeat_interleave(torch.arange(ends.numel(), device=offset.device), sizes)


def batch2offset(batch):
    return batch.bincount().cumsum(0).int()
