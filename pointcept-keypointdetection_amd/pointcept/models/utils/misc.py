"""offset <-> batch helpers (reference: pointcept/models/utils/misc.py:13-34). Index plumbing on torch."""
import torch


@torch.no_grad()
def offset2bincount(offset):
    # new_zeros instead of the reference's torch.tensor([0], device=...): a host list -> device copy is a
    # blocking transfer on the current stream, i.e. a hidden pipeline drain in front of every forward
    return torch.diff(offset, prepend=offset.new_zeros(1))


@torch.no_grad()
def bincount2offset(bincount):
    return torch.cumsum(bincount, dim=0)


@torch.no_grad()
def offset2batch(offset, num_points=None):
    """num_points (= offset[-1], known from any per-point tensor) avoids the device->host read that
    repeat_interleave otherwise needs to size its output."""
    bincount = offset2bincount(offset)
    ids = torch.arange(len(bincount), device=offset.device, dtype=torch.long)
    if num_points is not None:
        return ids.repeat_interleave(bincount, output_size=int(num_points))
    return ids.repeat_interleave(bincount)


@torch.no_grad()
def batch2offset(batch):
    return torch.cumsum(batch.bincount(), dim=0).long()


def off_diagonal(x):
    n, m = x.shape
    assert n == m
    return x.flatten()[:-1].view(n - 1, n + 1)[:, 1:].flatten()
