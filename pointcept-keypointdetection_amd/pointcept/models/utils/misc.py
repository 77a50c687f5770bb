"""offset <-> batch index helpers with the names of the reference's pointcept/models/utils/misc.py:13-34
(pure index plumbing, evaluated by torch).  Two departures, both to keep the launch stream free of hidden
host synchronisation: the leading zero is created on the device (`new_zeros`, not a host list), and
`offset2batch` takes the known number of points so that `repeat_interleave` need not read its size back."""
import torch


def _no_grad(fn):
    return torch.no_grad()(fn)


@_no_grad
def offset2bincount(offset):
    """cumulative scene ends -> points per scene"""
    return torch.diff(offset, prepend=offset.new_zeros(1))


@_no_grad
def bincount2offset(bincount):
    return bincount.cumsum(0)


@_no_grad
def offset2batch(offset, num_points=None):
    counts = offset2bincount(offset)
    scene = torch.arange(counts.numel(), device=offset.device, dtype=torch.long)
    extra = {} if num_points is None else {"output_size": int(num_points)}
    return scene.repeat_interleave(counts, **extra)


@_no_grad
def batch2offset(batch):
    return batch.bincount().cumsum(0).long()


def off_diagonal(x):
    """all entries of a square matrix except its diagonal, flattened (row-major)"""
    rows, cols = x.shape
    if rows != cols:
        raise AssertionError("off_diagonal expects a square matrix")
    return x.flatten()[:-1].view(rows - 1, rows + 1)[:, 1:].flatten()
