"""Space-filling-curve codes on the GPU (reference: utils/serialization/default.py:9-24).

Same call surface as the reference's `encode`; the arithmetic is ptv3_sfc_encode (HIP)."""
import torch

from ptv3_hip import ops


@torch.no_grad()
def encode(grid_coord, batch=None, depth=16, order="z"):
    assert order in {"z", "z-trans", "hilbert", "hilbert-trans"}
    gc = grid_coord if grid_coord.dtype in (torch.int32, torch.int64) else grid_coord.long()
    b = None if batch is None else batch.long().contiguous()
    return ops.sfc_encode(gc.contiguous(), b, depth, [order])[0]


def z_order_encode(grid_coord, depth=16):
    return encode(grid_coord, None, depth, "z")


def hilbert_encode(grid_coord, depth=16):
    return encode(grid_coord, None, depth, "hilbert")
