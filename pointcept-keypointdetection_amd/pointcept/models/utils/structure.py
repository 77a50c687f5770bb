"""Point: the batched point-cloud dict of Pointcept, with serialization / sparsify on the GPU.

Mirrors pointcept/models/utils/structure.py:20-146 of the reference (attribute dict; "offset" <->
"batch" derivation in __init__; serialization(order, depth, shuffle_orders); sparsify(pad)).
Space-filling-curve codes, the argsort and its inverse run in libptv3_hip.so.
"""
import torch

from ptv3_hip import ops
from .misc import offset2batch, batch2offset
from .sparse import SparseConvTensor


class AttrDict(dict):
    """dict with attribute access (the part of addict.Dict the reference relies on)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        try:
            del self[name]
        except KeyError:
            raise AttributeError(name)


class Point(AttrDict):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if "batch" not in self.keys() and "offset" in self.keys():
            n = None
            for key in ("feat", "coord", "grid_coord"):
                if key in self.keys() and torch.is_tensor(self[key]):
                    n = self[key].shape[0]
                    break
            self["batch"] = offset2batch(self.offset, n)
        elif "offset" not in self.keys() and "batch" in self.keys():
            self["offset"] = batch2offset(self.batch)

    # host copies of tiny per-scene quantities, fetched once per Point (one sync each)
    def offset_host(self):
        if "_offset_host" not in self.keys():
            self["_offset_host"] = [int(v) for v in self.offset.tolist()]
        return self["_offset_host"]

    def grid_max_host(self):
        if "_grid_max_host" not in self.keys():
            self["_grid_max_host"] = [int(v) for v in self.grid_coord.max(0).values.tolist()]
        return self["_grid_max_host"]

    def _ensure_grid_coord(self):
        if "grid_coord" not in self.keys():
            assert {"grid_size", "coord"}.issubset(self.keys())
            self["grid_coord"] = torch.div(self.coord - self.coord.min(0)[0], self.grid_size,
                                           rounding_mode="trunc").int()

    def serialization(self, order="z", depth=None, shuffle_orders=False):
        """structure.py:52-109.  relies on ["grid_coord" or "coord" + "grid_size", "batch"]."""
        self["order"] = order
        order = [order] if isinstance(order, str) else list(order)
        assert "batch" in self.keys()
        self._ensure_grid_coord()
        if depth is None:
            depth = int(max(self.grid_max_host()) + 1).bit_length()
        self["serialized_depth"] = depth
        nb = len(self.offset)
        assert depth * 3 + nb.bit_length() <= 63
        assert depth <= 16
        if shuffle_orders:
            # same CPU-RNG draw as the reference (structure.py:101-105); permuting the order list before
            # encoding equals permuting the rows afterwards
            perm = torch.randperm(len(order)).tolist()
            order = [order[p] for p in perm]
        gc = self.grid_coord
        if gc.dtype not in (torch.int32, torch.int64):
            gc = gc.long()
        code = ops.sfc_encode(gc.contiguous(), self.batch.long().contiguous(), depth, order)
        end_bit = max(1, depth * 3 + max(nb - 1, 0).bit_length())
        sorder, inverse = ops.argsort_codes(code, end_bit)
        self["serialized_code"] = code
        self["serialized_order"] = sorder
        self["serialized_inverse"] = inverse

    def sparsify(self, pad=96):
        """structure.py:111-146: prepares the sparse-conv tensor (site list; tables are built lazily)."""
        assert {"feat", "batch"}.issubset(self.keys())
        self._ensure_grid_coord()
        if "sparse_shape" in self.keys():
            sparse_shape = self.sparse_shape
        else:
            sparse_shape = [g + pad for g in self.grid_max_host()]
        indices = torch.cat([self.batch.unsqueeze(-1).int(), self.grid_coord.int()], dim=1).contiguous()
        t = SparseConvTensor(features=self.feat, indices=indices, spatial_shape=sparse_shape,
                             batch_size=len(self.offset))
        if "serialized_order" in self.keys():
            t.row_order = self.serialized_order[0].int()
        self["sparse_shape"] = sparse_shape
        self["sparse_conv_feat"] = t
