"""PointModule / PointSequential / PointModel (reference: pointcept/models/modules.py:28-120)."""
from collections import OrderedDict

import torch.nn as nn

from pointcept.models.utils.structure import Point
from pointcept.models.utils.sparse import SparseConvTensor, is_spconv_module
from pointcept.engines.hooks import HookBase


class PointModule(nn.Module):
    """Modules deriving from this take and return a Point inside PointSequential."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)


class PointSequential(PointModule):
    """Ordered container that hands each child the view of the data it understands (behaviour of the reference's
    models/modules.py:78-111): a PointModule receives the Point itself; a sparse-conv module receives
    `point.sparse_conv_feat` (its features are mirrored back to `point.feat`); any other nn.Module receives
    `point.feat` (and the sparse tensor is re-pointed at the result).  Bare tensors / sparse tensors are passed
    straight through the children."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        children = args[0].items() if len(args) == 1 and isinstance(args[0], OrderedDict) else \
            ((str(i), m) for i, m in enumerate(args))
        for name, module in children:
            self.add_module(name, module)
        for name, module in kwargs.items():
            if name in self._modules:
                raise ValueError("name exists.")
            self.add_module(name, module)

    def __len__(self):
        return len(self._modules)

    def __getitem__(self, idx):
        n = len(self)
        if not -n <= idx < n:
            raise IndexError("index {} is out of range".format(idx))
        return list(self._modules.values())[idx % n]

    def add(self, module, name=None):
        if name is None:
            name = str(len(self._modules))
            if name in self._modules:
                raise KeyError("name exists")
        self.add_module(name, module)

    @staticmethod
    def _through_point(module, point):
        if isinstance(module, PointModule):
            return module(point)
        if is_spconv_module(module):
            point.sparse_conv_feat = module(point.sparse_conv_feat)
            point.feat = point.sparse_conv_feat.features
            return point
        point.feat = module(point.feat)
        if "sparse_conv_feat" in point.keys():
            point.sparse_conv_feat = point.sparse_conv_feat.replace_feature(point.feat)
        return point

    @staticmethod
    def _through_other(module, value):
        if isinstance(module, PointModule) or is_spconv_module(module):
            return module(value)
        if isinstance(value, SparseConvTensor):
            return value.replace_feature(module(value.features)) if value.indices.shape[0] != 0 else value
        return module(value)

    def forward(self, input):
        for module in self._modules.values():
            step = self._through_point if isinstance(input, Point) else self._through_other
            input = step(module, input)
        return input


class PointModel(PointModule, HookBase):
    """Placeholder: a PointModel can be customised as a Pointcept hook."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
