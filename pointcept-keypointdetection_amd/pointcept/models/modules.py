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
    """Sequential container dispatching on module kind (modules.py:78-111): PointModule -> Point,
    sparse-conv module -> point.sparse_conv_feat (features mirrored to point.feat),
    plain nn.Module -> point.feat (sparse tensor features kept in sync)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            if name in self._modules:
                raise ValueError("name exists.")
            self.add_module(name, module)

    def __getitem__(self, idx):
        if not (-len(self) <= idx < len(self)):
            raise IndexError("index {} is out of range".format(idx))
        if idx < 0:
            idx += len(self)
        return list(self._modules.values())[idx]

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        if name is None:
            name = str(len(self._modules))
            if name in self._modules:
                raise KeyError("name exists")
        self.add_module(name, module)

    def forward(self, input):
        for _, module in self._modules.items():
            if isinstance(module, PointModule):
                input = module(input)
            elif is_spconv_module(module):
                if isinstance(input, Point):
                    input.sparse_conv_feat = module(input.sparse_conv_feat)
                    input.feat = input.sparse_conv_feat.features
                else:
                    input = module(input)
            else:
                if isinstance(input, Point):
                    input.feat = module(input.feat)
                    if "sparse_conv_feat" in input.keys():
                        input.sparse_conv_feat = input.sparse_conv_feat.replace_feature(input.feat)
                elif isinstance(input, SparseConvTensor):
                    if input.indices.shape[0] != 0:
                        input = input.replace_feature(module(input.features))
                else:
                    input = module(input)
        return input


class PointModel(PointModule, HookBase):
    """Placeholder: a PointModel can be customised as a Pointcept hook."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
