"""GridSample / Collect / ToTensor with the reference's names, constructor keywords and dict contract
(pointcept/datasets/transform.py:20-78, 826-964), running on the device.

The reference runs these in CPU dataloader workers on numpy arrays; here every per-point array of the sample is a
GPU tensor (numpy / CPU inputs are uploaded once) and the voxel hashing, key sort, unique and pick are HIP kernels
(ptv3_grid_hash, ptv3_argsort_i64, ptv3_pool_segments).  The one random draw of train mode
(`np.random.randint(0, count.max(), count.size)`, :864) is taken from numpy's global RNG with the same arguments,
so a seeded run consumes the RNG stream exactly like the reference; it needs `count.max()` on the host = the one
sync of the transform.  Equal hash keys are ordered stably (the reference's np.argsort leaves that order
unspecified, see oracle/gridsample.py)."""
from collections.abc import Sequence

import numpy as np
import torch

from pointcept.utils.registry import Registry
from ptv3_hip import ops

TRANSFORMS = Registry("transforms")


def _device():
    return torch.device("cuda", torch.cuda.current_device())


def _dev(v):
    if isinstance(v, np.ndarray):
        v = torch.from_numpy(v)
    return v.to(_device(), non_blocking=True) if torch.is_tensor(v) and not v.is_cuda else v


_DEFAULT_INDEX_KEYS = ("coord", "color", "normal", "superpoint", "strength", "segment", "instance")


def index_operator(data_dict, index, duplicate=False):
    """Row selection on every per-point entry named in data_dict["index_valid_keys"] (reference
    transform.py:23-50).  duplicate=False edits the dict in place; duplicate=True leaves it untouched and returns a
    shallow copy whose per-point entries are the selected rows."""
    valid = data_dict.setdefault("index_valid_keys", list(_DEFAULT_INDEX_KEYS))
    target = data_dict if not duplicate else {k: v for k, v in data_dict.items()}
    if duplicate:
        target["index_valid_keys"] = list(valid)
    for key in valid:
        if key in data_dict:
            target[key] = _dev(data_dict[key])[index]
    return target


@TRANSFORMS.register_module()
class Collect(object):
    """Picks the model inputs out of the sample (reference transform.py:54-78): `keys` are passed through, each
    `offset_keys_dict` entry stores the point count of the named array, and every extra `<name>_keys=[...]` keyword
    concatenates the listed arrays channel-wise as float32 under `<name>` (e.g. feat_keys -> "feat")."""

    def __init__(self, keys, offset_keys_dict=None, **kwargs):
        self.keys = [keys] if isinstance(keys, str) else keys
        self.offset_keys = dict(offset="coord") if offset_keys_dict is None else offset_keys_dict
        self.kwargs = kwargs

    def __call__(self, data_dict):
        out = {key: data_dict[key] for key in self.keys}
        out.update({name: torch.tensor([data_dict[src].shape[0]]) for name, src in self.offset_keys.items()})
        for name, parts in self.kwargs.items():
            if not isinstance(parts, Sequence):
                raise TypeError(f"Collect: {name} must list keys")
            out[name.replace("_keys", "")] = torch.cat([_dev(data_dict[k]).float() for k in parts], dim=1)
        return out


@TRANSFORMS.register_module()
class ToTensor(object):
    """numpy / python scalars -> torch (reference transform.py:81-111); tensors and strings pass through, containers
    are converted element-wise.  Uploading happens in the first device op that touches the entry."""

    def __call__(self, data):
        if torch.is_tensor(data) or isinstance(data, str):
            return data
        if isinstance(data, np.ndarray):
            return torch.from_numpy(data)
        if isinstance(data, (bool, np.bool_)):
            return torch.tensor([bool(data)])
        if isinstance(data, (int, np.integer)):
            return torch.tensor([int(data)], dtype=torch.long)
        if isinstance(data, (float, np.floating)):
            return torch.tensor([float(data)], dtype=torch.float32)
        if isinstance(data, dict):
            return {k: self(v) for k, v in data.items()}
        if isinstance(data, Sequence):
            return [self(v) for v in data]
        raise TypeError(f"type {type(data)} cannot be converted to tensor.")


@TRANSFORMS.register_module()
class GridSample(object):
    def __init__(self, grid_size=0.05, hash_type="fnv", mode="train", return_inverse=False, return_grid_coord=False,
                 return_min_coord=False, return_displacement=False, project_displacement=False):
        self.grid_size = grid_size
        self.hash_type = ops.HASH_FNV if hash_type == "fnv" else ops.HASH_RAVEL
        assert mode in ["train", "test"]
        self.mode = mode
        self.return_inverse = return_inverse
        self.return_grid_coord = return_grid_coord
        self.return_min_coord = return_min_coord
        self.return_displacement = return_displacement
        self.project_displacement = project_displacement

    def _plan(self, data_dict):
        coord = _dev(data_dict["coord"])
        if coord.dtype != torch.float32:
            raise TypeError("GridSample on the HIP path takes float32 coordinates (the datasets' dtype)")
        coord = coord.contiguous()
        data_dict["coord"] = coord
        grid, mm, key = ops.grid_hash(coord, self.grid_size, self.hash_type)
        idx_sort, inverse, seg_start, nvox = ops.voxel_unique(key)
        start = seg_start[:-1].long()
        count = (seg_start[1:] - seg_start[:-1]).long()
        return coord, grid, mm, idx_sort, inverse, start, count, nvox

    def _extras(self, out, coord, grid, mm, inverse, idx):
        if self.return_inverse:
            out["inverse"] = inverse
        if self.return_grid_coord:
            out["grid_coord"] = grid[idx]
            if "grid_coord" not in out["index_valid_keys"]:
                out["index_valid_keys"].append("grid_coord")
        if self.return_min_coord:
            out["min_coord"] = (mm[:3].double() * self.grid_size).reshape(1, 3)
        if self.return_displacement:
            # [0, 1] -> [-0.5, 0.5] displacement to the voxel centre (:883-893), float64 like numpy
            sel = coord[idx].double() / self.grid_size - mm[:3].double()
            disp = sel - grid[idx].double() - 0.5
            if self.project_displacement:
                disp = torch.sum(disp * _dev(out["normal"]).double(), dim=-1, keepdim=True)
            out["displacement"] = disp
            if "displacement" not in out["index_valid_keys"]:
                out["index_valid_keys"].append("displacement")

    def __call__(self, data_dict):
        assert "coord" in data_dict.keys()
        if "sampled_index" in data_dict:
            raise NotImplementedError("GridSample on the HIP path: the ScanNet data-efficient 'sampled_index' branch "
                                      "(:867-874) is not part of the keypoint path")
        coord, grid, mm, idx_sort, inverse, start, count, nvox = self._plan(data_dict)
        cmax = int(count.max().item()) if nvox > 0 else 1   # the transform's host sync (numpy needs it too)
        if self.mode == "train":
            r = np.random.randint(0, cmax, nvox)            # same draw as the reference (:864)
            idx_select = start + torch.from_numpy(r).to(count.device) % count
            idx_unique = idx_sort[idx_select]
            full_coord = coord
            data_dict = index_operator(data_dict, idx_unique)
            self._extras(data_dict, full_coord, grid, mm, inverse, idx_unique)
            return data_dict
        parts = []
        for i in range(cmax):
            idx_part = idx_sort[start + i % count]
            part = index_operator(data_dict, idx_part, duplicate=True)
            part["index"] = idx_part
            self._extras(part, coord, grid, mm, inverse, idx_part)
            parts.append(part)
        return parts
