"""Batch assembly for the point path: the counterpart of collate_fn / point_collate_fn
(reference: pointcept/datasets/utils.py:16-120).  Per-point tensors that already live on the device are
concatenated there; every key containing "offset" carries per-sample point counts and becomes the cumulative
scene-end vector the models consume.  The image/"correspondence" branches of the reference's collate belong to
its 2D-3D datasets and are not part of the keypoint path."""
import random
from collections.abc import Mapping, Sequence

import torch


def _merge_offsets(per_sample):
    """[cumulative ends of sample 0, of sample 1, ...] -> cumulative ends of the merged batch."""
    counts = [o - torch.cat([o.new_zeros(1), o[:-1]]) for o in per_sample]
    return torch.cumsum(torch.cat(counts), dim=0)


def collate_fn(batch):
    if not isinstance(batch, Sequence):
        raise TypeError(f"{type(batch)} is not supported.")
    head = batch[0]
    if torch.is_tensor(head):
        return torch.stack(list(batch)) if head.ndim == 0 else torch.cat(list(batch))
    if isinstance(head, str):
        return list(batch)
    if isinstance(head, Mapping):
        return {key: (_merge_offsets([s[key] for s in batch]) if "offset" in key
                      else collate_fn([s[key] for s in batch])) for key in head}
    if isinstance(head, Sequence):
        # list samples: the first entry's length is appended as the per-sample count, then made cumulative
        cols = [list(s) + [torch.tensor([s[0].shape[0]])] for s in batch]
        merged = [collate_fn(col) for col in zip(*cols)]
        merged[-1] = torch.cumsum(merged[-1], dim=0).int()
        return merged
    from torch.utils.data.dataloader import default_collate
    return default_collate(batch)


def point_collate_fn(batch, mix_prob=0):
    """collate + Mix3D-style pairing: with probability mix_prob consecutive scene pairs are fused into one scene
    (instance ids of the second scene shifted past the first's, every other scene end dropped)."""
    if not isinstance(batch[0], Mapping):
        raise TypeError("point_collate_fn takes dict samples")
    merged = collate_fn(batch)
    if random.random() >= mix_prob:
        return merged
    if "instance" in merged:
        ends = merged["offset"].tolist()
        begin, shift = 0, 0
        for i, end in enumerate(ends):
            seg = merged["instance"][begin:end]
            if i % 2 == 0:
                shift = seg.max()
            else:
                seg += shift * (seg != -1)
            begin = end
    for key in [k for k in merged if "offset" in k]:
        ends = merged[key]
        merged[key] = torch.cat([ends[1:-1:2], ends[-1:]])
    return merged
