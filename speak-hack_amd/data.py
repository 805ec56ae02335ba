"""Synthetic stand-in for the reference's datasets.

``CelebADataset.py:133-138`` / ``AffectNetDataset`` yield dicts with ``source_image``, ``target_image``
(float32 [3,256,256], normalised to [-1,1] by ``Normalize([0.5],[0.5])``, train.py:374-379) and
``emotion_labels_s`` / ``emotion_labels_t`` (int64 scalars, 8 classes -- the width of ``IRFD.Cm``, model.py:56).
The real datasets need files / a hub download that do not exist here; this one generates deterministic images of the
same schema so that the reference's training iteration can be driven end to end.
"""
from __future__ import annotations

import torch
from torch.utils.data import DataLoader, Dataset


class SyntheticFacePairs(Dataset):
    """len(self) pairs; item i is a pure function of (seed, i)."""

    def __init__(self, length=64, resolution=256, num_emotions=8, seed=0):
        self.length, self.resolution, self.num_emotions, self.seed = length, resolution, num_emotions, seed

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1_000_003 + i)
        r = self.resolution
        return {
            "source_image": torch.rand(3, r, r, generator=g) * 2 - 1,
            "target_image": torch.rand(3, r, r, generator=g) * 2 - 1,
            "emotion_labels_s": torch.randint(0, self.num_emotions, (), generator=g, dtype=torch.long),
            "emotion_labels_t": torch.randint(0, self.num_emotions, (), generator=g, dtype=torch.long),
        }


def synthetic_loader(batch_size=8, length=64, resolution=256, seed=0, shuffle=False):
    return DataLoader(SyntheticFacePairs(length, resolution, seed=seed), batch_size=batch_size, shuffle=shuffle, drop_last=True)
