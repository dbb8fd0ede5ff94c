"""Minimal data harness so that train.py / test.py run (the reference's data.py - PNG->npy cache, DIV2K/PIRM layout -
is OUT OF SCOPE; benchmarks use synthetic crops).  Two datasets with the reference's sample contract
(reference data.py:79-126): `(lr, hr)` float CHW tensors holding raw 0..255 values, hr = 4x lr, 8-way flip/transpose
augmentation, random LR-aligned crops.
"""
import glob
import os
import random

import numpy as np
import torch
from torch.utils.data import Dataset

SCALE = 4


def augment(lr, hr, idx):
    """idx in 0..7: bit 2 transpose, bit 1 vertical flip, bit 0 horizontal flip (HWC arrays)."""
    if idx & 4:
        lr, hr = lr.transpose(1, 0, 2), hr.transpose(1, 0, 2)
    if idx & 2:
        lr, hr = lr[::-1], hr[::-1]
    if idx & 1:
        lr, hr = lr[:, ::-1], hr[:, ::-1]
    return np.ascontiguousarray(lr), np.ascontiguousarray(hr)


def to_tensor(a):
    return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)).astype(np.float32))


class SyntheticSRDataset(Dataset):
    """DIV2K-shaped random crops: iid integers 0..255.  Deterministic per index."""

    def __init__(self, length, patch_size, seed=1234):
        self.length, self.ps, self.seed = length, patch_size, seed

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed + i)
        lr = torch.randint(0, 256, (3, self.ps, self.ps), generator=g).float()
        hr = torch.randint(0, 256, (3, SCALE * self.ps, SCALE * self.ps), generator=g).float()
        return lr, hr


class FolderSRDataset(Dataset):
    """<root>/LR/*.png with matching <root>/HR/*.png (PIL); random crop + augmentation when patch_size is given."""

    def __init__(self, root, patch_size=None, num_repeats=1, is_aug=False, fixed_length=None):
        from PIL import Image
        self._open = Image.open
        self.lr_paths = sorted(glob.glob(os.path.join(root, "LR", "*.png")))
        if fixed_length:
            self.lr_paths = self.lr_paths[:fixed_length]
        self.hr_paths = [os.path.join(root, "HR", os.path.basename(p)) for p in self.lr_paths]
        self.ps, self.rep, self.aug = patch_size, num_repeats, is_aug

    def __len__(self):
        return len(self.lr_paths) * self.rep

    def __getitem__(self, i):
        i %= len(self.lr_paths)
        lr = np.asarray(self._open(self.lr_paths[i]).convert("RGB"))
        hr = np.asarray(self._open(self.hr_paths[i]).convert("RGB"))
        if self.ps:
            y = random.randint(0, lr.shape[0] - self.ps)
            x = random.randint(0, lr.shape[1] - self.ps)
            lr = lr[y:y + self.ps, x:x + self.ps]
            hr = hr[SCALE * y:SCALE * (y + self.ps), SCALE * x:SCALE * (x + self.ps)]
        if self.aug:
            lr, hr = augment(lr, hr, random.randint(0, 7))
        return to_tensor(lr), to_tensor(hr)
