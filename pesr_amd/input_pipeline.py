"""GPU-side input pipeline: the uint8 image pool lives in HBM; a batch of (LR, HR) training crops with
the reference's 8-way augmentation (reference data.py:79-126) is assembled by one kernel launch per resolution.
Replaces the reference's 4 DataLoader worker processes + pinned-memory copies (reference train.py:96-97) - at
8 x 127 patches/s the host path would have to deliver ~1000 crops/s."""
from __future__ import annotations

import random
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib

SCALE = 4


class GpuPatchSampler:
    def __init__(self, lr_images: Sequence[np.ndarray], hr_images: Sequence[np.ndarray], device: torch.device):
        assert len(lr_images) == len(hr_images) and len(lr_images) > 0
        self.device = device
        self.n = len(lr_images)
        self.lr_shapes = [im.shape for im in lr_images]
        for l, h in zip(lr_images, hr_images):
            assert l.dtype == np.uint8 and h.dtype == np.uint8 and l.shape[2] == 3
            assert h.shape[0] == SCALE * l.shape[0] and h.shape[1] == SCALE * l.shape[1], "HR must be 4x the LR image"
        self.lr_pool, self.lr_off = self._pool(lr_images)
        self.hr_pool, self.hr_off = self._pool(hr_images)

    def _pool(self, images):
        offs, total = [], 0
        for im in images:
            offs.append(total)
            total += im.size
        flat = np.concatenate([np.ascontiguousarray(im).reshape(-1) for im in images])
        return torch.from_numpy(flat).to(self.device), offs

    def draw(self, batch: int, patch: int, rng: Optional[random.Random] = None, augment: bool = True):
        """Host-side random choices, as the reference's random.randint calls: (image, y, x, aug) per sample."""
        rng = rng or random
        picks = []
        for _ in range(batch):
            i = rng.randrange(self.n)
            h, w, _ = self.lr_shapes[i]
            picks.append((i, rng.randint(0, h - patch), rng.randint(0, w - patch), rng.randint(0, 7) if augment else 0))
        return picks

    def draw_for(self, images: Sequence[int], patch: int, rng: Optional[random.Random] = None, augment: bool = True):
        """As draw(), for a GIVEN list of image indices (an epoch permutation dealt out by the caller): only the crop origin
        and the augmentation are random."""
        rng = rng or random
        picks = []
        for i in images:
            h, w, _ = self.lr_shapes[i]
            picks.append((i, rng.randint(0, h - patch), rng.randint(0, w - patch), rng.randint(0, 7) if augment else 0))
        return picks

    def assemble(self, picks: List[tuple], patch: int, nhwc: bool = False):
        """-> (lr [B,3,P,P], hr [B,3,4P,4P]) fp32 on the device (logical NCHW; channels_last memory when nhwc)."""
        B = len(picks)
        dl = np.empty((B, 3), dtype=np.int64)
        dh = np.empty((B, 3), dtype=np.int64)
        for b, (i, y, x, aug) in enumerate(picks):
            w = self.lr_shapes[i][1]
            dl[b] = (self.lr_off[i], w | (y << 32), x | (aug << 32))
            dh[b] = (self.hr_off[i], (SCALE * w) | ((SCALE * y) << 32), (SCALE * x) | (aug << 32))
        L = _lib.lib()
        s = torch.cuda.current_stream(self.device).cuda_stream
        outs = []
        for pool, d, P in ((self.lr_pool, dl, patch), (self.hr_pool, dh, SCALE * patch)):
            desc = torch.from_numpy(d).to(self.device)
            out = torch.empty((B, P, P, 3) if nhwc else (B, 3, P, P), dtype=torch.float32, device=self.device)
            _lib.check(L.pesr_crop_augment(pool.data_ptr(), desc.data_ptr(), out.data_ptr(), B, P, int(nhwc), s), "pesr_crop_augment")
            outs.append(out.permute(0, 3, 1, 2) if nhwc else out)
        return outs[0], outs[1]

    def sample(self, batch: int, patch: int, rng: Optional[random.Random] = None, augment: bool = True, nhwc: bool = False):
        return self.assemble(self.draw(batch, patch, rng, augment), patch, nhwc)
