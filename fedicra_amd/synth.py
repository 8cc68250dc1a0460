"""Synthetic FAZ/ODOC-like phantoms (SURVEY.md section 8d "Synthetic inputs").

There is no network and no h5py on the build/GPU boxes, so the metric is quoted on
synthetic slices of the reference's shapes: image in [0,1) (reference normalises to
[0,1], /root/reference/README.md:42), a filled disk (FAZ-like, 2 classes) or two
concentric disks (ODOC-like, 3 classes) on a noisy background, a dense mask, and a
weak label that is ``num_classes`` (= ignore, /root/reference/code/flower_pCE_2D.py:57)
everywhere except a random ``labeled_frac`` of pixels (scribble-like sparsity).
Pure numpy, seeded ``2022 + 1000*cid + epoch`` like the reference's seed 2022.
Batches are dicts {'image','label'} shaped like the reference DataLoader's
(FAZ: image [B,H,W], ODOC: [B,3,H,W]; label [B,H,W] uint8) -- flower_pCE_2D.py:76-81.
"""
from __future__ import annotations

import numpy as np

FAZ_SITE_SIZES = (244, 150, 200, 708, 30)     # real FAZ site sizes, SURVEY.md 2.1-20


def phantom_batch(batch: int, size: int, in_chns: int = 1, num_classes: int = 2, *, cid: int = 0,
                  epoch: int = 0, index: int = 0, labeled_frac: float = 0.05, dense: bool = False):
    """Returns (image float32, weak_label uint8, dense_mask uint8) numpy arrays."""
    rng = np.random.default_rng([2022 + 1000 * cid + epoch, index])
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    imgs = np.empty((batch, in_chns, size, size), np.float32)
    masks = np.zeros((batch, size, size), np.uint8)
    for b in range(batch):
        cy, cx = rng.uniform(size * 0.25, size * 0.75, 2)
        r = rng.uniform(size / 10.0, size / 4.0)
        d2 = (yy - cy) ** 2 + (xx - cx) ** 2
        base = rng.uniform(0.0, 0.3, (in_chns, size, size)).astype(np.float32)
        inner = d2 <= r * r
        base[:, inner] = rng.uniform(0.6, 1.0)
        masks[b][inner] = 1
        if num_classes >= 3:
            inner2 = d2 <= (0.5 * r) ** 2
            base[:, inner2] = rng.uniform(0.3, 0.6)
            masks[b][inner2] = 2
        base += rng.normal(0.0, 0.05, base.shape).astype(np.float32)
        imgs[b] = np.clip(base, 0.0, 0.999)
    if dense:
        weak = masks.copy()
    else:
        weak = np.full_like(masks, num_classes)
        pick = rng.random(masks.shape) < labeled_frac
        weak[pick] = masks[pick]
    if in_chns == 1:
        imgs = imgs[:, 0]                      # FAZ batches arrive [B,H,W] (flower_pCE_2D.py:77)
    return imgs, weak, masks


def client_num_batches(k: int, batch_size: int = 12):
    """n_k = len(trainloader) = ceil(site_size / batch) -- the FedAvg weight."""
    return [-(-FAZ_SITE_SIZES[i % len(FAZ_SITE_SIZES)] // batch_size) for i in range(k)]
