"""Writes a synthetic fundus-like dataset in the reference's on-disk layout (for smoke runs and tests:
there is no network for REFUGE / Drishti-GS / RIM-ONE-r3)."""
import os

import numpy as np
from PIL import Image


def write_dataset(root, dataset, split, n, size=512, seed=0):
    rs = np.random.RandomState(seed)
    img_dir = os.path.join(root, dataset, split, 'ROIs', 'image')
    msk_dir = os.path.join(root, dataset, split, 'ROIs', 'mask')
    os.makedirs(img_dir, exist_ok=True)
    os.makedirs(msk_dir, exist_ok=True)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    for i in range(n):
        cy, cx = (0.4 + 0.2 * rs.rand(2)) * size
        a, b = (0.18 + 0.09 * rs.rand(2)) * size
        k = 0.4 + 0.3 * rs.rand()
        r = np.sqrt(((yy - cy) / a) ** 2 + ((xx - cx) / b) ** 2)
        grey = np.full((size, size), 255, np.uint8)
        grey[r <= 1] = 128
        grey[r <= k] = 0
        base = 60 + 40 * rs.rand(size, size, 3)
        base += (r <= 1)[..., None] * 60 + (r <= k)[..., None] * 50
        Image.fromarray(base.clip(0, 255).astype(np.uint8)).save(os.path.join(img_dir, 'img_%03d.png' % i))
        Image.fromarray(grey).save(os.path.join(msk_dir, 'img_%03d.png' % i))
    return os.path.join(root, dataset, split)
