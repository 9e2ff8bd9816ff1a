"""Micro-benchmark of the device-side input-pipeline tail at the bench shape: uda_normalize_tf and
uda_field_smooth + uda_elastic_warp on a uint8 batch of 512x512 samples, next to the same work done the reference's way
(numpy + scipy.ndimage, one sample at a time, one host core - what one dataloader worker does).

Algorithmic HBM bytes per sample (H*W = 262144 pixels): normalize_tf reads 3 + 1 B/px and writes (3 + 2 + 1) * 4 B/px = 28 B/px
= 7.34 MB; the elastic warp reads and writes 4 B/px of uint8 plus two float fields = 16 B/px, the field smoothing reads and
writes 2 * 2 float planes twice = 32 B/px."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from kernel_cases import _fundus_like, gen
from uda_clr_amd import ops
from uda_clr_amd.dataloaders import custom_transforms as tr
from uda_clr_amd.kernels import HipKernels

dev = torch.device("cuda:0")
K = HipKernels()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S = 512


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


img, lab = _fundus_like(B, S, S, gen(1))
iu, lu = torch.from_numpy(img).to(dev), torch.from_numpy(lab).to(dev)
px = B * S * S
ms = timeit(lambda: K.normalize_tf(iu, lu))
print("normalize_tf      B=%d %dx%d  %7.3f ms  %8.1f samples/s  %7.1f GB/s algorithmic (28 B/px)" % (B, S, S, ms, B / ms * 1e3, 28.0 * px / ms / 1e6))
noise = torch.rand(2, B, S, S, device=dev, dtype=torch.float64) * 2 - 1
ms_f = timeit(lambda: K.field_smooth(noise, 0.08 * S, 2.0 * S))
fld = K.field_smooth(noise, 0.08 * S, 2.0 * S)
print("field_smooth      B=%d           %7.3f ms  %8.1f samples/s  %7.1f GB/s algorithmic (64 B/px of float64; radius %d taps per axis)"
      % (B, ms_f, B / ms_f * 1e3, 64.0 * px / ms_f / 1e6, int(4 * 0.08 * S + 0.5)))
ap = torch.ones(B, dtype=torch.uint8, device=dev)
ms_w = timeit(lambda: K.elastic_warp(iu, lu, fld[0], fld[1], ap))
print("elastic_warp      B=%d           %7.3f ms  %8.1f samples/s  %7.1f GB/s algorithmic (24 B/px)" % (B, ms_w, B / ms_w * 1e3, 24.0 * px / ms_w / 1e6))
# host->device copy of the uint8 batch (pinned), the only PCIe leg of the deferred tail: 4 B/px instead of 24 B/px of floats
hi, hl = torch.from_numpy(img).pin_memory(), torch.from_numpy(lab).pin_memory()
ms_c = timeit(lambda: (hi.to(dev, non_blocking=True), hl.to(dev, non_blocking=True)))
print("H2D uint8 batch   B=%d           %7.3f ms  %8.1f samples/s  (%.1f MB)" % (B, ms_c, B / ms_c * 1e3, 4.0 * px / 1e6))

# the reference's way: one sample at a time on one host core (scipy.ndimage), bounded sample
n = 6
t0 = time.perf_counter()
for b in range(n):
    tr.ToTensor()(tr.Normalize_tf()({"image": img[b % B], "label": lab[b % B], "img_name": ""}))
t_ntf = (time.perf_counter() - t0) / n
from scipy import ndimage
rs = np.random.RandomState(0)
t0 = time.perf_counter()
for b in range(n):
    dx = ndimage.gaussian_filter(rs.rand(S, S) * 2 - 1, 0.08 * S, mode="constant", cval=0) * 2 * S
    dy = ndimage.gaussian_filter(rs.rand(S, S) * 2 - 1, 0.08 * S, mode="constant", cval=0) * 2 * S
    gx, gy = np.meshgrid(np.arange(S), np.arange(S), indexing="ij")
    idx = np.reshape(gx + dx, (-1, 1)), np.reshape(gy + dy, (-1, 1))
    for c in range(3):
        ndimage.map_coordinates(img[b % B][:, :, c], idx, order=1)
    ndimage.map_coordinates(lab[b % B], idx, order=1, mode="nearest")
t_el = (time.perf_counter() - t0) / n
print("CPU (1 core, scipy): Normalize_tf+ToTensor %.1f ms/sample = %.1f samples/s; elastic_transform %.1f ms/sample = %.1f samples/s"
      % (t_ntf * 1e3, 1 / t_ntf, t_el * 1e3, 1 / t_el))

# evaluation post-processing: device batch vs scipy per image on one core
from kernel_cases import _scipy_postprocess
rs = np.random.RandomState(5)
yy, xx = np.meshgrid(np.arange(S), np.arange(S), indexing="ij")
pb = np.stack([np.stack([np.clip(1.3 - np.sqrt(((yy - 256) / r) ** 2 + ((xx - 240) / (1.2 * r)) ** 2), 0, 1) for r in (90.0, 170.0)]) for _ in range(16)]).astype(np.float32)
pb = np.clip(pb + 0.5 * (rs.rand(*pb.shape) < 0.02), 0, 1).astype(np.float32)
pd = torch.from_numpy(pb).to(dev)
K.postprocess(pd, 0.75, 0.75)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    K.postprocess(pd, 0.75, 0.75)
torch.cuda.synchronize()
t_dev = (time.perf_counter() - t0) / 5
t0 = time.perf_counter()
for b in range(2):
    _scipy_postprocess(pb[b], 0.75, 0.75)
t_cpu = (time.perf_counter() - t0) / 2
print("postprocess       B=16 512x512    %7.3f ms per batch (incl. the convergence read-back) = %.0f images/s;  scipy on one core %.1f ms/image = %.1f images/s"
      % (t_dev * 1e3, 16 / t_dev, t_cpu * 1e3, 1 / t_cpu))
