"""NHWC activation descriptors shared by the engine and the kernel bindings.

Every activation of the generator lives in HBM as a 2-D ``[P, ld]`` fp32 matrix
(P = N*H*W pixels, channels fastest, ``ld`` a multiple of 4 floats so every pixel
row is 16-byte aligned).  Channel windows of one buffer replace the reference's
``torch.cat`` calls (aspp.py:72, decoder.py:51,53).

``Act`` additionally carries the *pending* per-channel transform that the
consumer kernel applies while loading (training-mode BN cannot be folded into the
producer because its statistics need the whole batch):

    u[p, c] = act(x[p, c] * scale[c] + shift[c]) * (mask[p, c] * mask_scale)

so a conv output is written once (raw) and read once (by its consumer).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import torch

ACT_NONE, ACT_RELU, ACT_RELU6 = 0, 1, 2


def round4(c: int) -> int:
    return (c + 3) // 4 * 4


@dataclass
class BNRec:
    """What BN backward needs about one normalisation (training mode)."""
    key: str                       # state-dict prefix, e.g. 'aspp.bn1'
    mean: torch.Tensor             # [C] batch mean
    invstd: torch.Tensor           # [C] 1/sqrt(var+eps)
    count: float                   # elements per channel the statistics were taken over
    q1_border: bool = False        # statistics include the zero border of quirk Q1
    gain: Optional[torch.Tensor] = None     # TransNorm: [C] factor 1 + alpha already folded into scale / shift
    frozen: bool = False           # eval-mode BN inside a training pass (freeze_bn): mean / invstd are the RUNNING statistics,
                                   # count = inf (no batch-statistics terms in the backward)


@dataclass
class Act:
    x: torch.Tensor                # [P, C] view, stride (ld, 1)
    N: int
    H: int
    W: int
    scale: Optional[torch.Tensor] = None
    shift: Optional[torch.Tensor] = None
    act: int = ACT_NONE
    mask: Optional[torch.Tensor] = None     # uint8 [P, C] view (1 = keep)
    mask_scale: float = 1.0
    bn: Optional[BNRec] = None
    meta: dict = field(default_factory=dict)
    # TransNorm (--use_TN): images [0, split) and [split, N) were normalised separately; scale / shift (and the
    # BNRec's mean / invstd, count = per-half counts) then hold one row per domain half, [2, C]
    split: int = 0

    @property
    def C(self) -> int:
        return self.x.shape[1]

    @property
    def P(self) -> int:
        return self.x.shape[0]

    @property
    def lazy(self) -> bool:
        return self.scale is not None or self.mask is not None or self.act != ACT_NONE

    def half(self, h: int) -> "Act":
        """The domain half h of a split activation as an ordinary one (contiguous row range, its own coefficients)."""
        assert self.split > 0
        n0, n1 = (0, self.split) if h == 0 else (self.split, self.N)
        ppi = self.P // self.N
        rows = slice(n0 * ppi, n1 * ppi)
        bn = None
        if self.bn is not None:
            bn = BNRec(self.bn.key, self.bn.mean[h], self.bn.invstd[h], self.bn.count[h], self.bn.q1_border, self.bn.gain)
        return Act(self.x[rows], n1 - n0, self.H, self.W, None if self.scale is None else self.scale[h],
                   None if self.shift is None else self.shift[h], self.act,
                   None if self.mask is None else self.mask[rows], self.mask_scale, bn, self.meta)

    def check(self):
        assert self.split == 0, "a domain-split activation reaches the kernels one half at a time (domain_split.DomainSplit)"
        assert self.x.dim() == 2 and self.x.stride(1) == 1, "activation must be a [P, C] row-major view"
        assert self.x.shape[0] == self.N * self.H * self.W
        assert self.x.stride(0) % 4 == 0 and self.x.stride(0) >= round4(self.C)
        assert self.x.data_ptr() % 16 == 0
        if self.mask is not None:
            assert self.mask.dtype == torch.uint8 and self.mask.shape == self.x.shape
            assert self.mask.stride(1) == 1 and self.mask.stride(0) % 4 == 0
        return self


def nchw_view(x2d: torch.Tensor, N: int, H: int, W: int) -> torch.Tensor:
    """Logical [N, C, H, W] view (channels-last strides) of a [P, C] NHWC matrix."""
    ld, C = x2d.stride(0), x2d.shape[1]
    return x2d.as_strided((N, C, H, W), (H * W * ld, 1, W * ld, ld), x2d.storage_offset())


K_CHUNK = 32      # IG_BK of the implicit-GEMM kernels


def tap_chunked(w_rtc: torch.Tensor) -> torch.Tensor:
    """[R, T, C] (rows, taps, channels) -> the K order of the multi-tap implicit-GEMM kernels, [R, nCC*T, 32]:
    k = (cc*T + t)*32 + c % 32 with cc = c // 32, channels zero-padded to a multiple of 32.  All T taps of one
    32-channel slice are consecutive K-chunks, so a workgroup re-reads its pixel strip while it is L2-resident.
    With fewer than 32 channels the order stays tap-major, [R, T, round4(C)] (nothing to keep resident, no padding)."""
    R, T, Cc = w_rtc.shape
    if round4(Cc) < K_CHUNK:                  # fewer than 32 channels: tap-major, padded to round4(C) only
        out = w_rtc.new_zeros(R, T, round4(Cc))
        out[:, :, :Cc] = w_rtc
        return out
    ncc = (Cc + K_CHUNK - 1) // K_CHUNK
    out = w_rtc.new_zeros(R, T, ncc * K_CHUNK)
    out[:, :, :Cc] = w_rtc
    return out.reshape(R, T, ncc, K_CHUNK).permute(0, 2, 1, 3).reshape(R, ncc * T, K_CHUNK).contiguous()


def conv_weight_shape(rows: int, ksize: int, channels: int):
    """Shape of the relayouted weight operand of ``conv``: [rows, 1, round4(C)] for 1x1, tap-chunked otherwise."""
    if ksize == 1 or round4(channels) < K_CHUNK:
        return (rows, ksize * ksize, round4(channels))
    return (rows, ((channels + K_CHUNK - 1) // K_CHUNK) * ksize * ksize, K_CHUNK)
