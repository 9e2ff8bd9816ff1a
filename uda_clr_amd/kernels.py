"""ctypes bindings of libuda_clr_hip.so (include/uda_clr_hip.h) at the torch-tensor level.

PyTorch is only the owner of device memory and of the current HIP stream here: every method
checks layouts on the host, passes raw pointers + sizes through the C ABI and launches on
``torch.cuda.current_stream()``.  There is no fallback: a missing library or a non-GPU tensor
raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from .acts import conv_weight_shape, Act, round4

STAT_SLOTS = 16      # UDA_STAT_SLOTS in include/uda_clr_hip.h
_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libuda_clr_hip.so")
_lib = None


class UdaSrc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("ldx", C.c_int64), ("N", C.c_int32), ("H", C.c_int32),
                ("W", C.c_int32), ("C", C.c_int32), ("scale", C.c_void_p), ("shift", C.c_void_p),
                ("act", C.c_int32), ("_pad", C.c_int32), ("mask", C.c_void_p), ("ldm", C.c_int64),
                ("mask_scale", C.c_float), ("_pad2", C.c_float)]


class UdaConvArgs(C.Structure):
    _fields_ = [("src", UdaSrc), ("w", C.c_void_p), ("Cout", C.c_int32), ("ksize", C.c_int32),
                ("dil", C.c_int32), ("origin", C.c_int32), ("bias", C.c_void_p), ("addend", C.c_void_p),
                ("ld_add", C.c_int64), ("y", C.c_void_p), ("ldy", C.c_int64), ("stats", C.c_void_p),
                ("mfma", C.c_int32), ("stride", C.c_int32), ("x3_src", C.c_void_p), ("x3_w", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_uint64)]


class UdaWgradArgs(C.Structure):
    _fields_ = [("src", UdaSrc), ("dy", C.c_void_p), ("lddy", C.c_int64), ("Cout", C.c_int32),
                ("ksize", C.c_int32), ("dil", C.c_int32), ("origin", C.c_int32), ("dw", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_uint64), ("mfma", C.c_int32), ("stride", C.c_int32),
                ("x3_src", C.c_void_p), ("x3_dy", C.c_void_p)]


# every symbol declared in include/uda_clr_hip.h: name -> (restype, argtypes)
_P, _I, _L, _F, _D, _U = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double, C.c_uint64
SYMBOLS = {
    "uda_last_error": (C.c_char_p, []),
    "uda_version": (_I, []),
    "uda_relayout_ohwi": (_I, [_P, _I, _I, _I, _P, _P]),
    "uda_relayout_dgrad": (_I, [_P, _I, _I, _I, _P, _P]),
    "uda_relayout_dw": (_I, [_P, _I, _P, _P]),
    "uda_conv_fwd": (_I, [C.POINTER(UdaConvArgs), _P]),
    "uda_conv_uses_x3": (_I, [C.POINTER(UdaConvArgs)]),
    "uda_conv_fwd_workspace_bytes": (_U, [C.POINTER(UdaConvArgs)]),
    "uda_x3_packed_bytes": (_U, [_L, _I]),
    "uda_x3_pack": (_I, [C.POINTER(UdaSrc), _P, _P]),
    "uda_conv_wgrad_workspace_bytes": (_U, [_L, _I, _I, _I]),
    "uda_conv_wgrad": (_I, [C.POINTER(UdaWgradArgs), _P]),
    "uda_conv_wgrad_uses_x3": (_I, [C.POINTER(UdaWgradArgs)]),
    "uda_dwconv_workspace_bytes": (_U, [_L, _I]),
    "uda_dwconv_fwd": (_I, [C.POINTER(UdaSrc), _P, _I, _I, _I, _P, _L, _P, _P]),
    "uda_dwconv_dgrad": (_I, [_P, _L, _P, _I, _I, _I, _I, _I, _I, _P, _L, _P]),
    "uda_dwconv_wgrad": (_I, [C.POINTER(UdaSrc), _P, _L, _I, _I, _I, _P, _P, _U, _P]),
    "uda_stem_workspace_bytes": (_U, [_L]),
    "uda_stem_fwd": (_I, [_P, _I, _I, _I, _P, _P, _L, _P, _P]),
    "uda_stem_wgrad": (_I, [_P, _I, _I, _I, _P, _L, _P, _P, _U, _P]),
    "uda_stem7_workspace_bytes": (_U, [_L]),
    "uda_stem7_fwd": (_I, [_P, _I, _I, _I, _P, _P, _L, _P, _P]),
    "uda_stem7_wgrad": (_I, [_P, _I, _I, _I, _P, _L, _P, _P, _U, _P]),
    "uda_maxpool_fwd": (_I, [C.POINTER(UdaSrc), _P, _L, _P, _L, _P]),
    "uda_maxpool_bwd": (_I, [_P, _L, _P, _L, _I, _I, _I, _I, _P, _L, _P]),
    "uda_rows_stride": (_I, [_P, _L, _I, _I, _I, _I, _I, _I, _P, _L, _P]),
    "uda_bn_add_relu": (_I, [C.POINTER(UdaSrc), C.POINTER(UdaSrc), _P, _L, _P]),
    "uda_relu_gate": (_I, [_P, _L, _P, _L, _L, _I, _P, _L, _P]),
    "uda_relayout_s2d": (_I, [_P, _I, _I, _I, _P, _P]),
    "uda_s2d_fwd": (_I, [_P, _L, _I, _I, _I, _I, _I, _I, _I, _F, _P, _L, _I, _I, _P]),
    "uda_s2d_bwd": (_I, [_P, _P, _L, _I, _I, _F, _I, _I, _I, _I, _I, _I, _P, _L, _I, _P]),
    "uda_s2d_bwd_gate": (_I, [_P, _L, _I, _I, _P, _L, _F, _I, _I, _I, _I, _I, _I, _P, _L, _P]),
    "uda_x3_pack_s2d_fwd": (_I, [_P, _L, _I, _I, _I, _I, _I, _I, _F, _I, _I, _P, _P]),
    "uda_x3_pack_s2d_bwd": (_I, [_P, _L, _I, _I, _P, _L, _F, _I, _I, _I, _I, _I, _I, _P, _P]),
    "uda_bn_finalize": (_I, [_P, _I, _D, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P]),
    "uda_bn_running_replay": (_I, [_P, _P, _I, _D, _I, _F, _F, _P, _P, _P]),
    "uda_bn_eval_coeffs": (_I, [_P, _P, _P, _P, _I, _F, _P, _P, _P]),
    "uda_tn_gain": (_I, [_P, _P, _I, _D, _D, _F, _P, _P, _P, _P, _P, _P]),
    "uda_tn_eval_coeffs": (_I, [_P, _P, _P, _P, _P, _P, _I, _F, _P, _P, _P]),
    "uda_bn_apply": (_I, [C.POINTER(UdaSrc), _P, _L, _P, _L, _P]),
    "uda_colstats": (_I, [_P, _L, _L, _I, _I, _P, _P]),
    "uda_colstats_window": (_I, [_P, _L, _L, _I, _I, _P, _I, _P]),
    "uda_upsample_fwd_stats": (_I, [_P, _L, _I, _I, _I, _I, _P, _L, _I, _I, _P, _I, _P]),
    "uda_mc_seg_head": (_I, [_P, _L, _I, _I, _I, _I, _P, _L, _I, _L, _P, _L, _I, _I, _P, _P, _I, _P, _L, _F, _P, _L, _P, _P, _L, _P]),
    "uda_bnbwd_reduce": (_I, [_P, _L, C.POINTER(UdaSrc), _P, _P, _P, _P]),
    "uda_bnbwd_finalize": (_I, [_P, _I, _D, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "uda_bnbwd_apply": (_I, [_P, _L, C.POINTER(UdaSrc), _P, _P, _P, _P, _P, _L, _P, _L, _P]),
    "uda_bnbwd_reduce_lowrank": (_I, [_P, _L, _I, _P, C.POINTER(UdaSrc), _P, _P, _P, _P]),
    "uda_bnbwd_apply_lowrank": (_I, [_P, _L, _I, _P, C.POINTER(UdaSrc), _P, _P, _P, _P, _P, _L, _P, _L, _P]),
    "uda_upsample_fwd": (_I, [_P, _L, _I, _I, _I, _I, _P, _L, _I, _I, _P]),
    "uda_upsample_bwd": (_I, [_P, _L, _I, _I, _I, _I, _P, _L, _I, _I, _P]),
    "uda_head_upsample_fwd": (_I, [_P, _L, _I, _I, _I, _I, _P, _I, _I, _P]),
    "uda_head_upsample_bwd": (_I, [_P, _I, _I, _I, _I, _P, _L, _I, _I, _I, _P]),
    "uda_gap_fwd": (_I, [_P, _L, _I, _I, _I, _F, _P, _L, _P]),
    "uda_broadcast_rows": (_I, [_P, _L, _I, _I, _I, _F, _P, _L, _P, _L, _P]),
    "uda_dropout_mask": (_I, [_P, _L, _L, _I, _F, _U, _U, _P]),
    "uda_seg_loss_fwd": (_I, [_P, _P, _L, _P, _P, _L, _P, _P, _P]),
    "uda_seg_loss_bwd": (_I, [_P, _P, _L, _P, _P, _L, _P, _P, _P, _P]),
    "uda_seg_counts": (_I, [_P, _P, _I, _I, _L, _F, _P, _P]),
    "uda_mc_stats": (_I, [_P, _I, _L, _P, _P, _P]),
    "uda_proto_weights": (_I, [_I, _I, _I, _I, _I, _I, _P, _P, _L, _P, _P, _P, _P, _P, _P]),
    "uda_proto_workspace_bytes": (_U, [_L, _I]),
    "uda_proto_reduce": (_I, [_P, _L, _L, _I, _P, _P, _P, _U, _P]),
    "uda_proto_finalize": (_I, [_P, _I, _P, _P]),
    "uda_proto_bwd": (_I, [_P, _L, _L, _I, _P, _P, _P, _P, _P, _L, _I, _P, _P]),
    "uda_feat_dot4": (_I, [_P, _L, _L, _I, _P, _P, _P]),
    "uda_feat_rank4": (_I, [_P, _P, _L, _I, _P, _L, _I, _P]),
    "uda_adam_step": (_I, [_P, _P, _P, _P, _L, _D, _D, _D, _D, _L, _P]),
    "uda_proto_align_fwd": (_I, [_P, _P, _P, _P, _F, _F, _I, _P, _P, _P, _P]),
    "uda_proto_align_bwd": (_I, [_P, _P, _P, _F, _F, _I, _P, _P, _P]),
    "uda_adv_loss_fwd": (_I, [_P, _I, _P, _I, _F, _F, _P, _P]),
    "uda_adv_loss_bwd": (_I, [_P, _I, _P, _I, _F, _F, _P, _P, _P, _P]),
    "uda_adv_s2d_fwd": (_I, [_P, _I, _I, _I, _I, _I, _P, _L, _I, _I, _P]),
    "uda_adv_s2d_bwd": (_I, [_P, _L, _I, _I, _P, _I, _I, _I, _I, _I, _P, _P]),
    "uda_upconv_fused_stats": (_I, [_I, _I, _I, _I, _I, _I]),
    "uda_upconv_fwd": (_I, [_P, _L, _I, _I, _I, _I, _I, _P, _L, _L, _P, _L, _I, _I, _P, _P]),
    "uda_upconv_bwd": (_I, [_P, _L, _I, _I, _I, _I, _I, _P, _L, _I, _I, _P]),
    "uda_postprocess_workspace_bytes": (_U, [_I, _I, _I]),
    "uda_postprocess": (_I, [_P, _I, _I, _I, _F, _F, _I, _P, _P, _P, _U, _P]),
    "uda_normalize_tf_workspace_bytes": (_U, [_I, _I, _I]),
    "uda_normalize_tf": (_I, [_P, _P, _I, _I, _I, C.POINTER(C.c_double), _I, _P, _P, _P, _P, _U, _P]),
    "uda_field_smooth": (_I, [_P, _I, _I, _I, _P, _I, _D, _P, _P, _P]),
    "uda_elastic_warp": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P]),
    "uda_photometric_u8": (_I, [_P, _I, _I, _I, _P, _P, _P, _I, _P, _P, _P]),
}


def load_library(path: Optional[str] = None):
    """dlopen the C-ABI library and bind every declared symbol (raises if one is missing)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or _LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError("libuda_clr_hip.so not found at %s - build it with "
                           "`python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)" % path)
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError -> symbol missing
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


class UdaError(RuntimeError):
    pass


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _mat(t: torch.Tensor, name="tensor"):
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError("%s must be a [P, C] row-major view" % name)
    return t.data_ptr(), t.stride(0)


class HipKernels:
    name = "hip"

    MFMA_F32, MFMA_BF16X3 = 0, 1

    def __init__(self, mfma=None):
        self.lib = load_library()
        # matrix instructions of the wide (MFMA-bound) conv tiles (UDA_CLR_MFMA, include/uda_clr_hip.h UDA_MFMA_*):
        #   "bf16x3" (default) = fp32 emulated on the bf16 pipe by exact 3-way operand splitting - fp32-level results (per-call
        #             error against float64 equal to the fp32 kernel's, tests/bench_x3.py), 1.5-1.7x the fp32-MFMA kernel's rate;
        #   "f32"    = v_mfma_f32_32x32x2_f32, the exact fp32 fma chain.
        mode = (mfma or os.environ.get("UDA_CLR_MFMA", "bf16x3")).lower()
        if mode not in ("f32", "bf16x3"):
            raise ValueError("UDA_CLR_MFMA must be 'f32' or 'bf16x3', got %r" % mode)
        self.mfma = self.MFMA_BF16X3 if mode == "bf16x3" else self.MFMA_F32

    # ------------------------------------------------------------------ plumbing
    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    def _ck(self, rc):
        if rc != 0:
            raise UdaError(self.lib.uda_last_error().decode())

    @staticmethod
    def _dev(t):
        if not t.is_cuda:
            raise UdaError("HIP kernels need device tensors (got %s); there is no CPU fallback" % t.device)
        if t.dtype not in (torch.float32, torch.float64, torch.uint8):
            raise UdaError("unsupported dtype %s" % t.dtype)

    def _src(self, a: Act) -> UdaSrc:
        a.check()
        self._dev(a.x)
        s = UdaSrc()
        s.x, s.ldx = a.x.data_ptr(), a.x.stride(0)
        s.N, s.H, s.W, s.C = a.N, a.H, a.W, a.C
        s.scale, s.shift = _ptr(a.scale), _ptr(a.shift)
        if a.scale is not None:
            assert a.scale.is_contiguous() and a.shift.is_contiguous() and a.scale.numel() == a.C
        s.act = a.act
        if a.mask is not None:
            s.mask, s.ldm = a.mask.data_ptr(), a.mask.stride(0)
        else:
            s.mask, s.ldm = None, 0
        s.mask_scale = a.mask_scale
        return s

    @staticmethod
    def _ws(like, nbytes):
        return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=like.device)

    # ------------------------------------------------------------------ weight layouts
    def relayout_ohwi(self, w):
        O, I, k, _ = w.shape
        out = torch.empty(conv_weight_shape(O, k, I), dtype=torch.float32, device=w.device)
        self._ck(self.lib.uda_relayout_ohwi(w.contiguous().data_ptr(), O, I, k, out.data_ptr(), self._stream()))
        return out

    def relayout_dgrad(self, w):
        O, I, k, _ = w.shape
        out = torch.empty(conv_weight_shape(I, k, O), dtype=torch.float32, device=w.device)
        self._ck(self.lib.uda_relayout_dgrad(w.contiguous().data_ptr(), O, I, k, out.data_ptr(), self._stream()))
        return out

    def relayout_dw(self, w):
        Cc = w.shape[0]
        out = torch.empty(9, Cc, dtype=torch.float32, device=w.device)
        self._ck(self.lib.uda_relayout_dw(w.contiguous().data_ptr(), Cc, out.data_ptr(), self._stream()))
        return out

    # ------------------------------------------------------------------ dense conv
    # The conv / weight-gradient kernels address their operands with 32-bit offsets: rows * row stride of one operand must stay
    # below 2^29 elements (fp32) resp. 2^27 sixteen-byte units of the packed bf16x3 operand.  Larger batches are run as groups
    # of whole images (a convolution never crosses an image; statistics are ADDED into, weight gradients summed).
    ELEM_LIMIT = 1 << 29

    def _image_groups(self, N, rows_per_image, widest_ld, x3_channels=0):
        """None when one launch can take the whole operand, else [(n0, n1)] image ranges that can."""
        lim_rows = self.ELEM_LIMIT // max(int(widest_ld), 1) - 256
        if x3_channels:
            lim_rows = min(lim_rows, (self.ELEM_LIMIT // 4) // (((x3_channels + 15) // 16) * 6) - 256)
        if N * rows_per_image <= lim_rows:
            return None
        g = lim_rows // rows_per_image
        if g < 1:
            raise UdaError("one image (%d pixels x %d floats per row) exceeds the 32-bit offsets of the conv kernels" % (rows_per_image, widest_ld))
        return [(n0, min(N, n0 + g)) for n0 in range(0, N, g)]

    @staticmethod
    def _act_rows(a: Act, n0, n1):
        ppi = a.P // a.N
        r = slice(n0 * ppi, n1 * ppi)
        return Act(a.x[r], n1 - n0, a.H, a.W, a.scale, a.shift, a.act, None if a.mask is None else a.mask[r], a.mask_scale, a.bn, a.meta)

    def conv(self, src: Act, w, ksize, dil, out, bias=None, addend=None, stats=None, origin=0, stride=1):
        """stride 2 (wide tiles only: the C side raises otherwise): ``out`` / ``addend`` / ``stats`` live on the strided grid"""
        groups = self._image_groups(src.N, src.P // src.N, max(src.x.stride(0), out.stride(0), 0 if addend is None else addend.stride(0),
                                                               0 if src.mask is None else src.mask.stride(0) // 4),
                                    src.C if self.mfma != self.MFMA_F32 else 0)
        Po = src.N * ((src.H - 1) // stride + 1) * ((src.W - 1) // stride + 1)
        if groups is not None:
            ppo = Po // src.N
            for n0, n1 in groups:
                r = slice(n0 * ppo, n1 * ppo)
                self.conv(self._act_rows(src, n0, n1), w, ksize, dil, out[r], bias, None if addend is None else addend[r], stats, origin, stride)
            return
        a = UdaConvArgs()
        a.origin = origin
        a.src = self._src(src)
        Cout = out.shape[1]
        assert w.is_contiguous() and tuple(w.shape) == conv_weight_shape(Cout, ksize, src.C), \
            "weight layout %s does not match conv %dx%d %d->%d" % (tuple(w.shape), ksize, ksize, src.C, Cout)
        assert out.shape[0] == Po and stride in (1, 2)
        a.w, a.Cout, a.ksize, a.dil, a.stride = w.data_ptr(), Cout, ksize, dil, stride
        a.bias = _ptr(bias)
        if bias is not None:
            assert bias.is_contiguous() and bias.numel() == Cout
        if addend is not None:
            assert addend.shape == out.shape
            a.addend, a.ld_add = _mat(addend, "addend")
        else:
            a.addend, a.ld_add = None, 0
        a.y, a.ldy = _mat(out, "out")
        if stats is not None:
            assert stats.dtype == torch.float64 and stats.is_contiguous() and tuple(stats.shape) == (STAT_SLOTS, 2, Cout)
        a.stats = _ptr(stats)
        a.mfma = self.mfma
        if self.mfma != self.MFMA_F32 and self.lib.uda_conv_uses_x3(C.byref(a)):
            # bf16x3: both operands as their three bf16 pieces.  The packed forms are kept on the descriptor / the relayouted
            # weight, so an activation read by several convolutions (the ASPP input) and a weight used by several passes of one
            # step are split once.
            xs = self._packed(src, a.src, out.device)
            xw = getattr(w, "_x3", None)
            if xw is None:
                xw = self.x3_pack_rows(w)
                w._x3 = xw
            a.x3_src, a.x3_w = xs.data_ptr(), xw.data_ptr()
            nws = int(self.lib.uda_conv_fwd_workspace_bytes(C.byref(a)))
            if nws:         # fp32 partial tiles of the K-split last round of tiles
                ws = self._ws(out, nws)
                a.workspace, a.workspace_bytes = ws.data_ptr(), nws
            return self._conv_x3(a)
        self._ck(self.lib.uda_conv_fwd(C.byref(a), self._stream()))

    def _conv_x3(self, a):
        """the bf16x3 GEMM launch alone (bench.py times this)"""
        self._ck(self.lib.uda_conv_fwd(C.byref(a), self._stream()))

    def x3_pack(self, usrc: UdaSrc, rows, K, device):
        out = torch.empty(int(self.lib.uda_x3_packed_bytes(rows, K)), dtype=torch.uint8, device=device)
        self._ck(self.lib.uda_x3_pack(C.byref(usrc), out.data_ptr(), self._stream()))
        return out

    def _packed(self, src: Act, usrc: UdaSrc, device):
        """The packed form of a conv operand, split once per step: it rides on the descriptor (a pending transform belongs to the
        descriptor) and, for a raw operand (a gradient matrix, a discriminator's space-to-depth image), on the tensor object, so
        the forward conv, the input-gradient conv and the weight gradient that read the same matrix share one packing pass.
        The engines never write into a matrix after it has been read as a conv operand (gradients are complete before the
        producer's backward reads them)."""
        xs = getattr(src, "_x3", None)
        raw = not src.lazy
        if xs is None and raw:
            xs = getattr(src.x, "_x3", None)
        if xs is not None and xs.numel() != int(self.lib.uda_x3_packed_bytes(src.P, src.C)):
            raise RuntimeError("bf16x3 packed operand of %d bytes attached to a [%d, %d] matrix (expected %d): stale attachment"
                               % (xs.numel(), src.P, src.C, int(self.lib.uda_x3_packed_bytes(src.P, src.C))))
        if xs is None:
            xs = self.x3_pack(usrc, src.P, src.C, device)
            if raw:
                try:
                    src.x._x3 = xs
                except AttributeError:
                    pass
        src._x3 = xs
        return xs

    def x3_pack_rows(self, w):
        """a relayouted weight [rows, taps, K'] (contiguous rows) as packed rows"""
        rows, klen = w.shape[0], w.numel() // w.shape[0]
        s = UdaSrc()
        s.x, s.ldx, s.N, s.H, s.W, s.C = w.data_ptr(), klen, 1, 1, rows, klen
        s.scale = s.shift = s.mask = None
        s.act, s.ldm, s.mask_scale = 0, 0, 1.0
        return self.x3_pack(s, rows, klen, w.device)

    def conv_wgrad(self, src: Act, dy, ksize, dil, dw, origin=0, stride=1):
        """stride 2 (wide tiles only): ``dy`` lives on the strided output grid of the forward conv"""
        groups = self._image_groups(src.N, src.P // src.N, max(src.x.stride(0), dy.stride(0), 0 if src.mask is None else src.mask.stride(0) // 4))
        Ho, Wo = (src.H - 1) // stride + 1, (src.W - 1) // stride + 1
        Po = src.N * Ho * Wo
        if groups is not None:
            ppo = Po // src.N
            tmp = torch.empty_like(dw)
            for i, (n0, n1) in enumerate(groups):
                self.conv_wgrad(self._act_rows(src, n0, n1), dy[n0 * ppo:n1 * ppo], ksize, dil, dw if i == 0 else tmp, origin, stride)
                if i:
                    dw.add_(tmp)
            return
        a = UdaWgradArgs()
        a.origin = origin
        a.src = self._src(src)
        Cout = dy.shape[1]
        assert dy.shape[0] == Po and stride in (1, 2) and dw.is_contiguous() and tuple(dw.shape) == (Cout, src.C, ksize, ksize)
        a.dy, a.lddy = _mat(dy, "dy")
        a.Cout, a.ksize, a.dil, a.stride = Cout, ksize, dil, stride
        a.dw = dw.data_ptr()
        ws = self._ws(dy, self.lib.uda_conv_wgrad_workspace_bytes(Po, Cout, src.C, ksize))
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
        a.mfma = self.mfma
        if self.mfma != self.MFMA_F32 and self.lib.uda_conv_wgrad_uses_x3(C.byref(a)):
            # bf16x3: the source's packed form usually exists already (the forward conv packed the same descriptor); dy is packed
            # once for its input-gradient conv and this weight gradient (the packed form rides on the tensor object)
            xs = self._packed(src, a.src, dy.device)
            dsrc = Act(dy, src.N, Ho, Wo)
            xd = self._packed(dsrc, self._src(dsrc), dy.device)
            a.x3_src, a.x3_dy = xs.data_ptr(), xd.data_ptr()
        self._ck(self.lib.uda_conv_wgrad(C.byref(a), self._stream()))

    # ------------------------------------------------------------------ depthwise
    def dwconv_fwd(self, src: Act, w9c, stride, dil, border_mode, out, stats=None):
        s = self._src(src)
        Ho, Wo = (src.H - 1) // stride + 1, (src.W - 1) // stride + 1
        assert out.shape == (src.N * Ho * Wo, src.C) and w9c.is_contiguous() and tuple(w9c.shape) == (9, src.C)
        y, ldy = _mat(out, "out")
        if stats is not None:
            assert stats.dtype == torch.float64 and stats.is_contiguous() and tuple(stats.shape) == (STAT_SLOTS, 2, src.C)
        self._ck(self.lib.uda_dwconv_fwd(C.byref(s), w9c.data_ptr(), stride, dil, border_mode, y, ldy, _ptr(stats),
                                         self._stream()))

    def dwconv_dgrad(self, dy, w9c, stride, dil, N, H, W, out):
        Cc = dy.shape[1]
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        assert dy.shape[0] == N * Ho * Wo and out.shape == (N * H * W, Cc)
        g, ldg = _mat(dy, "dy")
        o, ldo = _mat(out, "out")
        self._ck(self.lib.uda_dwconv_dgrad(g, ldg, w9c.data_ptr(), Cc, stride, dil, N, H, W, o, ldo, self._stream()))

    def dwconv_wgrad(self, src: Act, dy, stride, dil, border_mode, dw):
        s = self._src(src)
        Ho, Wo = (src.H - 1) // stride + 1, (src.W - 1) // stride + 1
        assert dy.shape == (src.N * Ho * Wo, src.C) and dw.is_contiguous() and dw.numel() == 9 * src.C
        g, ldg = _mat(dy, "dy")
        ws = self._ws(dy, self.lib.uda_dwconv_workspace_bytes(dy.shape[0], src.C))
        self._ck(self.lib.uda_dwconv_wgrad(C.byref(s), g, ldg, stride, dil, border_mode, dw.data_ptr(), ws.data_ptr(),
                                           ws.numel(), self._stream()))

    # ------------------------------------------------------------------ stem
    def stem_fwd(self, x, w, out, stats=None):
        self._dev(x)
        N, c3, H, W = x.shape
        assert c3 == 3 and x.is_contiguous() and w.is_contiguous() and tuple(w.shape) == (32, 3, 3, 3)
        Po = N * ((H - 1) // 2 + 1) * ((W - 1) // 2 + 1)
        assert out.shape == (Po, 32)
        y, ldy = _mat(out, "out")
        if stats is not None:
            assert stats.dtype == torch.float64 and stats.is_contiguous() and tuple(stats.shape) == (STAT_SLOTS, 2, 32)
        self._ck(self.lib.uda_stem_fwd(x.data_ptr(), N, H, W, w.data_ptr(), y, ldy, _ptr(stats), self._stream()))

    def stem_wgrad(self, x, dy, dw):
        N, _, H, W = x.shape
        Po = N * ((H - 1) // 2 + 1) * ((W - 1) // 2 + 1)
        assert dy.shape == (Po, 32) and dw.is_contiguous() and dw.numel() == 864
        g, ldg = _mat(dy, "dy")
        ws = self._ws(dy, self.lib.uda_stem_workspace_bytes(Po))
        self._ck(self.lib.uda_stem_wgrad(x.data_ptr(), N, H, W, g, ldg, dw.data_ptr(), ws.data_ptr(), ws.numel(),
                                         self._stream()))

    # ------------------------------------------------------------------ ResNet-101 pieces
    def stem7_fwd(self, x, w, out, stats=None):
        self._dev(x)
        N, c3, H, W = x.shape
        assert c3 == 3 and x.is_contiguous() and w.is_contiguous() and tuple(w.shape) == (64, 3, 7, 7)
        Po = N * ((H - 1) // 2 + 1) * ((W - 1) // 2 + 1)
        assert out.shape == (Po, 64)
        y, ldy = _mat(out, "out")
        if stats is not None:
            assert stats.dtype == torch.float64 and stats.is_contiguous() and tuple(stats.shape) == (STAT_SLOTS, 2, 64)
        self._ck(self.lib.uda_stem7_fwd(x.data_ptr(), N, H, W, w.data_ptr(), y, ldy, _ptr(stats), self._stream()))

    def stem7_wgrad(self, x, dy, dw):
        N, _, H, W = x.shape
        Po = N * ((H - 1) // 2 + 1) * ((W - 1) // 2 + 1)
        assert dy.shape == (Po, 64) and dw.is_contiguous() and dw.numel() == 64 * 147
        g, ldg = _mat(dy, "dy")
        ws = self._ws(dy, self.lib.uda_stem7_workspace_bytes(Po))
        self._ck(self.lib.uda_stem7_wgrad(x.data_ptr(), N, H, W, g, ldg, dw.data_ptr(), ws.data_ptr(), ws.numel(),
                                          self._stream()))

    def maxpool_fwd(self, src: Act, out, idx):
        s = self._src(src)
        Po = src.N * ((src.H - 1) // 2 + 1) * ((src.W - 1) // 2 + 1)
        assert out.shape == (Po, src.C) and idx.shape == (Po, src.C) and idx.dtype == torch.uint8
        o, ldo = _mat(out, "out")
        i, ldi = _mat(idx, "idx")
        self._ck(self.lib.uda_maxpool_fwd(C.byref(s), o, ldo, i, ldi, self._stream()))

    def maxpool_bwd(self, dz, idx, N, H, W, out):
        Cc = dz.shape[1]
        assert out.shape == (N * H * W, Cc) and idx.shape == dz.shape
        g, ldg = _mat(dz, "dz")
        i, ldi = _mat(idx, "idx")
        o, ldo = _mat(out, "out")
        self._ck(self.lib.uda_maxpool_bwd(g, ldg, i, ldi, N, H, W, Cc, o, ldo, self._stream()))

    def rows_stride(self, src, N, H, W, stride, out, scatter=False):
        """scatter=False: out[(n,oh,ow)] = src[(n,oh*s,ow*s)] (src is the H x W side); scatter=True: the
        transpose, out is the H x W side and receives zeros between the samples."""
        Cc = src.shape[1]
        Po = N * ((H - 1) // stride + 1) * ((W - 1) // stride + 1)
        big, small = (out, src) if scatter else (src, out)
        assert big.shape == (N * H * W, Cc) and small.shape == (Po, Cc)
        a, lda = _mat(src, "src")
        o, ldo = _mat(out, "out")
        self._ck(self.lib.uda_rows_stride(a, lda, N, H, W, Cc, stride, int(scatter), o, ldo, self._stream()))

    def bn_add_relu(self, a: Act, b: Act, out):
        sa, sb = self._src(a), self._src(b)
        assert out.shape == a.x.shape == b.x.shape
        o, ldo = _mat(out, "out")
        self._ck(self.lib.uda_bn_add_relu(C.byref(sa), C.byref(sb), o, ldo, self._stream()))

    def relu_gate(self, dz, z, out):
        assert dz.shape == z.shape == out.shape
        g, ldg = _mat(dz, "dz")
        v, ldv = _mat(z, "z")
        o, ldo = _mat(out, "out")
        self._ck(self.lib.uda_relu_gate(g, ldg, v, ldv, dz.shape[0], dz.shape[1], o, ldo, self._stream()))

    # ------------------------------------------------------------------ patch-discriminator geometry
    def relayout_s2d(self, w, dgrad):
        O, Cc = w.shape[0], w.shape[1]
        assert tuple(w.shape[2:]) == (4, 4)
        out = torch.empty(conv_weight_shape(4 * Cc, 2, O) if dgrad else conv_weight_shape(O, 2, 4 * Cc), dtype=torch.float32,
                          device=w.device)
        self._ck(self.lib.uda_relayout_s2d(w.detach().contiguous().data_ptr(), O, Cc, int(bool(dgrad)), out.data_ptr(), self._stream()))
        return out

    def s2d_fwd(self, src, nchw, N, Hs, Ws, Cc, vh, vw, slope, z):
        """src: NCHW tensor (nchw=True) or [N*Hs*Ws, C] rows; z: [N*Hz*Wz, 4C] rows with Hz = (vh+5)//2."""
        self._dev(src)
        Hz, Wz = (vh + 5) // 2, (vw + 5) // 2
        assert z.shape == (N * Hz * Wz, 4 * Cc)
        if nchw:
            assert src.is_contiguous() and tuple(src.shape) == (N, Cc, Hs, Ws)
            sp, lds_ = src.data_ptr(), 0
        else:
            assert src.shape == (N * Hs * Ws, Cc)
            sp, lds_ = _mat(src, "src")
        zp, ldz = _mat(z, "z")
        self._ck(self.lib.uda_s2d_fwd(sp, lds_, int(nchw), N, Hs, Ws, Cc, vh, vw, float(slope), zp, ldz, Hz, Wz, self._stream()))

    # ---- bf16x3: space-to-depth operands in packed form only (no fp32 image); the engine asks the routing predicates first
    def conv_route_x3(self, N, H, W, Cin, Cout, ksize):
        """True when ``conv`` of a RAW [N*H*W, Cin] operand to Cout outputs runs on the bf16x3 kernel in one launch."""
        if self.mfma != self.MFMA_BF16X3 or self._image_groups(N, H * W, max(round4(Cin), round4(Cout)), Cin) is not None:
            return False
        a = UdaConvArgs()
        a.src.N, a.src.H, a.src.W, a.src.C = N, H, W, Cin
        a.src.scale = a.src.shift = a.src.mask = None
        a.src.act, a.Cout, a.ksize, a.dil, a.mfma = 0, Cout, ksize, 1, self.mfma
        a.stats = None
        return bool(self.lib.uda_conv_uses_x3(C.byref(a)))

    def wgrad_route_x3(self, N, H, W, Cin, Cout, ksize):
        """True when ``conv_wgrad`` of a raw [N*H*W, Cin] source against a [N*H*W, Cout] gradient runs on the bf16x3 kernel."""
        if self.mfma != self.MFMA_BF16X3 or self._image_groups(N, H * W, max(round4(Cin), round4(Cout))) is not None:
            return False
        a = UdaWgradArgs()
        a.src.N, a.src.H, a.src.W, a.src.C = N, H, W, Cin
        a.src.scale = a.src.shift = a.src.mask = None
        a.src.act, a.Cout, a.ksize, a.dil, a.mfma = 0, Cout, ksize, 1, self.mfma
        return bool(self.lib.uda_conv_wgrad_uses_x3(C.byref(a)))

    def _packed_only(self, rows, Cc, device):
        """A [rows, Cc] fp32 matrix that exists ONLY in packed bf16x3 form: the returned tensor is an (uninitialised, never
        read) fp32 view over the head of the packed buffer, which rides on it as ``_x3`` like any cached packed operand -
        the x3 conv / weight-gradient kernels read nothing else.  Returns (matrix, packed bytes)."""
        xs = torch.empty(int(self.lib.uda_x3_packed_bytes(rows, Cc)), dtype=torch.uint8, device=device)
        m = xs[:rows * Cc * 4].view(torch.float32).view(rows, Cc)
        m._x3 = xs
        return m, xs

    def s2d_pack_fwd(self, src, N, Hs, Ws, Cc, vh, vw, slope):
        """uda_s2d_fwd + uda_x3_pack in one pass: src rows [N*Hs*Ws, Cc] -> the packed z image (a packed-only matrix
        [N*Hz*Wz, 4*Cc], see ``_packed_only``)."""
        self._dev(src)
        Hz, Wz = (vh + 5) // 2, (vw + 5) // 2
        assert src.shape == (N * Hs * Ws, Cc) and Cc % 8 == 0
        sp, lds_ = _mat(src, "src")
        z, xs = self._packed_only(N * Hz * Wz, 4 * Cc, src.device)
        self._ck(self.lib.uda_x3_pack_s2d_fwd(sp, lds_, N, Hs, Ws, Cc, vh, vw, float(slope), Hz, Wz, xs.data_ptr(), self._stream()))
        return z

    def s2d_pack_bwd(self, dz, gate, slope, N, Hs, Ws, Cc, vh, vw):
        """uda_s2d_bwd_gate + uda_x3_pack in one pass: the packed-only gradient matrix [N*Hs*Ws, Cc]."""
        Hz, Wz = (vh + 5) // 2, (vw + 5) // 2
        assert dz.shape == (N * Hz * Wz, 4 * Cc) and Cc % 8 == 0 and (gate is None or gate.shape == (N * Hs * Ws, Cc))
        gp, ldz = _mat(dz, "dz")
        tp, ldt = (None, 0) if gate is None else _mat(gate, "gate")
        d, xs = self._packed_only(N * Hs * Ws, Cc, dz.device)
        self._ck(self.lib.uda_x3_pack_s2d_bwd(gp, ldz, Hz, Wz, tp, ldt, float(slope), N, Hs, Ws, Cc, vh, vw, xs.data_ptr(), self._stream()))
        return d

    def s2d_bwd(self, dz, z_sign, slope, N, Hs, Ws, Cc, vh, vw, dst, nchw, gate=None):
        """``gate``: rows [N*Hs*Ws, Cc] of the forward's source whose sign is the LeakyReLU gate (instead of z_sign)."""
        Hz, Wz = (vh + 5) // 2, (vw + 5) // 2
        assert dz.shape == (N * Hz * Wz, 4 * Cc) and (z_sign is None or (z_sign.shape == dz.shape and z_sign.stride(0) == dz.stride(0)))
        gp, ldz = _mat(dz, "dz")
        if gate is not None:
            assert z_sign is None and not nchw and gate.shape == (N * Hs * Ws, Cc) and dst.shape == (N * Hs * Ws, Cc)
            tp, ldt = _mat(gate, "gate")
            dp, ldd = _mat(dst, "dst")
            self._ck(self.lib.uda_s2d_bwd_gate(gp, ldz, Hz, Wz, tp, ldt, float(slope), N, Hs, Ws, Cc, vh, vw, dp, ldd, self._stream()))
            return
        if nchw:
            assert dst.is_contiguous() and tuple(dst.shape) == (N, Cc, Hs, Ws)
            dp, ldd = dst.data_ptr(), 0
        else:
            assert dst.shape == (N * Hs * Ws, Cc)
            dp, ldd = _mat(dst, "dst")
        self._ck(self.lib.uda_s2d_bwd(gp, _ptr(z_sign), ldz, Hz, Wz, float(slope), N, Hs, Ws, Cc, vh, vw, dp, ldd, int(nchw),
                                      self._stream()))

    def adv_s2d_fwd(self, logits, pre_op, z):
        """z = s2d(pre(logits)), logits NCHW [N,C,H,W]; pre_op 1 = sigmoid, 2 = -sigmoid * log(sigmoid + 1e-7)."""
        self._dev(logits)
        N, Cc, H, W = logits.shape
        Hz, Wz = (H + 5) // 2, (W + 5) // 2
        assert logits.is_contiguous() and logits.dtype == torch.float32 and z.shape == (N * Hz * Wz, 4 * Cc)
        zp, ldz = _mat(z, "z")
        self._ck(self.lib.uda_adv_s2d_fwd(logits.data_ptr(), N, Cc, H, W, int(pre_op), zp, ldz, Hz, Wz, self._stream()))

    def adv_s2d_bwd(self, dz, logits, pre_op, d_logits):
        N, Cc, H, W = logits.shape
        Hz, Wz = (H + 5) // 2, (W + 5) // 2
        assert dz.shape == (N * Hz * Wz, 4 * Cc) and logits.is_contiguous() and d_logits.is_contiguous()
        assert tuple(d_logits.shape) == tuple(logits.shape)
        gp, ldz = _mat(dz, "dz")
        self._ck(self.lib.uda_adv_s2d_bwd(gp, ldz, Hz, Wz, logits.data_ptr(), N, Cc, H, W, int(pre_op), d_logits.data_ptr(),
                                          self._stream()))

    # ------------------------------------------------------------------ batch norm
    def bn_finalize(self, stats, count, gamma, beta, rmean, rvar, momentum, eps, scale, shift, mean, invstd):
        Cc = gamma.numel()
        for t in (gamma, beta, rmean, rvar, scale, shift, mean, invstd):
            assert t.is_contiguous() and t.numel() == Cc
        assert stats.dtype == torch.float64 and stats.is_contiguous() and tuple(stats.shape) == (STAT_SLOTS, 2, Cc)
        self._ck(self.lib.uda_bn_finalize(stats.data_ptr(), Cc, float(count), gamma.data_ptr(), beta.data_ptr(),
                                          rmean.data_ptr(), rvar.data_ptr(), momentum, eps, scale.data_ptr(),
                                          shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(), self._stream()))

    def bn_running_replay(self, mean, invstd, count, k, momentum, eps, rmean, rvar):
        Cc = mean.numel()
        for t in (mean, invstd, rmean, rvar):
            assert t.is_contiguous() and t.numel() == Cc
        self._ck(self.lib.uda_bn_running_replay(mean.data_ptr(), invstd.data_ptr(), Cc, float(count), int(k), momentum, eps,
                                                rmean.data_ptr(), rvar.data_ptr(), self._stream()))

    def bn_eval_coeffs(self, gamma, beta, rmean, rvar, eps, scale, shift):
        Cc = gamma.numel()
        for t in (gamma, beta, rmean, rvar, scale, shift):
            assert t.is_contiguous() and t.numel() == Cc
        self._ck(self.lib.uda_bn_eval_coeffs(gamma.data_ptr(), beta.data_ptr(), rmean.data_ptr(), rvar.data_ptr(), Cc,
                                             eps, scale.data_ptr(), shift.data_ptr(), self._stream()))

    def tn_gain(self, stats0, stats1, count0, count1, eps, scale0, shift0, scale1, shift1, gain):
        Cc = gain.numel()
        for t in (scale0, shift0, scale1, shift1, gain):
            assert t.is_contiguous() and t.numel() == Cc
        for st in (stats0, stats1):
            assert st.dtype == torch.float64 and st.is_contiguous() and tuple(st.shape) == (STAT_SLOTS, 2, Cc)
        self._ck(self.lib.uda_tn_gain(stats0.data_ptr(), stats1.data_ptr(), Cc, float(count0), float(count1), eps,
                                      scale0.data_ptr(), shift0.data_ptr(), scale1.data_ptr(), shift1.data_ptr(),
                                      gain.data_ptr(), self._stream()))

    def tn_eval_coeffs(self, gamma, beta, rmean_s, rvar_s, rmean_t, rvar_t, eps, scale, shift):
        Cc = gamma.numel()
        for t in (gamma, beta, rmean_s, rvar_s, rmean_t, rvar_t, scale, shift):
            assert t.is_contiguous() and t.numel() == Cc
        self._ck(self.lib.uda_tn_eval_coeffs(gamma.data_ptr(), beta.data_ptr(), rmean_s.data_ptr(), rvar_s.data_ptr(),
                                             rmean_t.data_ptr(), rvar_t.data_ptr(), Cc, eps, scale.data_ptr(),
                                             shift.data_ptr(), self._stream()))

    def bn_apply(self, src: Act, out, residual=None):
        s = self._src(src)
        assert out.shape == src.x.shape and src.C % 4 == 0
        o, ldo = _mat(out, "out")
        r, ldr = (None, 0) if residual is None else _mat(residual, "residual")
        self._ck(self.lib.uda_bn_apply(C.byref(s), r, ldr, o, ldo, self._stream()))

    def colstats(self, x, stats):
        self._dev(x)
        p, ld = _mat(x, "x")
        slots, nq, Cc = stats.shape
        assert slots == STAT_SLOTS and Cc == x.shape[1] and stats.dtype == torch.float64 and stats.is_contiguous()
        self._ck(self.lib.uda_colstats(p, ld, x.shape[0], Cc, nq, stats.data_ptr(), self._stream()))

    def colstats_window(self, x, stats, c_off):
        """(sum, sum of squares) of x's channels ADDED into channels [c_off, c_off + C) of the wider accumulator ``stats``."""
        self._dev(x)
        p, ld = _mat(x, "x")
        slots, nq, Cs = stats.shape
        Cc = x.shape[1]
        assert slots == STAT_SLOTS and c_off >= 0 and c_off + Cc <= Cs and stats.dtype == torch.float64 and stats.is_contiguous()
        self._ck(self.lib.uda_colstats_window(p, ld, x.shape[0], Cc, nq, stats.data_ptr() + 8 * c_off, Cs, self._stream()))

    def colsum(self, x, out):
        st = torch.zeros(STAT_SLOTS, 1, x.shape[1], dtype=torch.float64, device=x.device)
        self.colstats(x, st)
        out.copy_(st.sum(0)[0])

    @staticmethod
    def _lowrank(lowrank, y):
        """(d [P, k] rows, w [k, C] contiguous) of dU = d @ w, k = 1 or 2"""
        d, w = lowrank
        k = d.shape[1]
        assert k in (1, 2) and d.shape[0] == y.P and tuple(w.shape) == (k, y.C) and w.is_contiguous() and d.stride(1) == 1
        assert d.dtype == w.dtype == torch.float32
        return d.data_ptr(), d.stride(0), k, w.data_ptr()

    def bnbwd_reduce(self, dU, y: Act, sums, lowrank=None):
        s = self._src(y)
        assert sums.dtype == torch.float64 and tuple(sums.shape) == (STAT_SLOTS, 3, y.C) and sums.is_contiguous()
        if lowrank is not None:
            dp, ldd, k, wp = self._lowrank(lowrank, y)
            self._ck(self.lib.uda_bnbwd_reduce_lowrank(dp, ldd, k, wp, C.byref(s), y.bn.mean.data_ptr(), y.bn.invstd.data_ptr(),
                                                       sums.data_ptr(), self._stream()))
            return
        assert dU.shape == y.x.shape
        d, ldu = _mat(dU, "dU")
        self._ck(self.lib.uda_bnbwd_reduce(d, ldu, C.byref(s), y.bn.mean.data_ptr(), y.bn.invstd.data_ptr(),
                                           sums.data_ptr(), self._stream()))

    def bnbwd_finalize(self, sums, y: Act, c1, c2, dgamma, dbeta, q1_total=None):
        if q1_total is not None:
            assert q1_total.is_contiguous() and q1_total.numel() == y.C and q1_total.dtype == torch.float32
        self._ck(self.lib.uda_bnbwd_finalize(sums.data_ptr(), y.C, float(y.bn.count), int(y.bn.q1_border), y.act,
                                             y.shift.data_ptr(), y.bn.mean.data_ptr(), y.bn.invstd.data_ptr(), _ptr(q1_total),
                                             c1.data_ptr(), c2.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                             self._stream()))

    def bnbwd_apply(self, dU, y: Act, c1, c2, out, addend=None, lowrank=None):
        s = self._src(y)
        assert y.x.shape == out.shape
        o, ldo = _mat(out, "out")
        ad, lda = (None, 0) if addend is None else _mat(addend, "addend")
        if lowrank is not None:
            dp, ldd, k, wp = self._lowrank(lowrank, y)
            self._ck(self.lib.uda_bnbwd_apply_lowrank(dp, ldd, k, wp, C.byref(s), y.bn.mean.data_ptr(), y.bn.invstd.data_ptr(),
                                                      c1.data_ptr(), c2.data_ptr(), ad, lda, o, ldo, self._stream()))
            return
        assert dU.shape == y.x.shape
        d, ldu = _mat(dU, "dU")
        self._ck(self.lib.uda_bnbwd_apply(d, ldu, C.byref(s), y.bn.mean.data_ptr(), y.bn.invstd.data_ptr(),
                                          c1.data_ptr(), c2.data_ptr(), ad, lda, o, ldo, self._stream()))

    # ------------------------------------------------------------------ resampling / pooling
    def upsample_fwd(self, x, N, h, w, out, H, W, stats=None):
        """``stats`` (fp64 [SLOTS, 2, Cs >= C]): also ADD the output's per-channel (sum, sum of squares) into its channels [0, C)."""
        self._dev(x)
        p, ld = _mat(x, "x")
        o, ldo = _mat(out, "out")
        assert x.shape[0] == N * h * w and out.shape == (N * H * W, x.shape[1])
        if stats is not None:
            assert stats.dtype == torch.float64 and stats.is_contiguous() and stats.shape[:2] == (STAT_SLOTS, 2) and stats.shape[2] >= x.shape[1]
            self._ck(self.lib.uda_upsample_fwd_stats(p, ld, N, h, w, x.shape[1], o, ldo, H, W, stats.data_ptr(), stats.shape[2], self._stream()))
            return
        self._ck(self.lib.uda_upsample_fwd(p, ld, N, h, w, x.shape[1], o, ldo, H, W, self._stream()))

    def upsample_stats(self, x, N, h, w, H, W, stats):
        """per-channel (sum, sum of squares) of the bilinearly upsampled tensor ADDED into channels [0, C) of ``stats``; the
        tensor itself is not written"""
        self._dev(x)
        p, ld = _mat(x, "x")
        assert x.shape[0] == N * h * w and stats.dtype == torch.float64 and stats.is_contiguous()
        assert stats.shape[:2] == (STAT_SLOTS, 2) and stats.shape[2] >= x.shape[1]
        self._ck(self.lib.uda_upsample_fwd_stats(p, ld, N, h, w, x.shape[1], None, 0, H, W, stats.data_ptr(), stats.shape[2], self._stream()))

    def mc_seg_head(self, feature, N, h, w, low, bnd, H, W, scale, shift, act, mask, mask_scale, wgt, bias, out):
        """decoder.last_conv on the virtual x_feature = cat(up(feature), low rows (shared by the repeated batch), boundary):
        ``uda_mc_seg_head``.  feature [N*h*w, Cf], low [P_low, Cl] with P_low | N*H*W, bnd [N*H*W, 1], wgt = relayout_ohwi of the
        [2, C, 1, 1] weight, out [N*H*W, 2]."""
        self._dev(feature)
        fp, ldf = _mat(feature, "feature")
        lp, ldl = _mat(low, "low")
        P = N * H * W
        Cf, Cl = feature.shape[1], low.shape[1]
        Cc = Cf + Cl + 1
        assert feature.shape[0] == N * h * w and P % low.shape[0] == 0 and bnd.shape == (P, 1) and out.shape == (P, 2)
        assert bnd.stride(1) == 1 and out.stride(1) == 1 and wgt.is_contiguous() and wgt.shape[0] == 2 and wgt.numel() // 2 >= Cc
        assert scale.is_contiguous() and shift.is_contiguous() and scale.numel() == Cc and bias.numel() == 2
        mp, ldm = (None, 0)
        if mask is not None:
            assert mask.dtype == torch.uint8 and mask.shape == (P, Cc) and mask.stride(1) == 1
            mp, ldm = mask.data_ptr(), mask.stride(0)
        self._ck(self.lib.uda_mc_seg_head(fp, ldf, N, h, w, Cf, lp, ldl, Cl, low.shape[0], bnd.data_ptr(), bnd.stride(0), H, W,
                                          scale.data_ptr(), shift.data_ptr(), int(act), mp, ldm, float(mask_scale), wgt.data_ptr(),
                                          wgt.numel() // 2, bias.data_ptr(), out.data_ptr(), out.stride(0), self._stream()))

    def upsample_bwd(self, dout, N, H, W, dx, h, w):
        p, ld = _mat(dout, "dout")
        o, ldo = _mat(dx, "dx")
        assert dout.shape[0] == N * H * W and dx.shape == (N * h * w, dout.shape[1])
        self._ck(self.lib.uda_upsample_bwd(p, ld, N, H, W, dout.shape[1], o, ldo, h, w, self._stream()))

    def upconv_fwd(self, g, N, h, w, out, H, W, addend=None, dil=1, stats=None):
        """out[p] = addend[p % addend.rows] + sum over the 9 taps of the bilinear (align_corners) read of g's tap plane at the tap
        position; g: [N*h*w, 9*C] (tap-major columns), out: [N*H*W, C]; stats ([SLOTS, 2, C] fp64, added into): column sums
        and sums of squares of out (fused into the kernel where the geometry allows, a colstats pass otherwise)."""
        self._dev(g)
        Cc = out.shape[1]
        assert g.shape == (N * h * w, 9 * Cc) and out.shape[0] == N * H * W
        gp, ldg = _mat(g, "g")
        o, ldo = _mat(out, "out")
        ad, lda, rows = (None, 0, 1) if addend is None else (_mat(addend, "addend") + (addend.shape[0],))
        if addend is not None:
            assert addend.shape[1] == Cc and (N * H * W) % addend.shape[0] == 0
        fused = stats is not None and bool(self.lib.uda_upconv_fused_stats(h, w, H, W, Cc, dil))
        if stats is not None:
            assert stats.dtype == torch.float64 and stats.is_contiguous() and tuple(stats.shape) == (STAT_SLOTS, 2, Cc)
        self._ck(self.lib.uda_upconv_fwd(gp, ldg, N, h, w, Cc, dil, ad, lda, rows, o, ldo, H, W, _ptr(stats) if fused else None,
                                         self._stream()))
        if stats is not None and not fused:
            self.colstats(out, stats)

    def upconv_bwd(self, dy, N, H, W, dg, h, w, dil=1):
        self._dev(dy)
        Cc = dy.shape[1]
        assert dy.shape[0] == N * H * W and dg.shape == (N * h * w, 9 * Cc)
        d, ldy = _mat(dy, "dy")
        gp, ldg = _mat(dg, "dg")
        self._ck(self.lib.uda_upconv_bwd(d, ldy, N, H, W, Cc, dil, gp, ldg, h, w, self._stream()))

    def head_upsample_fwd(self, x, N, h, w, out):
        self._dev(x)
        p, ld = _mat(x, "x")
        Cc = x.shape[1]
        assert out.is_contiguous() and out.shape[0] == N and out.shape[1] == Cc and x.shape[0] == N * h * w
        self._ck(self.lib.uda_head_upsample_fwd(p, ld, N, h, w, Cc, out.data_ptr(), out.shape[2], out.shape[3],
                                                self._stream()))

    def head_upsample_bwd(self, dout, dx, N, h, w, accumulate=False):
        self._dev(dout)
        p, ld = _mat(dx, "dx")
        assert dout.is_contiguous() and dout.shape[0] == N and dout.shape[1] == dx.shape[1] and dx.shape[0] == N * h * w
        self._ck(self.lib.uda_head_upsample_bwd(dout.data_ptr(), N, dout.shape[1], dout.shape[2], dout.shape[3], p, ld,
                                                h, w, int(accumulate), self._stream()))

    def gap_fwd(self, x, N, out, scale):
        self._dev(x)
        p, ld = _mat(x, "x")
        o, ldo = _mat(out, "out")
        assert x.shape[0] % N == 0 and out.shape == (N, x.shape[1])
        self._ck(self.lib.uda_gap_fwd(p, ld, N, x.shape[0] // N, x.shape[1], scale, o, ldo, self._stream()))

    def broadcast_rows(self, g, N, out, scale, addend=None):
        self._dev(g)
        p, ld = _mat(g, "g")
        o, ldo = _mat(out, "out")
        assert g.shape == (N, out.shape[1]) and out.shape[0] % N == 0
        ad, lda = (None, 0) if addend is None else _mat(addend, "addend")
        self._ck(self.lib.uda_broadcast_rows(p, ld, N, out.shape[0] // N, out.shape[1], scale, ad, lda, o, ldo,
                                             self._stream()))

    def dropout_mask(self, mask, p, seed, offset):
        self._dev(mask)
        assert mask.dtype == torch.uint8 and mask.stride(1) == 1
        self._ck(self.lib.uda_dropout_mask(mask.data_ptr(), mask.stride(0), mask.shape[0], mask.shape[1], p,
                                           int(seed), int(offset), self._stream()))

    # ------------------------------------------------------------------ losses / metrics
    def seg_loss_fwd(self, o, tmap, b, tbd):
        """-> device float[3] = (BCE(sigmoid(o), map) + MSE(sigmoid(b), boundary), bce, mse)"""
        for t in (o, tmap, b, tbd):
            self._dev(t)
            assert t.is_contiguous()
        assert o.shape == tmap.shape and b.shape == tbd.shape
        loss = torch.empty(3, dtype=torch.float32, device=o.device)
        ws = torch.empty(2, dtype=torch.float64, device=o.device)
        self._ck(self.lib.uda_seg_loss_fwd(o.data_ptr(), tmap.data_ptr(), o.numel(), b.data_ptr(), tbd.data_ptr(),
                                           b.numel(), loss.data_ptr(), ws.data_ptr(), self._stream()))
        return loss

    def seg_loss_bwd(self, o, tmap, b, tbd, gscale):
        d_o, d_b = torch.empty_like(o), torch.empty_like(b)
        assert gscale.numel() == 1 and gscale.dtype == torch.float32
        self._ck(self.lib.uda_seg_loss_bwd(o.data_ptr(), tmap.data_ptr(), o.numel(), b.data_ptr(), tbd.data_ptr(),
                                           b.numel(), gscale.data_ptr(), d_o.data_ptr(), d_b.data_ptr(), self._stream()))
        return d_o, d_b

    def seg_counts(self, logits, target, thr):
        """-> int64 [C, 3] = (intersection, predicted, ground truth) for sigmoid(logits) > thr"""
        self._dev(logits)
        assert logits.is_contiguous() and target.is_contiguous() and logits.shape == target.shape
        B, Cc, H, W = logits.shape
        counts = torch.empty(Cc, 3, dtype=torch.int64, device=logits.device)
        self._ck(self.lib.uda_seg_counts(logits.data_ptr(), target.data_ptr(), B, Cc, H * W, thr, counts.data_ptr(),
                                         self._stream()))
        return counts

    # ------------------------------------------------------------------ prototypes
    def mc_stats(self, preds, T):
        self._dev(preds)
        assert preds.is_contiguous() and preds.shape[0] % T == 0
        shape = (preds.shape[0] // T,) + tuple(preds.shape[1:])
        std = torch.empty(shape, dtype=torch.float32, device=preds.device)
        mean = torch.empty_like(std)
        self._ck(self.lib.uda_mc_stats(preds.data_ptr(), T, std.numel(), std.data_ptr(), mean.data_ptr(), self._stream()))
        return std, mean

    def proto_weights(self, mode, B, h, w, map_=None, logits=None, std_map=None, mean_map=None):
        dev = (map_ if map_ is not None else logits).device
        P = B * h * w
        wts = torch.empty(P, 4, dtype=torch.float32, device=dev)
        H = W = 0
        m0 = m1 = None
        if mode == 0:
            assert map_.is_contiguous() and map_.shape[:2] == (B, 2)
            H, W = map_.shape[2], map_.shape[3]
        if mode in (1, 2):
            assert logits.shape == (P, 2) and logits.stride(1) == 1
        if mode == 2:
            assert std_map.is_contiguous() and mean_map.is_contiguous() and std_map.shape == mean_map.shape
            H, W = std_map.shape[2], std_map.shape[3]
            m0 = torch.empty(P, dtype=torch.float32, device=dev)
            m1 = torch.empty(P, dtype=torch.float32, device=dev)
        self._ck(self.lib.uda_proto_weights(mode, B, h, w, H, W, _ptr(map_), _ptr(logits),
                                            0 if logits is None else logits.stride(0), _ptr(std_map), _ptr(mean_map),
                                            wts.data_ptr(), _ptr(m0), _ptr(m1), self._stream()))
        return wts, m0, m1

    def proto_reduce(self, feat, wts, sums):
        self._dev(feat)
        f, ldf = _mat(feat, "feat")
        P, Cc = feat.shape
        assert wts.shape == (P, 4) and wts.is_contiguous() and sums.dtype == torch.float64 and tuple(sums.shape) == (4, Cc + 1)
        ws = self._ws(feat, self.lib.uda_proto_workspace_bytes(P, Cc))
        self._ck(self.lib.uda_proto_reduce(f, ldf, P, Cc, wts.data_ptr(), sums.data_ptr(), ws.data_ptr(), ws.numel(),
                                           self._stream()))

    def proto_finalize(self, sums):
        Cc = sums.shape[1] - 1
        cent = torch.empty(4, Cc, dtype=torch.float32, device=sums.device)
        self._ck(self.lib.uda_proto_finalize(sums.data_ptr(), Cc, cent.data_ptr(), self._stream()))
        return cent

    def proto_bwd(self, feat, wts, sums, dC, d_feat=None, accumulate=False, want_dw=False):
        f, ldf = _mat(feat, "feat")
        P, Cc = feat.shape
        assert dC.is_contiguous() and tuple(dC.shape) == (4, Cc) and dC.dtype == torch.float32
        coef = torch.empty(4, Cc + 1, dtype=torch.float32, device=feat.device)
        d_w = torch.empty(P, 4, dtype=torch.float32, device=feat.device) if want_dw else None
        dptr, ldd = (None, 0) if d_feat is None else _mat(d_feat, "d_feat")
        self._ck(self.lib.uda_proto_bwd(f, ldf, P, Cc, wts.data_ptr(), sums.data_ptr(), dC.data_ptr(), coef.data_ptr(),
                                        dptr, ldd, int(accumulate), _ptr(d_w), self._stream()))
        return d_w

    def proto_align_fwd(self, cur_src, cur_tgt, prev_src, prev_tgt, decay):
        """-> (new_src [4,C], new_tgt [4,C], losses float[2] = (intra, inter)); prev_* None on first use."""
        self._dev(cur_src)
        Cc = cur_src.shape[1]
        for t in (cur_src, cur_tgt, prev_src, prev_tgt):
            assert t is None or (t.is_contiguous() and tuple(t.shape) == (4, Cc) and t.dtype == torch.float32)
        new_src, new_tgt = torch.empty_like(cur_src), torch.empty_like(cur_tgt)
        losses = torch.empty(2, dtype=torch.float32, device=cur_src.device)
        self._ck(self.lib.uda_proto_align_fwd(cur_src.data_ptr(), cur_tgt.data_ptr(), _ptr(prev_src), _ptr(prev_tgt),
                                              float(1 - decay), float(decay), Cc, new_src.data_ptr(), new_tgt.data_ptr(),
                                              losses.data_ptr(), self._stream()))
        return new_src, new_tgt, losses

    def proto_align_bwd(self, new_src, new_tgt, g, w_src, w_tgt):
        Cc = new_src.shape[1]
        assert g.numel() == 1 and g.dtype == torch.float32 and new_src.is_contiguous() and new_tgt.is_contiguous()
        d_src, d_tgt = torch.empty_like(new_src), torch.empty_like(new_tgt)
        self._ck(self.lib.uda_proto_align_bwd(new_src.data_ptr(), new_tgt.data_ptr(), g.data_ptr(), float(w_src), float(w_tgt), Cc,
                                              d_src.data_ptr(), d_tgt.data_ptr(), self._stream()))
        return d_src, d_tgt

    def adv_loss_fwd(self, d1, d2, label, scale):
        self._dev(d1)
        assert d1.is_contiguous() and d2.is_contiguous() and d1.dtype == d2.dtype == torch.float32
        loss = torch.empty(1, dtype=torch.float32, device=d1.device)
        self._ck(self.lib.uda_adv_loss_fwd(d1.data_ptr(), d1.numel(), d2.data_ptr(), d2.numel(), float(label), float(scale),
                                           loss.data_ptr(), self._stream()))
        return loss

    def adv_loss_bwd(self, d1, d2, label, scale, g):
        assert g.numel() == 1 and g.dtype == torch.float32
        g1, g2 = torch.empty_like(d1), torch.empty_like(d2)
        self._ck(self.lib.uda_adv_loss_bwd(d1.data_ptr(), d1.numel(), d2.data_ptr(), d2.numel(), float(label), float(scale),
                                           g.data_ptr(), g1.data_ptr(), g2.data_ptr(), self._stream()))
        return g1, g2

    def feat_dot4(self, feat, coef):
        """[P,4]: feat @ coef[:, :C].T + coef[:, C]"""
        f, ldf = _mat(feat, "feat")
        P, Cc = feat.shape
        assert coef.is_contiguous() and tuple(coef.shape) == (4, Cc + 1) and coef.dtype == torch.float32
        out = torch.empty(P, 4, dtype=torch.float32, device=feat.device)
        self._ck(self.lib.uda_feat_dot4(f, ldf, P, Cc, coef.data_ptr(), out.data_ptr(), self._stream()))
        return out

    def feat_rank4(self, wts, coef, d_feat, accumulate=False):
        """d_feat (+)= wts @ coef[:, :C]"""
        d, ldd = _mat(d_feat, "d_feat")
        P, Cc = d_feat.shape
        assert wts.is_contiguous() and tuple(wts.shape) == (P, 4) and tuple(coef.shape) == (4, Cc + 1) and coef.is_contiguous()
        self._ck(self.lib.uda_feat_rank4(wts.data_ptr(), coef.data_ptr(), P, Cc, d, ldd, int(accumulate), self._stream()))

    # ------------------------------------------------------------------ optimiser
    # ------------------------------------------------------------------ input pipeline tail
    @staticmethod
    def gaussian_weights(sigma, radius):
        """The taps of scipy.ndimage.gaussian_filter1d, computed the way scipy computes them (numpy, float64), centre first."""
        import numpy as np
        x = np.arange(-radius, radius + 1)
        phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
        phi = phi / phi.sum()
        return phi[radius:].copy()

    def normalize_tf(self, image_u8, label_u8, sigma=3.0):
        """uint8 [B,H,W,3] + uint8 grey mask [B,H,W] -> (image f32 [B,3,H,W], map f32 [B,2,H,W], boundary f32 [B,1,H,W])."""
        for t in (image_u8, label_u8):
            self._dev(t)
            assert t.dtype == torch.uint8 and t.is_contiguous()
        B, H, W, ch = image_u8.shape
        assert ch == 3 and tuple(label_u8.shape) == (B, H, W)
        radius = int(4.0 * sigma + 0.5)
        wts = self.gaussian_weights(sigma, radius)
        dev = image_u8.device
        image = torch.empty(B, 3, H, W, dtype=torch.float32, device=dev)
        mp = torch.empty(B, 2, H, W, dtype=torch.float32, device=dev)
        bd = torch.empty(B, 1, H, W, dtype=torch.float32, device=dev)
        ws = self._ws(image_u8, self.lib.uda_normalize_tf_workspace_bytes(B, H, W))
        self._ck(self.lib.uda_normalize_tf(image_u8.data_ptr(), label_u8.data_ptr(), B, H, W,
                                           wts.ctypes.data_as(C.POINTER(C.c_double)), radius, image.data_ptr(), mp.data_ptr(),
                                           bd.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()))
        return image, mp, bd

    def field_smooth(self, noise, sigma, alpha):
        """alpha * scipy.ndimage.gaussian_filter(noise, sigma, mode='constant') per [H,W] plane of a float64 [..., H, W] tensor,
        bit-identical to scipy (same taps, same summation order, no fused multiply-add)."""
        self._dev(noise)
        assert noise.dtype == torch.float64 and noise.is_contiguous()
        H, W = noise.shape[-2:]
        radius = int(4.0 * sigma + 0.5)
        wts = torch.from_numpy(self.gaussian_weights(sigma, radius)).to(noise.device)
        tmp, out = torch.empty_like(noise), torch.empty_like(noise)
        self._ck(self.lib.uda_field_smooth(noise.data_ptr(), noise.numel() // (H * W), H, W, wts.data_ptr(), radius, float(alpha),
                                           tmp.data_ptr(), out.data_ptr(), self._stream()))
        return out

    def elastic_warp(self, image_u8, label_u8, dx, dy, apply=None):
        for t in (image_u8, label_u8, dx, dy):
            self._dev(t)
            assert t.is_contiguous()
        B, H, W, ch = image_u8.shape
        assert ch == 3 and tuple(label_u8.shape) == (B, H, W) == tuple(dx.shape) == tuple(dy.shape)
        assert image_u8.dtype == label_u8.dtype == torch.uint8 and dx.dtype == dy.dtype == torch.float64
        if apply is not None:
            assert apply.dtype == torch.uint8 and apply.numel() == B and apply.is_cuda
        io, lo = torch.empty_like(image_u8), torch.empty_like(label_u8)
        self._ck(self.lib.uda_elastic_warp(image_u8.data_ptr(), label_u8.data_ptr(), dx.data_ptr(), dy.data_ptr(), _ptr(apply),
                                           B, H, W, io.data_ptr(), lo.data_ptr(), self._stream()))
        return io, lo

    def photometric_u8(self, image_u8, sp_pos, sp_count, sp_value, lut, erase_box):
        """salt-and-pepper scatter, gamma table and box erasing on a uint8 [B,H,W,3] batch, in place."""
        self._dev(image_u8)
        B, H, W, ch = image_u8.shape
        assert ch == 3 and image_u8.dtype == torch.uint8 and image_u8.is_contiguous()
        assert sp_pos.dtype == sp_count.dtype == sp_value.dtype == erase_box.dtype == torch.int32 and lut.dtype == torch.uint8
        assert sp_pos.dim() == 3 and sp_pos.shape[0] == B and sp_pos.shape[2] == 2 and sp_count.numel() == B == sp_value.numel()
        assert tuple(lut.shape) == (B, 256) and tuple(erase_box.shape) == (B, 5)
        for t in (sp_pos, sp_count, sp_value, lut, erase_box):
            assert t.is_cuda and t.is_contiguous()
        self._ck(self.lib.uda_photometric_u8(image_u8.data_ptr(), B, H, W, sp_pos.data_ptr(), sp_count.data_ptr(),
                                             sp_value.data_ptr(), sp_pos.shape[1], lut.data_ptr(), erase_box.data_ptr(),
                                             self._stream()))
        return image_u8

    def postprocess(self, pred, thr_cup, thr_disc, sweeps=None):
        """pred f32 [B,2,H,W] probabilities -> uint8 [B,2,H,W] masks after the reference's evaluation post-processing.
        Runs the tile-wise propagations for ``sweeps`` launches (default: enough for any shape whose geodesic paths cross each
        tile row / column at most twice) and repeats with twice as many while the device reports unfinished tiles."""
        self._dev(pred)
        assert pred.dtype == torch.float32 and pred.is_contiguous() and pred.dim() == 4 and pred.shape[1] == 2
        B, _, H, W = pred.shape
        out = torch.empty(B, 2, H, W, dtype=torch.uint8, device=pred.device)
        flags = torch.zeros(2, dtype=torch.int32, device=pred.device)
        ws = self._ws(pred, self.lib.uda_postprocess_workspace_bytes(B, H, W))
        n = int(sweeps) if sweeps else 2 * ((H + 31) // 32 + (W + 31) // 32) + 4
        for _ in range(6):
            self._ck(self.lib.uda_postprocess(pred.data_ptr(), B, H, W, float(thr_cup), float(thr_disc), n, out.data_ptr(),
                                              flags.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()))
            if int(flags.sum()) == 0:            # host sync: evaluation path only
                return out
            n *= 2
        raise UdaError("uda_postprocess: label propagation did not converge after %d sweeps" % n)

    def adam_step(self, params, grads, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step):
        for t in (params, grads, exp_avg, exp_avg_sq):
            self._dev(t)
            assert t.is_contiguous() and t.dtype == torch.float32 and t.numel() == params.numel()
        self._ck(self.lib.uda_adam_step(params.data_ptr(), grads.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                                        params.numel(), lr, beta1, beta2, eps, int(step), self._stream()))
