"""Tensor-level wrappers over the C ABI (raw device pointers + current HIP stream).

These are plumbing only: argument marshalling, output allocation with torch (device memory), and
error propagation.  All arithmetic happens in libclipbert_hip.so.
"""
import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import (ACT_GELU, ACT_NONE, ACT_RELU, ACT_TANH, CB_BF16, CB_F32, KROW, KROW_GATHER,  # noqa: F401
                   KROW_TAPS, ROWK, ROWK_GATHER, GemmDesc)

# Set only by the CPU test-suite when it swaps in the host emulator build of the same sources.
_ALLOW_HOST_POINTERS = False


def dtype_code(t: torch.dtype) -> int:
    if t == torch.bfloat16:
        return CB_BF16
    if t == torch.float32:
        return CB_F32
    raise TypeError(f"unsupported dtype {t}")


def _ptr(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not t.is_cuda and not _ALLOW_HOST_POINTERS:
        raise RuntimeError("clipbert_amd ops need CUDA(HIP) tensors: there is no CPU path")
    return t.data_ptr()


def _stream(t: torch.Tensor):
    if t.is_cuda:
        return torch.cuda.current_stream(t.device).cuda_stream
    return None


def _chk(rc, what):
    _lib.check(rc, what)


def gemm(a: torch.Tensor, b: torch.Tensor, M: int, N: int, K: int, *, out: torch.Tensor, a_mode=ROWK,
         b_mode=ROWK, lda=None, ldb=None, ldc=None, a_tab=None, b_tab=None, R=1, S=1, Cin=0, H=0, W=0,
         sH=0, sW=0, flip_taps=False, c_rowmap=None, accumulate=False, split_k=1, act=ACT_NONE,
         scale=None, shift=None, residual=None, ldr=None, relu_after=False, mask=None, ldm=None,
         out2=None, ldc2=None, alpha=1.0, dropout_p=0.0, dropout_seed=0, tile=0):
    """C[M,N] (op)= epilogue(sum_k A(m,k) B(n,k)); see include/clipbert_hip.h cb_gemm_desc."""
    d = GemmDesc()
    d.dtype = dtype_code(a.dtype)
    assert b.dtype == a.dtype
    d.M, d.N, d.K = M, N, K
    d.a_mode, d.b_mode = a_mode, b_mode
    d.A, d.B = _ptr(a), _ptr(b)
    d.lda = lda if lda is not None else a.stride(0)
    d.ldb = ldb if ldb is not None else b.stride(0)
    d.a_tab, d.b_tab = _ptr(a_tab), _ptr(b_tab)
    d.R, d.S, d.Cin, d.H, d.W, d.sH, d.sW = R, S, Cin, H, W, sH, sW
    d.flip_taps = int(flip_taps)
    d.C = _ptr(out)
    d.ldc = ldc if ldc is not None else out.stride(0)
    d.c_rowmap = _ptr(c_rowmap)
    d.c_f32 = int(out.dtype == torch.float32)
    d.accumulate = int(accumulate)
    d.split_k = split_k
    d.act = act
    d.scale, d.shift = _ptr(scale), _ptr(shift)
    d.residual = _ptr(residual)
    d.ldr = (ldr if ldr is not None else residual.stride(0)) if residual is not None else 0
    d.relu_after = int(relu_after)
    d.mask = _ptr(mask)
    d.ldm = (ldm if ldm is not None else mask.stride(0)) if mask is not None else 0
    d.C2 = _ptr(out2)
    d.ldc2 = (ldc2 if ldc2 is not None else out2.stride(0)) if out2 is not None else 0
    d.alpha = alpha
    d.dropout_p = dropout_p
    d.dropout_seed = dropout_seed
    d.tile = tile
    _chk(_lib.get().cb_gemm(C.byref(d), _stream(a)), "cb_gemm")
    return out


def build_pixel_table(n: int, oh: int, ow: int, stride: int, pad: int, sN: int, sH: int, sW: int,
                      device) -> torch.Tensor:
    """int64 tensor viewed as cb_pixel {int32 off; int16 ih0; int16 iw0} entries."""
    tab = torch.empty(n * oh * ow, dtype=torch.int64, device=device)
    _chk(_lib.get().cb_build_pixel_table(_ptr(tab), n, oh, ow, stride, pad, sN, sH, sW, _stream(tab)),
         "cb_build_pixel_table")
    return tab
