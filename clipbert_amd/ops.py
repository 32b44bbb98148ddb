"""Tensor-level wrappers over the C ABI (raw device pointers + current HIP stream).

These are plumbing only: argument marshalling, output allocation with torch (device memory), and
error propagation.  All arithmetic happens in libclipbert_hip.so.
"""
import ctypes as C
import os
from typing import Optional

import torch

from . import _lib
from ._lib import (ACT_GELU, ACT_GELU_SAVE_GRAD, ACT_NONE, ACT_RELU, ACT_SAVED_GRAD, ACT_TANH, CB_BF16, CB_F32, KROW, KROW_GATHER,  # noqa: F401
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


def _extent_bytes(t: torch.Tensor) -> int:
    """bytes from data_ptr() to the end of the tensor's storage (upper bound of any strided view)"""
    return t.untyped_storage().nbytes() - t.storage_offset() * t.element_size()


def _stream(t: torch.Tensor):
    if t.is_cuda:
        return torch.cuda.current_stream(t.device).cuda_stream
    return None


def _chk(rc, what):
    _lib.check(rc, what)


def gemm_desc(a: torch.Tensor, b: torch.Tensor, M: int, N: int, K: int, *, out: torch.Tensor, a_mode=ROWK,
              b_mode=ROWK, lda=None, ldb=None, ldc=None, a_tab=None, b_tab=None, R=1, S=1, Cin=0, H=0, W=0,
              sH=0, sW=0, flip_taps=False, c_rowmap=None, accumulate=False, split_k=1, act=ACT_NONE,
              scale=None, shift=None, residual=None, ldr=None, relu_after=False, mask=None, ldm=None,
              out2=None, ldc2=None, alpha=1.0, dropout_p=0.0, dropout_seed=0, seed_ptr=None, tile=0, gelu_grad_pre=None, a_rowsum=None, batch=1, batch_strides=None,
              relu_bwd=False, post_scale=None, post_scale2=None, xcd_order=0, zero_fill_pitch=0, splitk_ws=None, schedule=0, sq_slots=None) -> GemmDesc:
    """The cb_gemm_desc of C[M,N] (op)= epilogue(sum_k A(m,k) B(n,k)); see include/clipbert_hip.h.  ``splitk_ws``: fp32 scratch for
    the K split of the 8-wave tiles (default: the per-device workspace of ``splitk_workspace``).  The descriptor borrows the
    tensors' memory: the caller keeps them alive until the launch that consumes it has been enqueued (``_refs`` holds them)."""
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
    d.zero_fill_pitch = zero_fill_pitch
    d.c_f32 = int(out.dtype == torch.float32)
    d.accumulate = int(accumulate)            # False / True, or 2 = first writer (C holds nothing; stored; K split only through slabs)
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
    d.dropout_seed_ptr = _ptr(seed_ptr)
    if _KEY_LOG is not None:                 # tools/tune_instep.py: the problem shapes of a step
        _KEY_LOG.append((a_mode, b_mode, M, N, K, batch, R * S, split_k))
    if tile == 0 and _LAUNCH_OVERRIDE:       # tools/tune_instep.py: a launch configuration under test for this problem shape
        ov = _LAUNCH_OVERRIDE.get((a_mode, b_mode, M, N, K, batch, R * S, split_k))
        if ov is not None:
            tile, xcd_order, schedule = ov[0], ov[1], ov[3]
            d.split_k = ov[2] if ov[2] > 0 else split_k
    d.tile = tile
    d.schedule = schedule
    d.xcd_order = xcd_order
    d.a_bytes = _extent_bytes(a)
    d.b_bytes = _extent_bytes(b)
    if gelu_grad_pre is not None:
        d.gelu_grad_pre, d.ld_gelu = _ptr(gelu_grad_pre), gelu_grad_pre.stride(0)
    if a_rowsum is not None:
        assert a_rowsum.dtype == torch.float32
        d.a_rowsum = _ptr(a_rowsum)
    if relu_bwd:
        d.relu_bwd = 1
        d.post_scale, d.post_scale2 = _ptr(post_scale), _ptr(post_scale2)
    if batch > 1:
        d.batch = batch
        d.batch_stride_a, d.batch_stride_b, d.batch_stride_c, d.batch_stride_rowsum = batch_strides
    if splitk_ws is None and d.dtype == CB_BF16 and not _SPLITK_OFF:
        splitk_ws = _SPLITK_SIDE if _SPLITK_SIDE is not None else splitk_workspace(a.device)
    if splitk_ws is not None:
        d.splitk_ws, d.splitk_ws_bytes = _ptr(splitk_ws), splitk_ws.numel() * splitk_ws.element_size()
    if sq_slots is not None:                  # fp32 slots for the tiles' shares of sum(C^2) (cb_gemm_desc.sq_slots)
        assert sq_slots.dtype == torch.float32 and sq_slots.is_contiguous()
        d.sq_slots, d.sq_slots_n = _ptr(sq_slots), sq_slots.numel()
    d._refs = (a, b, out, a_tab, b_tab, c_rowmap, scale, shift, residual, mask, out2, seed_ptr, gelu_grad_pre, a_rowsum, post_scale,
               post_scale2, splitk_ws, sq_slots)
    return d


def gemm(a: torch.Tensor, b: torch.Tensor, M: int, N: int, K: int, *, out: torch.Tensor, **kw):
    """C[M,N] (op)= epilogue(sum_k A(m,k) B(n,k)): one cb_gemm launch on a's current stream (arguments: gemm_desc)."""
    d = gemm_desc(a, b, M, N, K, out=out, **kw)
    _chk(_lib.get().cb_gemm(C.byref(d), _stream(a)), "cb_gemm")
    return out


def gemm_plan(a: torch.Tensor, b: torch.Tensor, M: int, N: int, K: int, *, out: torch.Tensor, use_table: bool = True, **kw):
    """cb_gemm_plan: (tile, split_k, schedule, xcd_order) cb_gemm would launch this call with; launches nothing."""
    d = gemm_desc(a, b, M, N, K, out=out, **kw)
    plan = (C.c_int32 * 4)()
    _chk(_lib.get().cb_gemm_plan(C.byref(d), 1 if use_table else 0, plan), "cb_gemm_plan")
    return tuple(plan)


def gemm_group(descs, like: torch.Tensor):
    """cb_gemm_group: the INDEPENDENT problems ``descs`` (gemm_desc results; no output overlaps another problem's output or
    operands) in as few launches as the library manages, on ``like``'s current stream."""
    n = len(descs)
    if n == 0:
        return
    arr = (GemmDesc * n)()
    for i, d in enumerate(descs):
        C.memmove(C.byref(arr[i]), C.byref(d), C.sizeof(GemmDesc))
    _chk(_lib.get().cb_gemm_group(C.cast(arr, C.c_void_p), n, _stream(like)), "cb_gemm_group")


_KEY_LOG = None                        # a list while tools/tune_instep.py records the keys of a step's problems
_LAUNCH_OVERRIDE = {}                  # (a_mode, b_mode, M, N, K, batch, taps, split_k) -> (tile, xcd_order, split_k, schedule); tuning only
_TILE_ID = {"128x128": 1, "64x64": 2, "128x64": 3, "128x128o2": 4, "8w256x256": 5, "8w128x256": 6, "8w256x128": 7}


def parse_launch_config(name: str):
    """'8w256x256/xcd/s1/m0' -> (tile, xcd_order, split_k, schedule): the configuration names of tools/tune_gemm.py"""
    parts = name.split("/")
    tile = _TILE_ID[parts[0]]
    xcd = 1 if parts[1] == "xcd" else 2
    split = ([int(x[1:]) for x in parts[2:] if x[0] == "s"] or [0])[0]
    sched = ([int(x[1:]) + 1 for x in parts[2:] if x[0] == "m"] or [0])[0]
    return (tile, xcd, split or (1 if tile >= 5 else 0), sched)


# diagnostics (A/B calls): CB_LAUNCH_OVERRIDE="a_mode,b_mode,M,N,K,batch,taps,split_k=config;..." pins launch configurations of problem shapes
for _item in filter(None, os.environ.get("CB_LAUNCH_OVERRIDE", "").split(";")):
    _k, _, _c = _item.partition("=")
    _LAUNCH_OVERRIDE[tuple(int(x) for x in _k.split(","))] = parse_launch_config(_c)
_SPLITK_WS = {}
_SPLITK_OFF = False                    # set while launches go to a SIDE stream (Runtime.side): the buffer belongs to the main stream's launches
_SPLITK_SIDE = None                    # set while launches go to a side stream that OWNS a scratch of its own (Runtime.side with side_ws)
SPLITK_WS_BYTES = 128 << 20


def splitk_workspace(device) -> torch.Tensor:
    """One fp32 scratch buffer per device for the K-split partial products of cb_gemm's 8-wave tiles.  Launches that follow one
    another on a stream (or on streams ordered by events, as a hipGraph capture after an eager warm-up) share it; launches on a
    concurrent side stream must not (ops._SPLITK_OFF).  Allocated on first use / by ClipBert.prepare(), never inside a capture."""
    dev = torch.device(device)
    key = str(dev)
    ws = _SPLITK_WS.get(key)
    if ws is None:
        if dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
            return None
        ws = new_splitk_workspace(device)
        _SPLITK_WS[key] = ws
    return ws


def new_splitk_workspace(device, nbytes: int = SPLITK_WS_BYTES) -> torch.Tensor:
    """A K-split scratch as include/clipbert_hip.h lays it out: partial products, then CB_SPLITK_WS_COUNTER_BYTES of arrival counters
    that start at zero (every launch leaves them zero).  One per stream whose launches may overlap another stream's."""
    assert nbytes % 16 == 0 and nbytes > _lib.SPLITK_WS_COUNTER_BYTES
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
    ws[-(_lib.SPLITK_WS_COUNTER_BYTES // 4):].zero_()
    return ws


def build_pixel_table(n: int, oh: int, ow: int, stride: int, pad: int, sN: int, sH: int, sW: int,
                      device) -> torch.Tensor:
    """int64 tensor viewed as cb_pixel {int32 off; int16 ih0; int16 iw0} entries."""
    tab = torch.empty(n * oh * ow, dtype=torch.int64, device=device)
    _chk(_lib.get().cb_build_pixel_table(_ptr(tab), n, oh, ow, stride, pad, sN, sH, sW, _stream(tab)),
         "cb_build_pixel_table")
    return tab


def _f3(vals):
    return (C.c_float * 3)(*[float(v) for v in vals])


def stem_pack(src: torch.Tensor, dtype: torch.dtype, pad: int = 3, mean=None, std=None, extra_w: int = 0) -> torch.Tensor:
    """(N,3,H,W) fp32 (already normalised) or uint8 (+mean/std) -> (N, H+2*pad, W+2*pad+extra_w, 4) zero-padded BGR0."""
    n, c, h, w = src.shape
    assert c == 3 and src.is_contiguous()
    hp, wp = h + 2 * pad, w + 2 * pad + extra_w
    dst = torch.empty(n, hp, wp, 4, dtype=dtype, device=src.device)
    u8 = src.dtype == torch.uint8
    assert u8 or src.dtype == torch.float32
    m3 = _f3(mean) if u8 else None
    s3 = _f3(std) if u8 else None
    _chk(_lib.get().cb_stem_pack(dtype_code(dtype), _ptr(src), int(u8), m3, s3, _ptr(dst), n, h, w, hp, wp, pad,
                                 _stream(src)), "cb_stem_pack")
    return dst


def image_norm(frames_u8: torch.Tensor, mean, std) -> torch.Tensor:
    """ImageNorm (a1): uint8 (..., 3, H, W) -> fp32."""
    assert frames_u8.dtype == torch.uint8 and frames_u8.is_contiguous() and frames_u8.shape[-3] == 3
    out = torch.empty(frames_u8.shape, dtype=torch.float32, device=frames_u8.device)
    hw = frames_u8.shape[-1] * frames_u8.shape[-2]
    n = frames_u8.numel() // (3 * hw)
    _chk(_lib.get().cb_image_norm(_ptr(frames_u8), _ptr(out), _f3(mean), _f3(std), n, hw, _stream(out)),
         "cb_image_norm")
    return out


def maxpool_fwd(x: torch.Tensor, k: int, stride: int, pad: int, relu: bool = False) -> torch.Tensor:
    """x (N,H,W,C) NHWC."""
    n, h, w, c = x.shape
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    y = torch.empty(n, oh, ow, c, dtype=x.dtype, device=x.device)
    _chk(_lib.get().cb_maxpool_fwd(dtype_code(x.dtype), _ptr(x), _ptr(y), n, h, w, c, oh, ow, k, stride, pad,
                                   int(relu), _stream(x)), "cb_maxpool_fwd")
    return y


def stem_pool(packed, weight, scale, shift, oh: int, ow: int) -> torch.Tensor:
    """cb_stem_pool: stem convolution + FrozenBN + ReLU + 3x3/2 max-pool of the packed (N, Hp, Wp, 4) bf16 image in one launch ->
    (N, PH, PW, 64)."""
    n, hp, wp, _ = packed.shape
    assert packed.dtype == torch.bfloat16 and packed.is_contiguous() and weight.dtype == torch.bfloat16
    ph, pw = (oh - 1) // 2 + 1, (ow - 1) // 2 + 1
    out = torch.empty(n, ph, pw, 64, dtype=packed.dtype, device=packed.device)
    _chk(_lib.get().cb_stem_pool(_ptr(packed), _ptr(weight), _ptr(scale), _ptr(shift), _ptr(out), n, hp, wp, oh, ow, ph, pw, _stream(packed)),
         "cb_stem_pool")
    return out


def stem_pool_u8(frames: torch.Tensor, mean, std, weight, scale, shift) -> torch.Tensor:
    """cb_stem_pool_u8: (N, 3, H, W) uint8 RGB frames -> (N, PH, PW, 64) bf16: ImageNorm + BGR flip + zero padding + stem convolution + FrozenBN
    + ReLU + 3x3/2 max-pool in one launch (cb_stem_pack folded into cb_stem_pool; bit-identical to the two launches)."""
    n, c, h, w = frames.shape
    assert frames.dtype == torch.uint8 and frames.is_contiguous() and c == 3 and weight.dtype == torch.bfloat16
    oh, ow = (h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1
    ph, pw = (oh - 1) // 2 + 1, (ow - 1) // 2 + 1
    out = torch.empty(n, ph, pw, 64, dtype=torch.bfloat16, device=frames.device)
    _chk(_lib.get().cb_stem_pool_u8(_ptr(frames), _f3(mean), _f3(std), _ptr(weight), _ptr(scale), _ptr(shift), _ptr(out), n, h, w, oh, ow, ph, pw,
                                    _stream(frames)), "cb_stem_pool_u8")
    return out


def res2_block(x, w1, w2, w3, ss1, ss2, ss3, wsc=None, sssc=None) -> torch.Tensor:
    """cb_res2_block: one res2 bottleneck block (64 mid channels, stride 1, FrozenBN, forward only) in one launch.  x (N, H, W, cin) bf16
    NHWC; w*: the convolutions' KRSC weight images; ss*: their (scale, shift) fp32 vectors; wsc / sssc: the projection shortcut of the
    stage-entry block (cin = 64), None for an identity shortcut (cin = 256)."""
    n, h, w, cin = x.shape
    assert x.dtype == torch.bfloat16 and x.is_contiguous()
    out = torch.empty(n, h, w, 256, dtype=x.dtype, device=x.device)
    d = _lib.Res2Desc()
    d.x, d.out = _ptr(x), _ptr(out)
    d.w1, d.w2, d.w3, d.wsc = _ptr(w1), _ptr(w2), _ptr(w3), _ptr(wsc)
    d.scale1, d.shift1, d.scale2, d.shift2, d.scale3, d.shift3 = (_ptr(t) for t in (*ss1, *ss2, *ss3))
    d.scale_sc, d.shift_sc = (_ptr(sssc[0]), _ptr(sssc[1])) if sssc is not None else (None, None)
    d.N, d.H, d.W, d.cin = n, h, w, cin
    d._refs = (x, out, w1, w2, w3, wsc, ss1, ss2, ss3, sssc)
    _chk(_lib.get().cb_res2_block(C.byref(d), _stream(x)), "cb_res2_block")
    return out


def maxpool2_bwd(x, y, dy, relu: bool = False) -> torch.Tensor:
    n, h, w, c = x.shape
    oh, ow = y.shape[1], y.shape[2]
    dx = torch.empty_like(x)
    _chk(_lib.get().cb_maxpool2_bwd(dtype_code(x.dtype), _ptr(x), _ptr(y), _ptr(dy), _ptr(dx), n, h, w, c, oh, ow,
                                    int(relu), _stream(x)), "cb_maxpool2_bwd")
    return dx


def relu_scale_bwd(dy, y, scale=None, want_dz=False, scale2=None):
    """returns (g = dy*(y>0)*scale | None, dz = dy*(y>0) | None, g2 = dz*scale2 | None)."""
    c = y.shape[-1]
    rows = y.numel() // c
    g = torch.empty_like(y) if scale is not None else None
    dz = torch.empty_like(y) if want_dz else None
    g2 = torch.empty_like(y) if scale2 is not None else None
    _chk(_lib.get().cb_relu_scale_bwd(dtype_code(y.dtype), _ptr(dy), _ptr(y), _ptr(scale), _ptr(g), _ptr(dz),
                                      _ptr(scale2), _ptr(g2), rows, c, _stream(y)), "cb_relu_scale_bwd")
    return g, dz, g2


def layernorm_fwd(x, gamma, beta, eps, save_stats=False, out=None):
    d = x.shape[-1]
    rows = x.numel() // d
    y = out if out is not None else torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device) if save_stats else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if save_stats else None
    _chk(_lib.get().cb_layernorm_fwd(dtype_code(x.dtype), _ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(mean),
                                     _ptr(rstd), rows, d, eps, 0, 0, 0, _stream(x)), "cb_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, dropout_p=0.0, dropout_seed=0, seed_ptr=None,
                  dx=None, rows=None, seg=(0, 0, 0), dx2=None):
    """returns (dx, dx_dropped|None); dgamma/dbeta (fp32) are accumulated in place.
    seg = (seg_len, seg_stride, seg_off) addresses a strided subset of rows (see the header)."""
    d = x.shape[-1]
    rows = x.numel() // d if rows is None else rows
    dx = torch.empty_like(x) if dx is None else dx
    dx2 = (dx2 if dx2 is not None else torch.empty_like(x)) if dropout_p > 0 else None
    _chk(_lib.get().cb_layernorm_bwd(dtype_code(x.dtype), _ptr(dy), _ptr(x), _ptr(gamma), _ptr(mean), _ptr(rstd),
                                     _ptr(dx), _ptr(dgamma), _ptr(dbeta), rows, d, _ptr(dx2), dropout_p,
                                     dropout_seed, _ptr(seed_ptr), seg[0], seg[1], seg[2], _stream(x)),
         "cb_layernorm_bwd")
    return dx, dx2


def ln_part_blocks(rows: int) -> int:
    """blocks cb_layernorm_bwd_part is launched with for `rows` rows (one row per wave, 12 waves per block, <= 256 blocks)"""
    return max(1, min(256, (rows + 11) // 12))


def layernorm_bwd_part(dy, x, gamma, mean, rstd, part, dropout_p=0.0, dropout_seed=0, seed_ptr=None, dx=None, dx2=None):
    """LayerNorm backward whose parameter-gradient partial sums go to part (nblocks, 2, D) fp32 (plain stores; ln_partials_reduce
    adds them to the gradients later).  returns (dx, dx_dropped|None)"""
    d = x.shape[-1]
    rows = x.numel() // d
    nblocks = part.shape[0]
    assert part.dtype == torch.float32 and part.is_contiguous() and part.shape[1:] == (2, d)
    dx = torch.empty_like(x) if dx is None else dx
    dx2 = (dx2 if dx2 is not None else torch.empty_like(x)) if dropout_p > 0 else None
    _chk(_lib.get().cb_layernorm_bwd_part(dtype_code(x.dtype), _ptr(dy), _ptr(x), _ptr(gamma), _ptr(mean), _ptr(rstd), _ptr(dx),
                                          _ptr(part), nblocks, rows, d, _ptr(dx2), dropout_p, dropout_seed, _ptr(seed_ptr),
                                          0, 0, 0, _stream(x)), "cb_layernorm_bwd_part")
    return dx, dx2


def ln_partials_reduce(part, grad, off_gamma, off_beta):
    """grad[off_gamma[j] + c] += sum_b part[j, b, 0, c]; grad[off_beta[j] + c] += sum_b part[j, b, 1, c]  (fixed order)"""
    njobs, nblocks, two, d = part.shape
    assert two == 2 and part.is_contiguous() and off_gamma.dtype == torch.int64 and off_gamma.numel() == njobs == off_beta.numel()
    _chk(_lib.get().cb_ln_partials_reduce(_ptr(part), _ptr(grad), _ptr(off_gamma), _ptr(off_beta), njobs, nblocks, d, _stream(part)),
         "cb_ln_partials_reduce")


def text_embed_fwd(ids, word, pos, type0, gamma, beta, out, pre, mean, rstd, lt, l_total, eps, attn_mask=None, key_mask=None,
                   repeat=1):
    """``repeat`` = n: the (P, Lt) ids / attn_mask are the text batch of ONE clip; n*P output rows are produced (row b reads row
    b % P).  ``key_mask`` (B, L_total) fp32 receives the text columns of the attention key mask."""
    b = ids.shape[0] * repeat
    d = word.shape[-1]
    _chk(_lib.get().cb_text_embed_fwd(dtype_code(word.dtype), _ptr(ids), _ptr(word), _ptr(pos), _ptr(type0),
                                      _ptr(gamma), _ptr(beta), _ptr(out), _ptr(pre), _ptr(mean), _ptr(rstd), b, lt,
                                      l_total, d, eps, _ptr(attn_mask), _ptr(key_mask), ids.shape[0] if repeat > 1 else 0,
                                      _stream(out)), "cb_text_embed_fwd")


def visual_embed_fwd(grid, src_row, sel, row_emb, col_emb, type0, gamma, beta, out, pre, mean, rstd, b, lv, lt,
                     l_total, eps, key_mask=None):
    _, t, hg, wg, d = grid.shape
    _chk(_lib.get().cb_visual_embed_fwd(dtype_code(grid.dtype), _ptr(grid), _ptr(src_row), _ptr(sel), _ptr(row_emb),
                                        _ptr(col_emb), _ptr(type0), _ptr(gamma), _ptr(beta), _ptr(out), _ptr(pre),
                                        _ptr(mean), _ptr(rstd), b, t, hg, wg, lv, lt, l_total, d, eps, _ptr(key_mask),
                                        _stream(out)), "cb_visual_embed_fwd")


def text_embed_bwd(dpre, ids, dword, dpos, dtype0, lt, l_total, pad_id, repeat=1):
    b = ids.shape[0] * repeat
    d = dword.shape[-1]
    _chk(_lib.get().cb_text_embed_bwd(dtype_code(dpre.dtype), _ptr(dpre), _ptr(ids), _ptr(dword), _ptr(dpos),
                                      _ptr(dtype0), b, lt, l_total, d, pad_id, ids.shape[0] if repeat > 1 else 0, _stream(dpre)),
         "cb_text_embed_bwd")


def mean_fwd(x: torch.Tensor) -> torch.Tensor:
    """scalar mean of a contiguous fp32 vector (the runners' loss.mean())"""
    out = torch.empty((), dtype=torch.float32, device=x.device)
    _chk(_lib.get().cb_mean_fwd(_ptr(x), x.numel(), _ptr(out), _stream(x)), "cb_mean_fwd")
    return out


def mean_bwd(dmean: torch.Tensor, n: int, like: torch.Tensor) -> torch.Tensor:
    dx = torch.empty(n, dtype=torch.float32, device=like.device)
    _chk(_lib.get().cb_mean_bwd(_ptr(dmean), n, _ptr(dx), _stream(like)), "cb_mean_bwd")
    return dx


def counter_add(counter: torch.Tensor, inc: int = 1):
    """*counter += inc on a device int64 word (the dropout seed word: advanced inside a captured step)"""
    assert counter.dtype == torch.int64 and counter.numel() == 1
    _chk(_lib.get().cb_counter_add(_ptr(counter), inc, _stream(counter)), "cb_counter_add")


def zero_(t: torch.Tensor) -> torch.Tensor:
    """t[...] = 0 for a contiguous tensor (or contiguous slice of one): cb_zero on t's stream"""
    assert t.is_contiguous()
    if t.numel():
        _chk(_lib.get().cb_zero(_ptr(t), t.numel() * t.element_size(), _stream(t)), "cb_zero")
    return t


def zero_many(tensors) -> None:
    """every (contiguous) tensor of the list = 0, four per launch (cb_zero_ranges) on the first one's stream"""
    ts = [t for t in tensors if t is not None and t.numel()]
    for i in range(0, len(ts), 4):
        chunk = ts[i:i + 4]
        if len(chunk) == 1:
            zero_(chunk[0])
            continue
        assert all(t.is_contiguous() for t in chunk)
        ptrs = (C.c_void_p * len(chunk))(*[_ptr(t) for t in chunk])
        nbytes = (C.c_int64 * len(chunk))(*[t.numel() * t.element_size() for t in chunk])
        _chk(_lib.get().cb_zero_ranges(ptrs, nbytes, len(chunk), _stream(chunk[0])), "cb_zero_ranges")


def zeros(shape, dtype, device) -> torch.Tensor:
    return zero_(torch.empty(shape, dtype=dtype, device=device))


def visual_embed_bwd(dpre, src_row, sel, dgrid, drow, dcol, dtype0, b, lv, lt, l_total):
    _, t, hg, wg, d = dgrid.shape
    _chk(_lib.get().cb_visual_embed_bwd(dtype_code(dpre.dtype), _ptr(dpre), _ptr(src_row), _ptr(sel), _ptr(dgrid),
                                        _ptr(drow), _ptr(dcol), _ptr(dtype0), b, t, hg, wg, lv, lt, l_total, d,
                                        _stream(dpre)), "cb_visual_embed_bwd")


def attention_fwd(qkv, key_mask, b, l, h, save_lse=False, dropout_p=0.0, dropout_seed=0, seed_ptr=None, out=None):
    ctx = out if out is not None else torch.empty(b * l, h * 64, dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(b * h * l, dtype=torch.float32, device=qkv.device) if save_lse else None
    _chk(_lib.get().cb_attention_fwd(dtype_code(qkv.dtype), _ptr(qkv), _ptr(key_mask), _ptr(ctx), _ptr(lse), b, l, h,
                                     dropout_p, dropout_seed, _ptr(seed_ptr), _stream(qkv)), "cb_attention_fwd")
    return ctx, lse


def attention_bwd(qkv, key_mask, ctx, dctx, lse, b, l, h, dropout_p=0.0, dropout_seed=0, seed_ptr=None, out=None):
    dqkv = out if out is not None else torch.empty_like(qkv)
    ws = torch.empty(b * h * l, dtype=torch.float32, device=qkv.device)
    _chk(_lib.get().cb_attention_bwd(dtype_code(qkv.dtype), _ptr(qkv), _ptr(key_mask), _ptr(ctx), _ptr(dctx),
                                     _ptr(lse), _ptr(ws), _ptr(dqkv), b, l, h, dropout_p, dropout_seed, _ptr(seed_ptr),
                                     _stream(qkv)),
         "cb_attention_bwd")
    return dqkv


def cross_entropy(logits, labels, want_loss=True, dloss=None, want_grad=False, ignore_index=-100):
    """logits fp32 (rows, C), row stride may exceed C (padded vocab logits)."""
    rows, c = logits.shape
    assert logits.dtype == torch.float32 and logits.stride(1) == 1
    ld = logits.stride(0)
    loss = torch.empty(rows, dtype=torch.float32, device=logits.device) if want_loss else None
    dlogits = None
    if want_grad:
        dlogits = torch.empty_strided(logits.shape, logits.stride(), dtype=torch.float32, device=logits.device)
    _chk(_lib.get().cb_cross_entropy(_ptr(logits), ld, _ptr(labels), _ptr(loss), _ptr(dlogits), _ptr(dloss), rows, c,
                                     ignore_index, _stream(logits)), "cb_cross_entropy")
    return loss, dlogits


LOSS_MSE, LOSS_BCE, LOSS_RANK = 0, 1, 2


def head_loss(kind: int, x, y=None, want_loss=True, dloss=None, want_grad=False, group=1, margin=0.0):
    """cb_head_loss on contiguous fp32 logits: (loss, dx).  kind LOSS_RANK: x (rows, group), loss (rows, group - 1)."""
    assert x.dtype == torch.float32 and x.is_contiguous() and (y is None or (y.dtype == torch.float32 and y.is_contiguous() and y.numel() == x.numel()))
    n = x.numel()
    if kind == LOSS_RANK:
        lshape = (n // group, group - 1)
    else:
        lshape = tuple(x.shape)
    loss = torch.empty(lshape, dtype=torch.float32, device=x.device) if want_loss else None
    dx = torch.empty_like(x) if want_grad else None
    _chk(_lib.get().cb_head_loss(kind, _ptr(x), _ptr(y), _ptr(loss), _ptr(dloss), _ptr(dx), n, group, float(margin), _stream(x)), "cb_head_loss")
    return loss, dx


def retrieval_scores(logits) -> torch.Tensor:
    """(rows, 2) -> softmax[:, 1]; (rows, 1) or (rows,) -> sigmoid; fp32"""
    assert logits.dtype == torch.float32 and logits.is_contiguous()
    c = logits.shape[1] if logits.dim() == 2 else 1
    rows = logits.numel() // c
    out = torch.empty(rows, dtype=torch.float32, device=logits.device)
    _chk(_lib.get().cb_retrieval_scores(_ptr(logits), _ptr(out), rows, c, _stream(logits)), "cb_retrieval_scores")
    return out


def colsum(g, out, m=None, n=None, ldg=None):
    """out[n] += sum_m g[m, n] (fp32 atomics)."""
    m = g.shape[0] if m is None else m
    n = g.shape[1] if n is None else n
    _chk(_lib.get().cb_colsum(dtype_code(g.dtype), _ptr(g), ldg if ldg is not None else g.stride(0), _ptr(out), m, n,
                              _stream(g)), "cb_colsum")


def cast(src, dst):
    assert src.numel() == dst.numel()
    _chk(_lib.get().cb_cast(dtype_code(src.dtype), _ptr(src), dtype_code(dst.dtype), _ptr(dst), src.numel(),
                            _stream(src)), "cb_cast")
    return dst


def act_bwd(act, dy, ref):
    dx = torch.empty_like(dy)
    _chk(_lib.get().cb_act_bwd(dtype_code(dy.dtype), act, _ptr(dy), _ptr(ref), _ptr(dx), dy.numel(), _stream(dy)),
         "cb_act_bwd")
    return dx


def sq_sum(g, out, ws=None):
    """out += sum(g^2); with a scratch tensor ``ws`` (fp32, <= 1024 floats used) the result is independent of workgroup timing"""
    if g.dtype == torch.bfloat16:
        assert ws is not None, "bf16 gradients use the order-independent reduction (pass ws)"
        _chk(_lib.get().cb_sq_sum_det_bf16(_ptr(g), g.numel(), _ptr(out), _ptr(ws), ws.numel(), _stream(g)), "cb_sq_sum_det_bf16")
    elif ws is not None:
        _chk(_lib.get().cb_sq_sum_det(_ptr(g), g.numel(), _ptr(out), _ptr(ws), ws.numel(), _stream(g)), "cb_sq_sum_det")
    else:
        _chk(_lib.get().cb_sq_sum(_ptr(g), g.numel(), _ptr(out), _stream(g)), "cb_sq_sum")


def sq_slot_count(M: int, N: int, batch: int = 1) -> int:
    """slots a cb_gemm call with ``sq_slots`` needs at most: one per 64 x 64 output tile and batch member"""
    return ((M + 63) // 64) * ((N + 63) // 64) * batch


def sq_sum_fold(g, segments, slots, out, ws):
    """out += sum of g[lo:hi]^2 over ``segments`` (<= 4 (lo, hi) element ranges) + sum(slots): the squared gradient norm of a step whose
    weight-gradient launches left their shares in ``slots`` (cb_sq_sum_fold; fixed order of addition)"""
    segs = [(int(lo), int(hi)) for lo, hi in segments if hi > lo]
    assert len(segs) <= 4
    arr = (C.c_int64 * (2 * max(1, len(segs))))(*[x for s in segs for x in s])
    _chk(_lib.get().cb_sq_sum_fold(_ptr(g), C.cast(arr, C.c_void_p), len(segs), _ptr(slots), slots.numel() if slots is not None else 0, _ptr(out), _ptr(ws),
                                   ws.numel(), _stream(out)), "cb_sq_sum_fold")


def adamw_hyper(lr, beta1, beta2, eps, weight_decay, step, max_norm=-1.0, grad_scale=1.0):
    """Host-side packing of the CB_HP_* array (see include/clipbert_hip.h)."""
    return [lr, beta1, beta2, eps, weight_decay, 1.0 - beta1 ** step, 1.0 - beta2 ** step, max_norm, grad_scale, 0.0, 0.0]


def adamw(p, g, m, v, w16, hyper_dev, grad_sq_sum=None):
    """hyper_dev: DEVICE fp32 tensor of >= HP_COUNT entries (adamw_hyper).  g: fp32, or bf16 (the reduced data-parallel wire image)."""
    if g.dtype == torch.bfloat16:
        _chk(_lib.get().cb_adamw_g16(_ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(w16), p.numel(), _ptr(hyper_dev), _ptr(grad_sq_sum),
                                     _stream(p)), "cb_adamw_g16")
        return
    _chk(_lib.get().cb_adamw(_ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(w16), p.numel(), _ptr(hyper_dev),
                             _ptr(grad_sq_sum), _stream(p)), "cb_adamw")


def dropout(x, p, seed=0, seed_ptr=None, out=None):
    y = torch.empty_like(x) if out is None else out
    _chk(_lib.get().cb_dropout(dtype_code(x.dtype), _ptr(x), _ptr(y), x.numel(), p, seed, _ptr(seed_ptr), _stream(x)),
         "cb_dropout")
    return y


def elu_bn1d_fwd(x, gamma, beta, running_mean, running_var, training: bool, momentum: float = 0.1, eps: float = 1e-5, save: bool = False):
    """ELU + BatchNorm1d over the batch (ClipBertForRegression.regressor[1:3]); returns (y, save_mean, save_invstd)."""
    b, d = x.shape
    y = torch.empty_like(x)
    sm = torch.empty(d, dtype=torch.float32, device=x.device) if save else None
    si = torch.empty(d, dtype=torch.float32, device=x.device) if save else None
    _chk(_lib.get().cb_elu_bn1d_fwd(dtype_code(x.dtype), _ptr(x), _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var), _ptr(y),
                                    _ptr(sm), _ptr(si), b, d, int(training), momentum, eps, _stream(x)), "cb_elu_bn1d_fwd")
    return y, sm, si


def elu_bn1d_bwd(dy, x, gamma, save_mean, save_invstd, dgamma, dbeta, training: bool):
    b, d = x.shape
    dx = torch.empty_like(x)
    _chk(_lib.get().cb_elu_bn1d_bwd(dtype_code(x.dtype), _ptr(dy), _ptr(x), _ptr(gamma), _ptr(save_mean), _ptr(save_invstd), _ptr(dx),
                                    _ptr(dgamma), _ptr(dbeta), b, d, int(training), _stream(x)), "cb_elu_bn1d_bwd")
    return dx


AGG_MEAN, AGG_MAX, AGG_LSE = 0, 1, 2


def clip_aggregate_fwd(stack: torch.Tensor, mode: int):
    """stack: (n_clips, ...) fp32 contiguous -> (out (...), argmax int32 | None)."""
    assert stack.dtype == torch.float32 and stack.is_contiguous()
    n = stack.shape[0]
    out = torch.empty(stack.shape[1:], dtype=torch.float32, device=stack.device)
    am = torch.empty(stack.shape[1:], dtype=torch.int32, device=stack.device) if mode == AGG_MAX else None
    _chk(_lib.get().cb_clip_aggregate_fwd(_ptr(stack), n, out.numel(), mode, _ptr(out), _ptr(am), _stream(stack)),
         "cb_clip_aggregate_fwd")
    return out, am


def clip_aggregate_bwd(dout: torch.Tensor, stack, out, argmax, n_clips: int, mode: int):
    dout = dout.contiguous()
    dx = torch.empty((n_clips,) + tuple(dout.shape), dtype=torch.float32, device=dout.device)
    _chk(_lib.get().cb_clip_aggregate_bwd(_ptr(dout), _ptr(stack), _ptr(out), _ptr(argmax), n_clips, dout.numel(), mode, _ptr(dx),
                                          _stream(dout)), "cb_clip_aggregate_bwd")
    return dx


def lse_loss(stack: torch.Tensor, labels: torch.Tensor, want_loss=True, dloss=None, want_grad=False):
    """stack: (n_clips, B, C) fp32 clip-major logits -> (loss (B,) | None, dlogits like stack | None)."""
    assert stack.dtype == torch.float32 and stack.is_contiguous() and stack.dim() == 3
    n, b, c = stack.shape
    loss = torch.empty(b, dtype=torch.float32, device=stack.device) if want_loss else None
    dx = torch.empty_like(stack) if want_grad else None
    _chk(_lib.get().cb_lse_loss(_ptr(stack), _ptr(labels.contiguous()), n, b, c, _ptr(loss), _ptr(dloss), _ptr(dx), _stream(stack)),
         "cb_lse_loss")
    return loss, dx
