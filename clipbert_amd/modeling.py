"""ClipBERT module API on top of libclipbert_hip (MI355X / gfx950).

The classes keep the reference's names, constructor / forward signatures, returned dicts and
state-dict keys:

    ClipBert(config, input_format="BGR", detectron2_model_cfg=..., transformer_cls=...)   e2e_model.py:14-50
    GridFeatBackbone                                                                      grid_feat.py:37-105
    ClipBertForPreTraining / ...VideoTextRetrieval / ...MultipleChoice / ...SequenceClassification
                                                                                          modeling.py:241-580

but they hold no PyTorch compute: parameters are views into flat HBM buffers (params.ParamBank) and
every forward / backward step is a call into the C ABI (clipbert_amd.ops).  The backward pass is written
out explicitly (two coarse autograd nodes: CNN trunk, cross-modal encoder) so that residual-gradient
sums, ReLU/FrozenBN masks and bias/LayerNorm reductions are fused into kernel epilogues instead of
being left to autograd's eager tensor ops.  Parameter gradients are accumulated by the kernels
directly into the flat fp32 gradient buffer (``p.grad`` is a view of it).
"""
import contextlib
import math
import os
from types import SimpleNamespace
from typing import List, Optional

import numpy as np
import torch
from torch import nn

from . import ops
from .ops import (ACT_GELU, ACT_NONE, ACT_RELU, ACT_TANH, KROW, KROW_GATHER, KROW_TAPS, ROWK, ROWK_GATHER)

_GELU_SAVE_GRAD = os.environ.get("CB_NO_GELU_SAVE_GRAD") is None      # FFN1 stores gelu'(pre) for the backward instead of the pre-activation
from .params import ParamBank

FROZEN_BN_EPS = 1e-5
RESNET50_STAGES = (("res2", 3, 64, 256, 1), ("res3", 4, 128, 512, 2), ("res4", 6, 256, 1024, 2),
                   ("res5", 3, 512, 2048, 2))


def _cfg_get(config, key, default=None):
    if isinstance(config, dict):
        return config.get(key, default)
    return getattr(config, key, default)


def as_config(config):
    """Accepts a dict (src/configs/base_model.json contents + task keys) or any attribute bag."""
    if isinstance(config, dict):
        return SimpleNamespace(**config)
    return config


# =================================================================================================
# parameter holders (names chosen so that state_dict() keys equal the reference's)
# =================================================================================================
class Linear(nn.Module):
    def __init__(self, in_features, out_features, std=0.02):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.randn(out_features, in_features) * std)
        self.bias = nn.Parameter(torch.zeros(out_features))


class Embedding(nn.Module):
    def __init__(self, n, dim, std=0.02, padding_idx=None):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(n, dim) * std)
        self.padding_idx = padding_idx
        if padding_idx is not None:
            with torch.no_grad():
                self.weight[padding_idx].zero_()


class LayerNorm(nn.Module):
    def __init__(self, dim, eps):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))
        self.eps = eps


class FrozenBatchNorm2d(nn.Module):
    """detectron2.layers.FrozenBatchNorm2d: four buffers, y = x*scale + shift with fixed statistics."""
    def __init__(self, c):
        super().__init__()
        self.register_buffer("weight", torch.ones(c))
        self.register_buffer("bias", torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))


class Conv2d(nn.Module):
    """Conv weight in the reference's OIHW logical shape (channels_last memory image = KRSC) plus an
    optional FrozenBN child called ``norm`` as detectron2 names it."""
    def __init__(self, cin, cout, k, stride=1, pad=0, norm=True):
        super().__init__()
        self.cin, self.cout, self.k, self.stride, self.pad = cin, cout, k, stride, pad
        w = torch.randn(cout, cin, k, k) * math.sqrt(2.0 / (cout * k * k))
        self.weight = nn.Parameter(w.contiguous(memory_format=torch.channels_last))
        self.norm = FrozenBatchNorm2d(cout) if norm else None
        self._ss = None

    def scale_shift(self):
        """fp32 per-channel (scale, shift) of the frozen affine; cached (buffers are constants)."""
        if self.norm is None:
            return None, None
        if self._ss is None or self._ss[0].device != self.norm.weight.device:
            n = self.norm
            scale = (n.weight.float() * (n.running_var.float() + FROZEN_BN_EPS).rsqrt()).contiguous()
            shift = (n.bias.float() - n.running_mean.float() * scale).contiguous()
            self._ss = (scale, shift)
        return self._ss

    def _load_from_state_dict(self, *a, **kw):
        self._ss = None
        return super()._load_from_state_dict(*a, **kw)


class BottleneckBlock(nn.Module):
    def __init__(self, cin, mid, cout, stride):
        super().__init__()
        self.shortcut = Conv2d(cin, cout, 1, stride) if cin != cout else None
        self.conv1 = Conv2d(cin, mid, 1, stride)          # stride in the 1x1 (STRIDE_IN_1X1=True)
        self.conv2 = Conv2d(mid, mid, 3, 1, 1)
        self.conv3 = Conv2d(mid, cout, 1)


class _Stem(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = Conv2d(3, 64, 7, 2, 3)


class _ResNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.stem = _Stem()
        cin = 64
        for name, n_blocks, mid, cout, stride in RESNET50_STAGES:
            blocks = []
            for b in range(n_blocks):
                blocks.append(BottleneckBlock(cin, mid, cout, stride if b == 0 else 1))
                cin = cout
            setattr(self, name, nn.ModuleList(blocks))


class _Detectron2Model(nn.Module):
    """Only ``backbone`` of the GeneralizedRCNN is ever executed by ClipBERT (grid_feat.py:95-97);
    the RPN / ROI heads of the reference checkpoint are ignored at load time."""
    def __init__(self):
        super().__init__()
        self.backbone = _ResNet()


class _GridConv(nn.Module):
    """grid_encoder[0]: conv3x3(2048 -> hidden, no bias)  (grid_feat.py:16-21,43-45)."""
    def __init__(self, cin, cout):
        super().__init__()
        self.cin, self.cout, self.k, self.stride, self.pad = cin, cout, 3, 1, 1
        w = torch.randn(cout, cin, 3, 3) * math.sqrt(2.0 / (cin * 9))
        self.weight = nn.Parameter(w.contiguous(memory_format=torch.channels_last))
        self.norm = None

    def scale_shift(self):
        return None, None


class _SideStream:
    """torch.cuda.stream(side) + the K-split scratch of the launches inside: the side stream's own (``ws``: partial products and
    arrival tickets belong to ONE stream, include/clipbert_hip.h) or none (the default scratch belongs to the main stream's launches)"""
    def __init__(self, stream, ws=None):
        self.ctx = torch.cuda.stream(stream)
        self.ws = ws

    def __enter__(self):
        self.prev = (ops._SPLITK_OFF, ops._SPLITK_SIDE)
        ops._SPLITK_OFF, ops._SPLITK_SIDE = self.ws is None, self.ws
        return self.ctx.__enter__()

    def __exit__(self, *exc):
        ops._SPLITK_OFF, ops._SPLITK_SIDE = self.prev
        return self.ctx.__exit__(*exc)


# =================================================================================================
# execution context shared by all modules of one ClipBert instance
# =================================================================================================
class Runtime:
    def __init__(self):
        self.bank: Optional[ParamBank] = None
        self.dtype = torch.bfloat16
        self.tables = {}
        self.rowmaps = {}
        self.seed_dev: Optional[torch.Tensor] = None     # device int64 added to every dropout seed
        self.anchor: Optional[torch.Tensor] = None       # requires_grad leaf that keeps the coarse nodes alive
        self.stem_w = None
        self.after_encoder_backward = None               # hook: launch the transformer-bucket all-reduce
        self.pending_encoder_nodes = 0                   # encoder autograd nodes created and not yet run backward: the hook
                                                         # fires when the LAST of them finished (multi-clip loops run several)
        self.after_res5_backward = None                  # hook: every gradient of grid_encoder + res5 is enqueued (fires inside the
                                                         # LAST ResNet backward of a step): their all-reduce can start while res4 / res3 run
        self.pending_cnn_nodes = 0
        self.prepare_args = None                         # keyword arguments of the prepare() call that built this runtime
        self._ln_off = None                              # (bank, offsets of the encoder LayerNorm gradients): cache of _ln_offsets
        self.forward_count = 0                           # host counter folded into every dropout seed: each forward (each
                                                         # clip of a clip loop) draws its own masks; kept in the saved pack
        self.side_stream = None                          # second HIP stream: weight-gradient GEMMs run beside the dgrad chain
        self.side_ws = None                              # its own K-split scratch (ops.new_splitk_workspace)
        self.overlap = 0                                 # what runs there (prepare(overlap_wgrad=...)): bit 0 the encoder's batched weight
                                                         # gradients beside the ResNet backward, bit 1 a ResNet stage's grouped weight
                                                         # gradients beside the next stage's data gradients, bit 2 every convolution's
                                                         # weight gradient on its own (the round-1 form), bit 3 the encoder's four batched
                                                         # weight-gradient launches on four concurrent branches (each one's last wave of
                                                         # tiles filled by the next one's first)
        self.fan_streams = []                            # bit 3: three more streams (the fourth branch is the issuing stream)
        self.group_wgrads = os.environ.get("CB_NO_GROUP_WGRAD") is None     # ResNet weight gradients per stage through cb_gemm_group
        self.group_enc_wgrads = os.environ.get("CB_NO_GROUP_ENC_WGRAD") is None   # the encoder's four batched weight-gradient kinds in one grouped launch
        self.group_fwd_pairs = os.environ.get("CB_GROUP_FWD_PAIRS", "0") == "1"   # shortcut + conv1 of the strided stage entries in one launch: measured SLOWER (profiles/r04f: +0.1 ms; the grouped gather kernel runs the pair in 106 us against 47 + 24 apart) -- kept as a switch
        self._side_refs = []

    def side(self, *tensors):
        """Context manager: run the enclosed launches on the side stream, ordered after everything issued so far on
        the current stream.  The weight-gradient GEMMs of the backward pass are independent of the data-gradient
        chain and each fills well under one block per CU, so the two streams overlap on the chip.  `tensors` are
        kept alive until join() so the caching allocator cannot hand their memory out while the side stream reads it."""
        if self.side_stream is None:
            return contextlib.nullcontext()
        ev = torch.cuda.Event()
        ev.record()
        self.side_stream.wait_event(ev)
        self._side_refs.extend(tensors)
        return _SideStream(self.side_stream, self.side_ws)

    def join(self):
        if self.side_stream is not None:
            torch.cuda.current_stream().wait_stream(self.side_stream)
            self._side_refs.clear()

    def table(self, n, oh, ow, stride, pad, sN, sH, sW, device):
        key = (n, oh, ow, stride, pad, sN, sH, sW, str(device))
        t = self.tables.get(key)
        if t is None:
            t = ops.build_pixel_table(n, oh, ow, stride, pad, sN, sH, sW, device)
            self.tables[key] = t
        return t

    def strided_rowmap(self, n, h, w, oh, ow, stride, device):
        key = (n, h, w, oh, ow, stride, str(device))
        t = self.rowmaps.get(key)
        if t is None:
            t = (torch.arange(n).view(n, 1, 1) * (h * w) + (torch.arange(oh) * stride).view(1, oh, 1) * w
                 + (torch.arange(ow) * stride).view(1, 1, ow)).reshape(-1).to(torch.int32).to(device)
            self.rowmaps[key] = t
        return t


def _pick_split(mo, no, kred):
    """(split_k, tile) for weight-gradient GEMMs (small outputs, long pixel/token reductions).  Measured on MI355X
    (tools/wgrad_probe2.py): one 64x64 block per CU is latency-bound (~0.3 us per 64-deep K tile), so long reductions
    are split until ~400 blocks are in flight; the partial sums combine through row-coalesced fp32 atomics.  Short
    reductions (transformer weights over 1312 tokens) lose more to the atomics than they gain."""
    ktiles = (kred + 63) // 64
    b64 = ((mo + 63) // 64) * ((no + 63) // 64)
    if b64 >= 200 or ktiles < 64:
        return 1, 0
    split = max(1, min(ktiles // 8, (400 + b64 // 2) // b64))
    return split, 0          # tile 0: cb_gemm's tuned table / heuristics choose the tile (and may refine the split)


# =================================================================================================
# CNN trunk: explicit forward / backward
# =================================================================================================
def _conv_fwd(rt: Runtime, x, conv, act=ACT_NONE, residual=None, relu_after=False, pending=None):
    """``pending`` (a list): the launch is only DESCRIBED and appended; the caller hands independent convolutions of one input to
    cb_gemm_group together (projection shortcut + conv1 of a stage-entry block)."""
    n, h, w, cin = x.shape
    k, s, p = conv.k, conv.stride, conv.pad
    oh, ow = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    cout = conv.cout
    m = n * oh * ow
    y = torch.empty(n, oh, ow, cout, dtype=x.dtype, device=x.device)
    wk = rt.bank.compute(conv.weight).view(cout, k * k * cin)
    scale, shift = conv.scale_shift()
    res2d = residual.view(m, cout) if residual is not None else None
    run = ops.gemm if pending is None else (lambda *a, **kw: pending.append(ops.gemm_desc(*a, **kw)))
    if k == 1 and s == 1:
        run(x.view(m, cin), wk, m, cout, cin, out=y.view(m, cout), scale=scale, shift=shift, act=act,
            residual=res2d, relu_after=relu_after)
    else:
        tab = rt.table(n, oh, ow, s, p, h * w * cin, w * cin, cin, x.device)
        run(x, wk, m, cout, k * k * cin, out=y.view(m, cout), a_mode=ROWK_GATHER, a_tab=tab, lda=0,
            ldb=k * k * cin, R=k, S=k, Cin=cin, H=h, W=w, sH=w * cin, sW=cin, scale=scale, shift=shift, act=act,
            residual=res2d, relu_after=relu_after)
    return y


def _conv_dgrad(rt: Runtime, g, conv, in_shape, scale=None, mask=None, residual=None, out=None, accumulate=False, fuse=None):
    """d(input) of a convolution given g = d(conv output) (already multiplied by the FrozenBN scale).
    Epilogue options: per-channel ``scale`` and ReLU ``mask`` of the PRODUCER of the input, ``residual``.
    ``fuse = (y, s_a, s_b)``: the input is the output y of a ResNet block; the launch also does that block's ReLU x
    FrozenBN-scale backward and returns (t*s_a, t*s_b) with t = d(input) where y > 0 (s_b None -> t itself)."""
    n, h, w, cin = in_shape
    _, oh, ow, cout = g.shape
    k, s, p = conv.k, conv.stride, conv.pad
    wk = rt.bank.compute(conv.weight).view(cout, k * k * cin)
    mi = n * h * w
    # stride-2 1x1 convolution: only the even pixels receive a gradient.  When the 2x2 patches tile the input exactly the launch
    # itself writes the zeros of the other three pixels (zero_fill_pitch); otherwise the output is pre-zeroed
    zfill = w if (k == 1 and s == 2 and h % 2 == 0 and w % 2 == 0 and cin % 8 == 0) else 0
    alloc = torch.zeros if (s > 1 and not zfill) else torch.empty
    if out is None:
        out = alloc(n, h, w, cin, dtype=g.dtype, device=g.device)
    o2 = out.view(mi, cin)
    r2 = residual.view(mi, cin) if residual is not None else None
    k2 = mask.view(mi, cin) if mask is not None else None
    extra = {}
    second = None
    if fuse is not None:
        assert mask is None and scale is None
        y, s_a, s_b = fuse
        second = alloc(n, h, w, cin, dtype=g.dtype, device=g.device)
        k2 = y.view(mi, cin)
        extra = dict(relu_bwd=True, post_scale=s_a, post_scale2=s_b, out2=second.view(mi, cin))
    if k == 1:
        rowmap = rt.strided_rowmap(n, h, w, oh, ow, s, g.device) if s > 1 else None
        ops.gemm(g.view(n * oh * ow, cout), wk, n * oh * ow, cin, cout, out=o2, b_mode=KROW_TAPS, ldb=cin, R=1, S=1,
                 Cin=cout, c_rowmap=rowmap, scale=scale, mask=k2, residual=r2, accumulate=accumulate, zero_fill_pitch=zfill if s > 1 else 0,
                 **extra)
    else:
        assert s == 1
        tab = rt.table(n, h, w, 1, k - 1 - p, oh * ow * cout, ow * cout, cout, g.device)
        ops.gemm(g, wk, mi, cin, k * k * cout, out=o2, a_mode=ROWK_GATHER, a_tab=tab, lda=0, b_mode=KROW_TAPS,
                 ldb=k * k * cin, R=k, S=k, Cin=cout, H=oh, W=ow, sH=ow * cout, sW=cout, flip_taps=True, scale=scale,
                 mask=k2, residual=r2, accumulate=accumulate, **extra)
    return (out, second) if fuse is not None else out


def _conv_wgrad(rt: Runtime, g, x, conv, pending=None):
    """dW[co][(r,s,c)] += sum_pixels g[m,co] * x[pix(m,r,s), c], straight into the flat fp32 grad buffer.
    ``pending`` (a list): the launch is only DESCRIBED and appended -- the caller hands the weight gradients of a whole ResNet stage
    to cb_gemm_group at once (they are off the data-gradient chain: a few launches that fill the chip instead of one per convolution)."""
    gw = rt.bank.grad_image(conv.weight)
    if gw is None:
        return
    n, h, w, cin = x.shape
    _, oh, ow, cout = g.shape
    k, s, p = conv.k, conv.stride, conv.pad
    m = n * oh * ow
    kk = k * k * cin
    split, tile = _pick_split(cout, kk, m)
    # first writer of this step (ParamBank.set_fresh_params: the range was not zeroed): the launch STORES (no read-modify-write of the
    # gradient) and leaves its share of the squared norm; a second backward of the step accumulates as before and voids the shares
    bank = rt.bank
    acc, slots = True, None
    if bank.take_fresh_param(conv.weight):
        if rt.dtype == torch.bfloat16 and kk % 8 == 0:
            acc = 2
            slots = bank.fold_take(ops.sq_slot_count(cout, kk), "cnn")
        else:
            ops.zero_(gw)                            # (a form the first-writer store does not cover: zero now, accumulate as ever)
    else:
        bank.fold_invalidate()
    # (descriptors that will be LAUNCHED on the side stream carry its scratch, whichever stream describes them)
    side_ws = rt.side_ws if (pending is not None and rt.overlap & 2) else None
    run = ops.gemm if pending is None else (lambda *a, **kw: pending.append(ops.gemm_desc(*a, splitk_ws=side_ws, **kw)))
    if k == 1 and s == 1:
        run(g.view(m, cout), x.view(m, cin), cout, cin, m, out=gw.view(cout, kk), a_mode=KROW, lda=cout,
            b_mode=KROW, ldb=cin, accumulate=acc, split_k=split, tile=tile, sq_slots=slots)
    else:
        tab = rt.table(n, oh, ow, s, p, h * w * cin, w * cin, cin, x.device)
        run(g.view(m, cout), x, cout, kk, m, out=gw.view(cout, kk), a_mode=KROW, lda=cout, b_mode=KROW_GATHER,
            b_tab=tab, ldb=0, R=k, S=k, Cin=cin, H=h, W=w, sH=w * cin, sW=cin, accumulate=acc, split_k=split,
            tile=tile, sq_slots=slots)


def _stem_weight(rt: Runtime, conv: Conv2d):
    """[64][7 rows][8 taps x 4 ch] image of the 7x7x3 stem filter (tap 7 and channel 3 are zero)."""
    if rt.stem_w is None:
        w = conv.weight.detach().float()                         # (64, 3, 7, 7)
        wp = torch.zeros(64, 7, 8, 4, dtype=torch.float32, device=w.device)
        wp[:, :, :7, :3] = w.permute(0, 2, 3, 1)
        rt.stem_w = wp.view(64, 224).to(rt.dtype).contiguous()
    return rt.stem_w


def _block_trainable(rt: Runtime, blk: BottleneckBlock) -> bool:
    return any(rt.bank.is_trainable(c.weight) for c in (blk.conv1, blk.conv2, blk.conv3) + ((blk.shortcut,) if blk.shortcut is not None else ()))


def _res2_block_fused(rt: Runtime, x, blk: BottleneckBlock):
    """one res2 block through cb_res2_block (conv1 -> conv2 -> conv3 + shortcut in ONE launch; forward only)"""
    w = lambda conv: rt.bank.compute(conv.weight)
    sc = blk.shortcut
    return ops.res2_block(x, w(blk.conv1), w(blk.conv2), w(blk.conv3), blk.conv1.scale_shift(), blk.conv2.scale_shift(), blk.conv3.scale_shift(),
                          wsc=w(sc) if sc is not None else None, sssc=sc.scale_shift() if sc is not None else None)


def _res2_fusable(rt: Runtime, x, blk: BottleneckBlock, save: bool) -> bool:
    """the fused forward kernel covers the frozen 64-mid-channel, stride-1 blocks in bf16 (FREEZE_AT = 2: nothing of res2 is saved for a
    backward); CB_NO_RES2_FUSE=1 restores the three / four cb_gemm launches"""
    if rt.dtype != torch.bfloat16 or os.environ.get("CB_NO_RES2_FUSE") is not None:
        return False
    if save and _block_trainable(rt, blk):
        return False
    c1, c2, c3 = blk.conv1, blk.conv2, blk.conv3
    return (c1.cout == 64 and c2.cin == 64 and c2.cout == 64 and c3.cout == 256 and c1.stride == 1 and c1.k == 1 and c2.k == 3 and c2.stride == 1
            and c3.k == 1 and c1.cin in (64, 256) and (blk.shortcut is not None) == (c1.cin == 64) and x.shape[-1] == c1.cin and x.is_contiguous())


def cnn_forward(bb: "GridFeatBackbone", x5: torch.Tensor, save: bool):
    """(B,T,3,H,W) fp32 RGB mean-subtracted (or uint8 RGB) -> grid (B,T,H',W',hidden) + saved activations."""
    rt = bb.rt
    b, t, c, h, w = x5.shape
    n = b * t
    x4 = x5.reshape(n, c, h, w)
    if not x4.is_contiguous():
        x4 = x4.contiguous()
    net = bb.feature.backbone
    stem = net.stem.conv1
    oh, ow = (h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1
    m = n * oh * ow
    scale, shift = stem.scale_shift()
    fuse_stem = rt.dtype == torch.bfloat16 and w % 2 == 0 and os.environ.get("CB_NO_STEM_FUSE") is None
    packed = None
    if x4.dtype == torch.uint8 and fuse_stem and os.environ.get("CB_NO_STEM_U8") is None:
        # uint8 frames straight into the first convolution: ImageNorm, BGR flip and padding inside cb_stem_pool's tile loader (round 6, N4)
        x = ops.stem_pool_u8(x4, bb.pixel_mean, bb.pixel_std, _stem_weight(rt, stem), scale, shift)
    elif x4.dtype == torch.uint8:
        packed = ops.stem_pack(x4, rt.dtype, 3, bb.pixel_mean, bb.pixel_std, extra_w=2)
    else:
        packed = ops.stem_pack(x4.float(), rt.dtype, 3, extra_w=2)
    hp, wp = (packed.shape[1], packed.shape[2]) if packed is not None else (0, 0)
    if packed is None:
        pass
    elif fuse_stem:
        # convolution + FrozenBN + ReLU + max-pool in one launch (the 112 x 112 x 64 map never leaves the CU); CB_NO_STEM_FUSE=1: two launches
        x = ops.stem_pool(packed, _stem_weight(rt, stem), scale, shift, oh, ow)
    else:
        tab = rt.table(n, oh, ow, 2, 0, hp * wp * 4, wp * 4, 4, x5.device)
        y = torch.empty(n, oh, ow, 64, dtype=rt.dtype, device=x5.device)
        ops.gemm(packed, _stem_weight(rt, stem), m, 64, 224, out=y.view(m, 64), a_mode=ROWK_GATHER, a_tab=tab, lda=0, ldb=224,
                 R=7, S=1, Cin=32, H=hp, W=wp, sH=wp * 4, sW=4, scale=scale, shift=shift, act=ACT_RELU)
        x = ops.maxpool_fwd(y, 3, 2, 1)
    saved = []
    for name, _nb, _mid, _cout, _s in RESNET50_STAGES:
        for blk in getattr(net, name):
            if _res2_fusable(rt, x, blk, save):
                x = _res2_block_fused(rt, x, blk)
                continue
            if blk.shortcut is not None and blk.conv1.stride > 1 and rt.group_fwd_pairs:
                # stage entry with a stride: projection shortcut and conv1 read the same strided pixels -- one grouped launch
                # (the stride-1 entry of res2 stays two launches: its shortcut is a streaming-kernel shape, cb_gemm tile 8)
                pair = []
                sc = _conv_fwd(rt, x, blk.shortcut, pending=pair)
                y1 = _conv_fwd(rt, x, blk.conv1, act=ACT_RELU, pending=pair)
                ops.gemm_group(pair, x)
            else:
                sc = _conv_fwd(rt, x, blk.shortcut) if blk.shortcut is not None else x
                y1 = _conv_fwd(rt, x, blk.conv1, act=ACT_RELU)
            y2 = _conv_fwd(rt, y1, blk.conv2, act=ACT_RELU)
            out = _conv_fwd(rt, y2, blk.conv3, residual=sc, relu_after=True)
            if save and _block_trainable(rt, blk):
                saved.append((blk, x, y1, y2, out))
            x = out
    gconv = bb.grid_encoder[0]
    gy = _conv_fwd(rt, x, gconv)
    grid = ops.maxpool_fwd(gy, 2, 2, 0, relu=True)
    hg, wg = grid.shape[1], grid.shape[2]
    return grid.view(b, t, hg, wg, gconv.cout), (saved, x, gy, grid) if save else None


def cnn_backward(bb: "GridFeatBackbone", saved_pack, dgrid: torch.Tensor):
    """the whole ResNet backward; rt.after_res5_backward (if set) is called at the point cnn_backward_steps yields, when this is the
    last pending ResNet backward of the step"""
    rt = bb.rt
    for _ in cnn_backward_steps(bb, saved_pack, dgrid):
        hook = rt.after_res5_backward
        if hook is not None and rt.pending_cnn_nodes <= 1:
            hook()


def cnn_backward_steps(bb: "GridFeatBackbone", saved_pack, dgrid: torch.Tensor):
    """Generator over the ResNet backward.  Yields ONCE, when every launch that writes a gradient of grid_encoder or res5 (the tail
    of the CNN range of the flat gradient buffer, and ~3/4 of its bytes) has been enqueued and earlier stages remain: a
    data-parallel caller starts that part of the exchange there (hook above, or between two captured graphs: bench.py).  Exhausting
    it without looking at the yield is the plain backward."""
    rt = bb.rt
    saved, res5, gy, grid = saved_pack
    gconv = bb.grid_encoder[0]
    dg = ops.maxpool2_bwd(gy, grid, dgrid.reshape(grid.shape).contiguous(), relu=True)
    with (rt.side(dg, res5) if rt.overlap & 6 else contextlib.nullcontext()):
        _conv_wgrad(rt, dg, res5, gconv)
    if not saved:
        rt.join()
        return
    res5_ids = {id(b) for b in bb.feature.backbone.res5}
    first_res5 = min((i for i, rec in enumerate(saved) if id(rec[0]) in res5_ids), default=None)
    # weight gradients of a stage's convolutions: described as the data-gradient chain passes them, launched together when the chain
    # leaves the stage (cb_gemm_group; not with the side-stream variant, which overlaps them one by one)
    stage_of = {id(b): name for name, *_ in RESNET50_STAGES for b in getattr(bb.feature.backbone, name)}
    pend = [] if (rt.group_wgrads and not rt.overlap & 4) else None
    conv_side = rt.side if pend is None else (lambda *t: contextlib.nullcontext())     # (pending: only described here, launched by flush)

    def flush():
        if pend:
            if rt.overlap & 2:                          # the stage's grouped launches beside the next stage's data gradients
                with rt.side(*pend):                    # (the descriptors keep their operands alive until join())
                    ops.gemm_group(pend, dg)
            else:
                ops.gemm_group(pend, dg)
            pend.clear()

    def fuse_spec(i):
        """the ReLU x FrozenBN-scale backward of block i, done by the launch that produces d(output of block i)"""
        b, _x, _y1, _y2, o = saved[i]
        s3_, _ = b.conv3.scale_shift()
        ssc_ = b.shortcut.scale_shift()[0] if b.shortcut is not None else None
        return (o, s3_, ssc_)

    # (g3, sec): g3 = d(conv3 output of the block), sec = d(identity shortcut) or d(shortcut conv output)
    g3, sec = _conv_dgrad(rt, dg, gconv, res5.shape, fuse=fuse_spec(len(saved) - 1))
    for idx in range(len(saved) - 1, -1, -1):
        blk, x, y1, y2, out = saved[idx]
        need_dx = idx > 0
        s2, _ = blk.conv2.scale_shift()
        s1, _ = blk.conv1.scale_shift()
        dz, gsc = (sec, None) if blk.shortcut is None else (None, sec)
        with conv_side(g3, y2, sec):
            _conv_wgrad(rt, g3, y2, blk.conv3, pend)
            if blk.shortcut is not None:
                _conv_wgrad(rt, gsc, x, blk.shortcut, pend)
        g2 = _conv_dgrad(rt, g3, blk.conv3, y2.shape, scale=s2, mask=y2)       # -> d(conv2 out) * mask * scale2
        with conv_side(g2, y1):
            _conv_wgrad(rt, g2, y1, blk.conv2, pend)
        g1 = _conv_dgrad(rt, g2, blk.conv2, y1.shape, scale=s1, mask=y1)
        with conv_side(g1, x):
            _conv_wgrad(rt, g1, x, blk.conv1, pend)
        if idx == 0 or stage_of.get(id(saved[idx - 1][0])) != stage_of.get(id(blk)):
            flush()                                     # the chain leaves this stage: its weight gradients in a few grouped launches
        if idx == first_res5 and idx > 0:
            rt.join()
            yield "grid_encoder+res5"
        if need_dx:
            spec = fuse_spec(idx - 1)
            if blk.shortcut is None:
                g3, sec = _conv_dgrad(rt, g1, blk.conv1, x.shape, residual=dz, fuse=spec)
            else:
                dout = _conv_dgrad(rt, g1, blk.conv1, x.shape)
                g3, sec = _conv_dgrad(rt, gsc, blk.shortcut, x.shape, out=dout, accumulate=True, fuse=spec)
    rt.join()


class _CnnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, x5, bb):
        save = ctx.needs_input_grad[0] and bb.has_trainable()    # anchor: True iff autograd is recording
        grid, pack = cnn_forward(bb, x5, save)
        ctx.bb, ctx.pack = bb, pack
        if pack is not None:
            bb.rt.pending_cnn_nodes += 1
        return grid

    @staticmethod
    def backward(ctx, dgrid):
        if ctx.pack is not None:
            cnn_backward(ctx.bb, ctx.pack, dgrid.contiguous())
            ctx.pack = None
            ctx.bb.rt.pending_cnn_nodes = max(0, ctx.bb.rt.pending_cnn_nodes - 1)
        return None, None, None


class GridFeatBackbone(nn.Module):
    """ResNet-50 grid-feature backbone + grid encoder (src/modeling/grid_feat.py:37-105)."""
    def __init__(self, detectron2_model_cfg=None, config=None, input_format="BGR", freeze_at=2):
        super().__init__()
        assert input_format == "BGR", "detectron 2 image input format should be BGR"
        config = as_config(config)
        self.detectron2_model_cfg = detectron2_model_cfg
        self.feature = _Detectron2Model()
        self.grid_encoder = nn.ModuleList([_GridConv(config.backbone_channel_in_size, config.hidden_size)])
        self.input_format = input_format
        self.config = config
        self.pixel_mean = (123.675, 116.28, 103.53)
        self.pixel_std = (1.0, 1.0, 1.0)
        self.rt: Optional[Runtime] = None
        # detectron2 FREEZE_AT=2: stem and res2 never receive gradients
        net = self.feature.backbone
        frozen = [net.stem] + ([net.res2] if freeze_at >= 2 else [])
        for mod in frozen:
            for p in mod.parameters():
                p.requires_grad = False

    @property
    def config_file(self):
        return f"clipbert_amd R-50 grid backbone (detectron2 cfg: {self.detectron2_model_cfg})"

    def has_trainable(self):
        return any(p.requires_grad for p in self.parameters())

    def forward(self, x):
        """x: (B, n_frm, 3, H, W) RGB float (mean-subtracted) or uint8 -> (B, n_frm, H', W', hidden)."""
        return _CnnFn.apply(self.rt.anchor, x, self)


# =================================================================================================
# cross-modal BERT
# =================================================================================================
class BertEmbeddings(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.word_embeddings = Embedding(config.vocab_size, config.hidden_size, config.initializer_range,
                                         padding_idx=_cfg_get(config, "pad_token_id", 0))
        self.position_embeddings = Embedding(config.max_position_embeddings, config.hidden_size, config.initializer_range)
        self.token_type_embeddings = Embedding(config.type_vocab_size, config.hidden_size, config.initializer_range)
        self.LayerNorm = LayerNorm(config.hidden_size, config.layer_norm_eps)


class VisualInputEmbedding(nn.Module):
    def __init__(self, config):
        super().__init__()
        r = config.initializer_range
        self.position_embeddings = Embedding(config.max_position_embeddings, config.hidden_size, r)   # unused (as in the reference)
        self.row_position_embeddings = Embedding(config.max_grid_row_position_embeddings, config.hidden_size, r)
        self.col_position_embeddings = Embedding(config.max_grid_col_position_embeddings, config.hidden_size, r)
        self.token_type_embeddings = Embedding(1, config.hidden_size, r)
        self.LayerNorm = LayerNorm(config.hidden_size, config.layer_norm_eps)


class _SelfAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        d = config.hidden_size
        self.query, self.key, self.value = Linear(d, d), Linear(d, d), Linear(d, d)


class _SelfOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = LayerNorm(config.hidden_size, config.layer_norm_eps)


class _Attention(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.self = _SelfAttention(config)
        self.output = _SelfOutput(config)


class _Intermediate(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = Linear(config.hidden_size, config.intermediate_size)


class _Output(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = LayerNorm(config.hidden_size, config.layer_norm_eps)


class BertLayer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.attention = _Attention(config)
        self.intermediate = _Intermediate(config)
        self.output = _Output(config)


class BertEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.layer = nn.ModuleList([BertLayer(config) for _ in range(config.num_hidden_layers)])


class BertPooler(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = Linear(config.hidden_size, config.hidden_size)


# dropout sites -> distinct seed streams
_SITE_EMB, _SITE_ATTN, _SITE_SELF_OUT, _SITE_OUT, _SITE_POOL = 1, 2, 3, 4, 5


def _seed(site, layer=0, fwd=0):
    """seed of one dropout site of one layer of the fwd-th forward of this process (the device word *rt.seed_dev is added
    on top by the kernels, so hipGraph replays -- where `fwd` is frozen at capture -- still draw fresh masks)"""
    return ((site * 1000003 + layer * 7919) * 2654435761 + fwd * 0x9E3779B97F4A7C15) % (1 << 62)


class ClipBertBaseModel(nn.Module):
    """Embeddings + 12-layer encoder + pooler (src/modeling/modeling.py:156-238)."""
    def __init__(self, config):
        super().__init__()
        config = as_config(config)
        assert _cfg_get(config, "hidden_act", "gelu") == "gelu", "only the exact-erf GELU of base_model.json is implemented"
        assert config.hidden_size // config.num_attention_heads == 64, "attention kernels are built for head size 64"
        self.config = config
        self.embeddings = BertEmbeddings(config)
        self.visual_embeddings = VisualInputEmbedding(config)
        self.encoder = BertEncoder(config)
        self.pooler = BertPooler(config)
        self.rt: Optional[Runtime] = None

    def get_input_embeddings(self):
        return self.embeddings.word_embeddings

    def forward(self, text_input_ids, visual_inputs, attention_mask, src_row=None, pooled_dropout=False, text_repeat=1):
        """visual_inputs: grid (Bv, n_frm, H', W', d); src_row maps each text row to its grid row
        (the fused form of repeat_tensor_rows).  Returns (sequence_output (B, L, d), pooled (B, d)).
        text_repeat = n: the (P, Lt) text batch stands for n*P rows (row b = text row b % P: the captions of a folded clip loop,
        read in place by the embedding kernels instead of from n repeated copies)."""
        seq, pooled = _EncoderFn.apply(self.rt.anchor, visual_inputs, self, text_input_ids, attention_mask, src_row,
                                       pooled_dropout, text_repeat)
        b = text_input_ids.shape[0] * text_repeat
        return seq.view(b, -1, self.config.hidden_size), pooled


def _drop_p(model, training, key="hidden_dropout_prob"):
    return float(_cfg_get(model.config, key, 0.0)) if training else 0.0


def encoder_forward(model: ClipBertBaseModel, grid, ids, mask, src_row, pooled_dropout, save, text_repeat=1):
    rt, cfg = model.rt, model.config
    bank, dt, dev = rt.bank, rt.dtype, ids.device
    d, nh, eps = cfg.hidden_size, cfg.num_attention_heads, cfg.layer_norm_eps
    training = model.training
    p_h = _drop_p(model, training)
    p_a = _drop_p(model, training, "attention_probs_dropout_prob")
    fwd_i = 0
    if p_h > 0 or p_a > 0:                    # every training forward draws its own dropout masks (the reference's
        fwd_i = rt.forward_count              # nn.Dropout does); the backward regenerates them from pack.fwd_i
        rt.forward_count += 1
    bsz, lt = ids.shape[0] * text_repeat, ids.shape[1]
    bv, t, hg, wg, _ = grid.shape
    # optional random pixel sub-sampling: training phase of pre-training only (modeling.py:80-88)
    sel = None
    lv = hg * wg
    nsamp = int(_cfg_get(cfg, "pixel_random_sampling_size", 0) or 0)
    if nsamp > 0 and training and nsamp < lv:
        idx = np.sort(np.random.choice(lv, size=nsamp, replace=False))     # numpy global RNG, as the reference
        sel = torch.from_numpy(idx.astype(np.int32)).to(dev)
        lv = nsamp
    L = lt + lv
    M = bsz * L
    if src_row is None:
        assert bv == bsz, "visual batch and text batch differ: pass src_row (n_examples_list)"
    key_mask = torch.empty(bsz, L, dtype=torch.float32, device=dev)        # filled by the two embedding kernels
    mask_c = mask if (mask.dtype == torch.int64 and mask.is_contiguous()) else mask.to(torch.int64).contiguous()
    nl, ff = len(model.encoder.layer), cfg.intermediate_size
    # The operands of the weight-gradient GEMMs are kept LAYER-STACKED ([layer, M, *]): the backward then computes the
    # weight gradients of all layers of one kind in a single strided-batched launch (encoder_backward).
    stk = None
    if save:
        stk = SimpleNamespace(x=torch.empty(nl, M, d, dtype=dt, device=dev), ctx=torch.empty(nl, M, d, dtype=dt, device=dev),
                              a=torch.empty(nl, M, d, dtype=dt, device=dev), hact=torch.empty(nl, M, ff, dtype=dt, device=dev))
    x = stk.x[0] if save else torch.empty(M, d, dtype=dt, device=dev)
    pre = torch.empty(M, d, dtype=dt, device=dev) if save else None
    mean0 = torch.empty(M, dtype=torch.float32, device=dev) if save else None
    rstd0 = torch.empty(M, dtype=torch.float32, device=dev) if save else None
    emb, vemb = model.embeddings, model.visual_embeddings
    ids_c = ids.contiguous()
    ops.text_embed_fwd(ids_c, bank.compute(emb.word_embeddings.weight), bank.compute(emb.position_embeddings.weight),
                       bank.compute(emb.token_type_embeddings.weight)[0], emb.LayerNorm.weight, emb.LayerNorm.bias, x, pre,
                       mean0, rstd0, lt, L, eps, attn_mask=mask_c, key_mask=key_mask, repeat=text_repeat)
    grid_c = grid.contiguous()
    ops.visual_embed_fwd(grid_c, src_row, sel, bank.compute(vemb.row_position_embeddings.weight),
                         bank.compute(vemb.col_position_embeddings.weight), bank.compute(vemb.token_type_embeddings.weight)[0],
                         vemb.LayerNorm.weight, vemb.LayerNorm.bias, x, pre, mean0, rstd0, bsz, lv, lt, L, eps, key_mask=key_mask)
    if p_h > 0:
        ops.dropout(x, p_h, _seed(_SITE_EMB, 0, fwd_i), rt.seed_dev, out=x)
    layers = []
    for li, layer in enumerate(model.encoder.layer):
        att, so, it, ou = layer.attention.self, layer.attention.output, layer.intermediate, layer.output
        wqkv = bank.compute_span(att.query.weight, att.value.weight, (3 * d, d))
        bqkv = bank.master_span(att.query.bias, att.value.bias, (3 * d,))
        qkv = torch.empty(M, 3 * d, dtype=dt, device=dev)
        ops.gemm(x, wqkv, M, 3 * d, d, out=qkv, shift=bqkv)
        ctx, lse = ops.attention_fwd(qkv, key_mask, bsz, L, nh, save_lse=save, dropout_p=p_a,
                                     dropout_seed=_seed(_SITE_ATTN, li, fwd_i), seed_ptr=rt.seed_dev, out=stk.ctx[li] if save else None)
        a_pre = torch.empty(M, d, dtype=dt, device=dev)
        ops.gemm(ctx, bank.compute(so.dense.weight), M, d, d, out=a_pre, shift=so.dense.bias, residual=x, dropout_p=p_h,
                 dropout_seed=_seed(_SITE_SELF_OUT, li, fwd_i), seed_ptr=rt.seed_dev)
        a, mean1, rstd1 = ops.layernorm_fwd(a_pre, so.LayerNorm.weight, so.LayerNorm.bias, eps, save_stats=save,
                                            out=stk.a[li] if save else None)
        hact = stk.hact[li] if save else torch.empty(M, ff, dtype=dt, device=dev)
        hsave = torch.empty(M, ff, dtype=dt, device=dev) if save else None     # gelu'(pre) (pack.gelu_saved_grad) or the pre-activation
        # (training: the second output is gelu'(pre-activation), all the backward needs of it -- one evaluation of exp / erfc for both, and
        # the FFN2 data-gradient epilogue multiplies by the stored value instead of evaluating the derivative: CB_ACT_GELU_SAVE_GRAD)
        ops.gemm(a, bank.compute(it.dense.weight), M, ff, d, out=hact, shift=it.dense.bias,
                 act=ops.ACT_GELU_SAVE_GRAD if (save and _GELU_SAVE_GRAD) else ACT_GELU, out2=hsave)
        o_pre = torch.empty(M, d, dtype=dt, device=dev)
        ops.gemm(hact, bank.compute(ou.dense.weight), M, d, ff, out=o_pre, shift=ou.dense.bias, residual=a, dropout_p=p_h,
                 dropout_seed=_seed(_SITE_OUT, li, fwd_i), seed_ptr=rt.seed_dev)
        out, mean2, rstd2 = ops.layernorm_fwd(o_pre, ou.LayerNorm.weight, ou.LayerNorm.bias, eps, save_stats=save,
                                              out=stk.x[li + 1] if (save and li + 1 < nl) else None)
        if save:
            layers.append((x, qkv, ctx, lse, a_pre, mean1, rstd1, a, hsave, hact, o_pre, mean2, rstd2))
        x = out
    pooled = torch.empty(bsz, d, dtype=dt, device=dev)
    p_pool = p_h if pooled_dropout else 0.0
    pooled_raw = torch.empty(bsz, d, dtype=dt, device=dev) if (save and p_pool > 0) else None
    pw = model.pooler.dense
    if pooled_raw is not None:
        ops.gemm(x, bank.compute(pw.weight), bsz, d, d, out=pooled_raw, lda=L * d, shift=pw.bias, act=ACT_TANH)
        ops.dropout(pooled_raw, p_pool, _seed(_SITE_POOL, 0, fwd_i), rt.seed_dev, out=pooled)
    else:
        ops.gemm(x, bank.compute(pw.weight), bsz, d, d, out=pooled, lda=L * d, shift=pw.bias, act=ACT_TANH,
                 dropout_p=p_pool, dropout_seed=_seed(_SITE_POOL, 0, fwd_i), seed_ptr=rt.seed_dev)
    pack = None
    if save:
        pack = SimpleNamespace(layers=layers, x_final=x, pooled=pooled, pooled_raw=pooled_raw, p_pool=p_pool, pre=pre,
                               mean0=mean0, rstd0=rstd0, ids=ids_c, key_mask=key_mask, src_row=src_row, sel=sel, bsz=bsz,
                               lt=lt, lv=lv, L=L, grid_shape=tuple(grid.shape), p_h=p_h, p_a=p_a, stk=stk, fwd_i=fwd_i, text_repeat=text_repeat,
                               gelu_saved_grad=bool(_GELU_SAVE_GRAD))
    return x, pooled, pack


def _linear_wgrad(rt: Runtime, g, x, lin_weight, lin_bias, m, n, k, ldx=None, grad_w=None, grad_b=None):
    """dW[n,k] += g[m,n]^T x[m,k];  db[n] += colsum(g)."""
    gw = grad_w if grad_w is not None else (rt.bank.grad_image(lin_weight) if lin_weight is not None else None)
    gb = grad_b if grad_b is not None else (rt.bank.grad_image(lin_bias) if lin_bias is not None else None)
    if gw is not None:
        split, tile = _pick_split(n, k, m)
        ops.gemm(g, x, n, k, m, out=gw, a_mode=KROW, lda=g.stride(0), b_mode=KROW, ldb=ldx if ldx is not None else x.stride(0),
                 accumulate=True, split_k=split, tile=tile, a_rowsum=gb)       # db rides on the same kernel
    elif gb is not None:
        ops.colsum(g, gb, m, n)


def _uniform_stride(tensors):
    """element stride between consecutive tensors of a list if they are equally spaced views of one buffer, else None"""
    if any(t is None for t in tensors):
        return None
    if len(tensors) == 1:
        return 0
    esz = tensors[0].element_size()
    step = tensors[1].data_ptr() - tensors[0].data_ptr()
    if step <= 0 or step % esz:
        return None
    for i in range(2, len(tensors)):
        if tensors[i].data_ptr() - tensors[i - 1].data_ptr() != step:
            return None
    return step // esz


def _encoder_wgrads(model, pk, gs, M):
    """Weight and bias gradients of every encoder layer: one strided-batched weight-gradient GEMM per kind when the
    layers' gradient images are equally spaced in the flat gradient buffer (they are: same parameter order in every
    layer), per-layer launches otherwise (frozen layers)."""
    rt, cfg = model.rt, model.config
    bank = rt.bank
    d, ff = cfg.hidden_size, cfg.intermediate_size
    layers = list(model.encoder.layer)
    nl, stk = len(layers), pk.stk

    def qkv_w(l):
        return bank.grad_span(l.attention.self.query.weight, l.attention.self.value.weight, (3 * d, d)) \
            if bank.is_trainable(l.attention.self.query.weight) else None

    def qkv_b(l):
        return bank.grad_span(l.attention.self.query.bias, l.attention.self.value.bias, (3 * d,)) \
            if bank.is_trainable(l.attention.self.query.bias) else None

    kinds = [  # (upstream gradient stack, input stack, out features, in features, dW images, db images, name of the kind)
        (gs.out, stk.hact, d, ff, [bank.grad_image(l.output.dense.weight) for l in layers], [bank.grad_image(l.output.dense.bias) for l in layers], "out"),
        (gs.hp, stk.a, ff, d, [bank.grad_image(l.intermediate.dense.weight) for l in layers], [bank.grad_image(l.intermediate.dense.bias) for l in layers], "ffn"),
        (gs.att, stk.ctx, d, d, [bank.grad_image(l.attention.output.dense.weight) for l in layers], [bank.grad_image(l.attention.output.dense.bias) for l in layers], "att"),
        (gs.qkv, stk.x, 3 * d, d, [qkv_w(l) for l in layers], [qkv_b(l) for l in layers], "qkv"),
    ]
    # first-writer stores: after zero_grad(lazy=True) the encoder weight gradients were NOT zeroed -- the batched launches
    # overwrite them (no memset, no fp32 read-modify-write); any other path zeroes the span first
    fresh = bank.take_fresh()
    batched = [(_uniform_stride(gws), _uniform_stride(gbs)) for _g, _x, _n, _k, gws, gbs, _kind in kinds]
    all_batched = all(sw is not None and sb is not None and n % 8 == 0 and k % 8 == 0 for (sw, sb), (_g, _x, n, k, _w, _b, _kd) in zip(batched, kinds))
    if fresh and not all_batched:
        a, b = bank.lazy_span
        bank.grad[a:b].zero_()
        fresh = False
    if not fresh:
        bank.fold_invalidate()                       # (a second backward of the step accumulates: the first one's norm shares are void)
    if all_batched and rt.group_enc_wgrads and rt.dtype == torch.bfloat16:
        # all four kinds in ONE grouped launch (cb_gemm_group's row-sum / strided-batch class): 48 problems' 5184 tiles of 128x128 share a
        # grid, so only one last wave of tiles runs on a part-filled chip instead of four (profiles/r06k_enc_wgrad_group_ab.txt)
        descs = []
        for g, x, n, k, gws, gbs, kind in kinds:
            slots = bank.fold_take(ops.sq_slot_count(n, k, nl), "enc:" + kind) if fresh else None
            descs.append(ops.gemm_desc(g, x, n, k, M, out=gws[0], a_mode=KROW, lda=n, b_mode=KROW, ldb=k, ldc=k, accumulate=not fresh, a_rowsum=gbs[0],
                                       batch=nl, batch_strides=(M * n, M * k, _uniform_stride(gws), _uniform_stride(gbs)), sq_slots=slots, tile=4))
        ops.gemm_group(descs, gs.out)
        return
    fan = rt.fan_streams if (rt.overlap & 8 and all_batched) else []
    if fan:
        fork = torch.cuda.Event()
        fork.record()
    for i, (g, x, n, k, gws, gbs, kind) in enumerate(kinds):
        sw, sb = _uniform_stride(gws), _uniform_stride(gbs)
        if sw is not None and sb is not None and n % 8 == 0 and k % 8 == 0:
            slots = bank.fold_take(ops.sq_slot_count(n, k, nl), "enc:" + kind) if (fresh and rt.dtype == torch.bfloat16) else None
            branch = contextlib.nullcontext()
            if fan and i > 0:
                fan[i - 1].wait_event(fork)
                branch = _SideStream(fan[i - 1])            # (no K split on a branch: the scratch belongs to the issuing stream)
            with branch:
                ops.gemm(g, x, n, k, M, out=gws[0], a_mode=KROW, lda=n, b_mode=KROW, ldb=k, ldc=k, accumulate=not fresh, a_rowsum=gbs[0],
                         batch=nl, batch_strides=(M * n, M * k, sw, sb), sq_slots=slots)
        else:
            for li in range(nl):
                _linear_wgrad(rt, g[li], x[li], None, None, M, n, k, grad_w=gws[li], grad_b=gbs[li])
    for st in fan:
        torch.cuda.current_stream().wait_stream(st)


def _ln_offsets(model, dev):
    """(2, 2*n_layers) int64 device tensor: element offsets of the encoder LayerNorms' (weight | bias) gradients in bank.grad, in
    the slot order encoder_backward uses (2*l: attention.output.LayerNorm, 2*l+1: output.LayerNorm); None if any is frozen."""
    rt = model.rt
    cached = rt._ln_off
    if cached is not None and cached[0] is rt.bank:
        return cached[1]
    bank, offs = rt.bank, ([], [])
    for layer in model.encoder.layer:
        for ln in (layer.attention.output.LayerNorm, layer.output.LayerNorm):
            gw, gb = bank.grad_image(ln.weight), bank.grad_image(ln.bias)
            if gw is None or gb is None:
                rt._ln_off = (bank, None)
                return None
            offs[0].append((gw.data_ptr() - bank.grad.data_ptr()) // 4)
            offs[1].append((gb.data_ptr() - bank.grad.data_ptr()) // 4)
    t = torch.tensor(offs, dtype=torch.int64).to(dev)
    rt._ln_off = (bank, t)
    return t


def encoder_backward(model: ClipBertBaseModel, pk, d_seq, d_pooled):
    rt, cfg = model.rt, model.config
    bank, dt = rt.bank, rt.dtype
    d, nh, ff = cfg.hidden_size, cfg.num_attention_heads, cfg.intermediate_size
    bsz, L, lt, lv = pk.bsz, pk.L, pk.lt, pk.lv
    M = bsz * L
    dev = pk.x_final.device
    # ---- pooler ------------------------------------------------------------------------------------
    if d_seq is None:
        dx = ops.zeros((M, d), dt, dev)
    else:
        dx = d_seq.reshape(M, d)
        if dx.dtype != dt or not dx.is_contiguous():
            dx = ops.cast(dx.contiguous(), torch.empty(M, d, dtype=dt, device=dev))
    if d_pooled is not None:
        g = d_pooled.to(dt).contiguous()
        if pk.pooled_raw is not None:
            g = ops.dropout(g, pk.p_pool, _seed(_SITE_POOL, 0, pk.fwd_i), rt.seed_dev)
            g = ops.act_bwd(ACT_TANH, g, pk.pooled_raw)
        else:
            g = ops.act_bwd(ACT_TANH, g, pk.pooled)
        pw = model.pooler.dense
        _linear_wgrad(rt, g, pk.x_final, pw.weight, pw.bias, bsz, d, d, ldx=L * d)
        ops.gemm(g, bank.compute(pw.weight), bsz, d, d, out=dx, b_mode=KROW, ldc=L * d, accumulate=True)
    # ---- encoder layers, last to first -----------------------------------------------------------------
    # The dgrad chain runs layer by layer; the upstream gradients that the weight gradients need are written into
    # layer-stacked buffers and ALL layers' weight (+ bias) gradients of one kind follow in one strided-batched GEMM
    # each (4 launches instead of 4 per layer: 12x the blocks per launch, 128x128 tiles, no launch tails).
    nl, stk = len(pk.layers), pk.stk
    # LayerNorm parameter gradients: every LN backward stores per-block partial sums (no atomics); ONE launch after the layer
    # loop adds all 2*nl of them onto the gradient buffer in a fixed order (deterministic).  Frozen LN parameters -> atomics path.
    ln_off = _ln_offsets(model, dev)
    nb = ops.ln_part_blocks(M)
    ln_part = torch.empty(2 * nl, nb, 2, d, dtype=torch.float32, device=dev) if ln_off is not None else None

    def ln_bwd(slot, dy, xpre, ln, mean, rstd, seed, keep):
        if ln_part is not None:
            return ops.layernorm_bwd_part(dy, xpre, ln.weight, mean, rstd, ln_part[slot], pk.p_h, seed, rt.seed_dev, **keep)
        return ops.layernorm_bwd(dy, xpre, ln.weight, mean, rstd, bank.grad_image(ln.weight), bank.grad_image(ln.bias), pk.p_h, seed,
                                 rt.seed_dev, **keep)

    gs = SimpleNamespace(out=torch.empty(nl, M, d, dtype=dt, device=dev), hp=torch.empty(nl, M, ff, dtype=dt, device=dev),
                         att=torch.empty(nl, M, d, dtype=dt, device=dev), qkv=torch.empty(nl, M, 3 * d, dtype=dt, device=dev))
    for li in range(nl - 1, -1, -1):
        layer = model.encoder.layer[li]
        att, so, it, ou = layer.attention.self, layer.attention.output, layer.intermediate, layer.output
        x, qkv, ctx, lse, a_pre, mean1, rstd1, a, hsave, hact, o_pre, mean2, rstd2 = pk.layers[li]
        keep = dict(dx2=gs.out[li]) if pk.p_h > 0 else dict(dx=gs.out[li])
        d_o_pre, d_o_drop = ln_bwd(2 * li + 1, dx, o_pre, ou.LayerNorm, mean2, rstd2, _seed(_SITE_OUT, li, pk.fwd_i), keep)
        g = d_o_drop if d_o_drop is not None else d_o_pre
        dhp = gs.hp[li]
        # (which form the forward saved travels in the pack: a backward never multiplies by the wrong one whatever the flag says by now)
        ops.gemm(g, bank.compute(ou.dense.weight), M, ff, d, out=dhp, b_mode=KROW, gelu_grad_pre=hsave,      # dgrad x GELU'
                 act=ops.ACT_SAVED_GRAD if pk.gelu_saved_grad else ACT_NONE)
        da = torch.empty(M, d, dtype=dt, device=dev)
        ops.gemm(dhp, bank.compute(it.dense.weight), M, d, ff, out=da, b_mode=KROW, residual=d_o_pre)
        keep = dict(dx2=gs.att[li]) if pk.p_h > 0 else dict(dx=gs.att[li])
        d_a_pre, d_a_drop = ln_bwd(2 * li, da, a_pre, so.LayerNorm, mean1, rstd1, _seed(_SITE_SELF_OUT, li, pk.fwd_i), keep)
        g = d_a_drop if d_a_drop is not None else d_a_pre
        dctx = torch.empty(M, d, dtype=dt, device=dev)
        ops.gemm(g, bank.compute(so.dense.weight), M, d, d, out=dctx, b_mode=KROW)
        dqkv = ops.attention_bwd(qkv, pk.key_mask, ctx, dctx, lse, bsz, L, nh, pk.p_a, _seed(_SITE_ATTN, li, pk.fwd_i), rt.seed_dev,
                                 out=gs.qkv[li])
        wqkv = bank.compute_span(att.query.weight, att.value.weight, (3 * d, d))
        dx = torch.empty(M, d, dtype=dt, device=dev)
        ops.gemm(dqkv, wqkv, M, d, 3 * d, out=dx, b_mode=KROW, residual=d_a_pre)
    if ln_part is not None:
        ops.ln_partials_reduce(ln_part, bank.grad, ln_off[0], ln_off[1])
    if rt.overlap & 1 and rt.side_stream is not None and rt.after_encoder_backward is None:
        # the four batched weight-gradient launches on the second branch: they run beside the embedding backwards and the ResNet
        # backward and are joined where that ends (cnn_backward_steps) or, without a ResNet backward, in _EncoderFn.backward
        with rt.side(gs.out, gs.hp, gs.att, gs.qkv, stk.x, stk.ctx, stk.a, stk.hact):
            _encoder_wgrads(model, pk, gs, M)
    else:
        _encoder_wgrads(model, pk, gs, M)
        rt.join()
    # ---- embeddings -----------------------------------------------------------------------------------
    if pk.p_h > 0:
        dx = ops.dropout(dx, pk.p_h, _seed(_SITE_EMB, 0, pk.fwd_i), rt.seed_dev)
    emb, vemb = model.embeddings, model.visual_embeddings
    dpre = torch.empty(M, d, dtype=dt, device=dev)
    ops.layernorm_bwd(dx, pk.pre, emb.LayerNorm.weight, pk.mean0, pk.rstd0, bank.grad_image(emb.LayerNorm.weight),
                      bank.grad_image(emb.LayerNorm.bias), dx=dpre, rows=bsz * lt, seg=(lt, L, 0))
    ops.layernorm_bwd(dx, pk.pre, vemb.LayerNorm.weight, pk.mean0, pk.rstd0, bank.grad_image(vemb.LayerNorm.weight),
                      bank.grad_image(vemb.LayerNorm.bias), dx=dpre, rows=bsz * lv, seg=(lv, L, lt))
    we = emb.word_embeddings
    ops.text_embed_bwd(dpre, pk.ids, bank.grad_image(we.weight), bank.grad_image(emb.position_embeddings.weight),
                       bank.grad_image(emb.token_type_embeddings.weight)[0], lt, L,
                       we.padding_idx if we.padding_idx is not None else -1, repeat=pk.text_repeat)
    dgrid = ops.zeros(pk.grid_shape, torch.float32, dev)
    ops.visual_embed_bwd(dpre, pk.src_row, pk.sel, dgrid, bank.grad_image(vemb.row_position_embeddings.weight),
                         bank.grad_image(vemb.col_position_embeddings.weight), bank.grad_image(vemb.token_type_embeddings.weight)[0],
                         bsz, lv, lt, L)
    if dt == torch.float32:
        return dgrid
    return ops.cast(dgrid, torch.empty(pk.grid_shape, dtype=dt, device=dev))


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, grid, model, ids, mask, src_row, pooled_dropout, text_repeat=1):
        save = ctx.needs_input_grad[0]
        ctx.set_materialize_grads(False)
        seq, pooled, pack = encoder_forward(model, grid, ids, mask, src_row, pooled_dropout, save, text_repeat)
        ctx.model, ctx.pack = model, pack
        if save:
            model.rt.pending_encoder_nodes += 1
        return seq, pooled

    @staticmethod
    def backward(ctx, d_seq, d_pooled):
        dgrid = encoder_backward(ctx.model, ctx.pack, d_seq, d_pooled)
        ctx.pack = None
        rt = ctx.model.rt
        if rt.pending_cnn_nodes == 0:
            rt.join()                                   # (no ResNet backward follows: the side branch ends here)
        rt.pending_encoder_nodes = max(0, rt.pending_encoder_nodes - 1)
        hook = rt.after_encoder_backward
        # a clip LOOP (train_n_clips forwards before one backward) runs several encoder backwards per step: the
        # transformer gradients are complete -- and may start their all-reduce -- only after the last of them
        if hook is not None and rt.pending_encoder_nodes == 0:
            hook()
        return None, dgrid, None, None, None, None, None, None


# =================================================================================================
# heads (small autograd nodes: one consumer each, so autograd never has to add tensors)
# =================================================================================================
class _LinearFn(torch.autograd.Function):
    """y = act(x W^T + b).  ``rows`` = (n_seg, seg_len, seg_stride_rows) selects x rows (b*stride + t)."""
    @staticmethod
    def forward(ctx, anchor, x, rt, weight, bias, act, out_f32, rows):
        bank = rt.bank
        n, k = weight.shape
        dev = x.device
        x2 = x.reshape(-1, k)
        tab, rowmap = None, None
        if rows is None:
            m = x2.shape[0]
        else:
            nseg, seglen, segstride = rows
            m = nseg * seglen
            tab = rt.table(nseg, 1, seglen, 1, 0, segstride * k, 0, k, dev)
        out_dt = torch.float32 if out_f32 else rt.dtype
        ld = n if n < 4 else (n + 3) // 4 * 4              # (1- / 2-column head outputs stay contiguous: the losses read them in place)
        store = torch.empty(m, ld, dtype=out_dt, device=dev)
        y = store[:, :n]
        save = ctx.needs_input_grad[0]
        pre = torch.empty(m, ld, dtype=rt.dtype, device=dev)[:, :n] if (save and act == ACT_GELU) else None
        w = bank.compute(weight)
        if tab is None:
            ops.gemm(x2, w, m, n, k, out=y, shift=bias, act=act, out2=pre)
        else:
            ops.gemm(x2, w, m, n, k, out=y, a_mode=ROWK_GATHER, a_tab=tab, lda=0, R=1, S=1, Cin=k, H=1, W=rows[1], sH=0, sW=k,
                     shift=bias, act=act, out2=pre)
        ctx.rt, ctx.weight, ctx.bias, ctx.act, ctx.rows = rt, weight, bias, act, rows
        ctx.x2, ctx.y, ctx.pre, ctx.m, ctx.x_shape = (x2 if save else None), (y if save else None), pre, m, x.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        rt, weight, bias, act = ctx.rt, ctx.weight, ctx.bias, ctx.act
        bank, dt = rt.bank, rt.dtype
        n, k = weight.shape
        m = ctx.m
        g = dy
        if g.dtype != dt or g.stride(1) != 1:
            g = ops.cast(g.contiguous(), torch.empty(m, n, dtype=dt, device=dy.device))
        if act == ACT_GELU:
            g = ops.act_bwd(ACT_GELU, g.contiguous(), ctx.pre.contiguous())
        elif act != ACT_NONE:
            g = ops.act_bwd(act, g.contiguous(), ctx.y.contiguous().to(dt))
        x2 = ctx.x2
        gw = bank.grad_image(weight)
        gb = bank.grad_image(bias) if bias is not None else None
        if ctx.rows is None:
            if gw is not None:
                # the bias gradient rides on the weight-gradient launch (row sums of g on the matrix core, cb_gemm_desc.a_rowsum)
                split, tile = _pick_split(n, k, m)
                ops.gemm(g, x2, n, k, m, out=gw, a_mode=KROW, lda=g.stride(0), b_mode=KROW, ldb=x2.stride(0), accumulate=True,
                         split_k=split, tile=tile, a_rowsum=gb)
                gb = None
            dx = torch.empty(x2.shape, dtype=dt, device=dy.device)
            ops.gemm(g, bank.compute(weight), m, k, n, out=dx, lda=g.stride(0), b_mode=KROW)
        else:
            nseg, seglen, segstride = ctx.rows
            tab = rt.table(nseg, 1, seglen, 1, 0, segstride * k, 0, k, dy.device)
            if gw is not None:
                split, tile = _pick_split(n, k, m)
                ops.gemm(g, x2, n, k, m, out=gw, a_mode=KROW, lda=g.stride(0), b_mode=KROW_GATHER, b_tab=tab, ldb=0, R=1, S=1,
                         Cin=k, H=1, W=seglen, sH=0, sW=k, accumulate=True, split_k=split, tile=tile)
            rowmap = rt.strided_rowmap(nseg, 1, segstride, 1, seglen, 1, dy.device)
            dx = ops.zeros(x2.shape, dt, dy.device)
            ops.gemm(g, bank.compute(weight), m, k, n, out=dx, lda=g.stride(0), b_mode=KROW, c_rowmap=rowmap)
        if gb is not None:
            ops.colsum(g, gb, m, n, ldg=g.stride(0))
        return None, dx.view(ctx.x_shape), None, None, None, None, None, None


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, x, rt, ln):
        save = ctx.needs_input_grad[0]
        xc = x.contiguous()
        y, mean, rstd = ops.layernorm_fwd(xc, ln.weight, ln.bias, ln.eps, save_stats=save)
        ctx.rt, ctx.ln, ctx.saved = rt, ln, (xc, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, mean, rstd = ctx.saved
        bank = ctx.rt.bank
        dx, _ = ops.layernorm_bwd(dy.contiguous(), xc, ctx.ln.weight, mean, rstd, bank.grad_image(ctx.ln.weight),
                                  bank.grad_image(ctx.ln.bias))
        return None, dx, None, None


class _CrossEntropyFn(torch.autograd.Function):
    """CrossEntropyLoss(reduction='none', ignore_index=-100) on fp32 logits (rows, C)."""
    @staticmethod
    def forward(ctx, logits, labels):
        loss, _ = ops.cross_entropy(logits, labels)
        ctx.save_for_backward(logits, labels)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        logits, labels = ctx.saved_tensors
        _, dlogits = ops.cross_entropy(logits, labels, want_loss=False, dloss=dloss.contiguous(), want_grad=True)
        return dlogits, None


class _HeadLossFn(torch.autograd.Function):
    """Element-wise head losses of the reference through cb_head_loss (reduction "none"): MSE (num_labels == 1), BCE with logits (VQA-style
    soft targets), sigmoid margin ranking of the retrieval head -- src/modeling/modeling.py:359-381, 431-446, 567-575."""
    @staticmethod
    def forward(ctx, logits, targets, kind, group, margin):
        loss, _ = ops.head_loss(kind, logits, targets, group=group, margin=margin)
        ctx.save_for_backward(logits, targets)
        ctx.args = (kind, group, margin)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        logits, targets = ctx.saved_tensors
        kind, group, margin = ctx.args
        _, dx = ops.head_loss(kind, logits, targets, want_loss=False, dloss=dloss.contiguous().float(), want_grad=True, group=group, margin=margin)
        return dx, None, None, None, None


def head_loss_none(kind: int, logits: torch.Tensor, targets: Optional[torch.Tensor] = None, group: int = 1, margin: float = 0.0) -> torch.Tensor:
    lg = (logits if logits.dtype == torch.float32 else logits.float()).contiguous()
    tg = None if targets is None else targets.to(torch.float32).contiguous()
    return _HeadLossFn.apply(lg, tg, kind, group, margin)


def cross_entropy_none(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    lg = logits if logits.dtype == torch.float32 else logits.float()
    return _CrossEntropyFn.apply(lg, labels.contiguous())


class _ClipBertHead(nn.Module):
    """Common part of the task models: owns ``bert`` and the runtime."""
    def __init__(self, config):
        super().__init__()
        config = as_config(config)
        self.config = config
        self.bert = ClipBertBaseModel(config)
        self.rt: Optional[Runtime] = None

    def _mlp(self, pooled, seq):
        rt = self.rt
        h = _LinearFn.apply(rt.anchor, pooled, rt, seq[0].weight, seq[0].bias, ACT_RELU, False, None)
        return _LinearFn.apply(rt.anchor, h, rt, seq[2].weight, seq[2].bias, ACT_NONE, True, None)


def _make_mlp(d, n_out):
    return nn.ModuleList([Linear(d, d * 2), nn.Identity(), Linear(d * 2, n_out)])


class ClipBertForVideoTextRetrieval(_ClipBertHead):
    """src/modeling/modeling.py:523-580."""
    def __init__(self, config):
        super().__init__(config)
        self.classifier = _make_mlp(self.config.hidden_size, self.config.num_labels)
        self.margin = _cfg_get(self.config, "margin", 0.0)

    def forward(self, text_input_ids, visual_inputs, text_input_mask, labels=None, sample_size=-1, src_row=None, text_repeat=1):
        _, pooled = self.bert(text_input_ids, visual_inputs, text_input_mask, src_row, pooled_dropout=True, text_repeat=text_repeat)
        logits = self._mlp(pooled, self.classifier)
        logits, loss = self.calc_loss(logits, labels, sample_size=sample_size)
        return dict(logits=logits, loss=loss)

    def calc_loss(self, logits, labels, sample_size=-1):
        if labels is None:
            return logits, 0
        if self.config.loss_type == "ce":
            loss = cross_entropy_none(logits.view(-1, self.config.num_labels), labels.view(-1))
        elif self.config.loss_type == "rank":
            # sigmoid margin ranking, modeling.py:567-575: rows of (1 positive + negatives) scores per video
            assert sample_size > 0
            group = logits.numel() // sample_size
            if group < 2:                      # no negatives: the reference's scores[:, 1:] is (B, 0) and so is its loss (modeling.py:572-575)
                loss = logits.new_zeros((sample_size, 0), dtype=torch.float32)
            else:
                loss = head_loss_none(ops.LOSS_RANK, logits.reshape(sample_size, -1), group=group, margin=self.margin)
        else:
            raise ValueError("Invalid option for config.loss_type")
        return logits, loss


class ClipBertForMultipleChoice(_ClipBertHead):
    """src/modeling/modeling.py:387-451."""
    def __init__(self, config):
        super().__init__(config)
        self.classifier = _make_mlp(self.config.hidden_size, 1)

    def forward(self, text_input_ids, visual_inputs, text_input_mask, labels=None, src_row=None, text_repeat=1):
        _, pooled = self.bert(text_input_ids, visual_inputs, text_input_mask, src_row, pooled_dropout=True, text_repeat=text_repeat)
        logits = self._mlp(pooled, self.classifier)
        logits, loss = self.calc_loss(logits, labels)
        return dict(logits=logits, loss=loss)

    def calc_loss(self, logits, labels):
        if self.config.loss_type == "ce":
            logits = logits.reshape(-1, self.config.num_labels)
        if labels is None:
            return logits, 0
        if self.config.num_labels == 1:
            loss = head_loss_none(ops.LOSS_MSE, logits.reshape(-1), labels.reshape(-1))
        elif self.config.loss_type == "bce":
            loss = head_loss_none(ops.LOSS_BCE, logits, labels)
        elif self.config.loss_type == "ce":
            loss = cross_entropy_none(logits.contiguous(), labels.view(-1))
        else:
            raise ValueError("Invalid option for config.loss_type")
        return logits, loss


class ClipBertForSequenceClassification(_ClipBertHead):
    """src/modeling/modeling.py:327-384."""
    def __init__(self, config):
        super().__init__(config)
        self.classifier = _make_mlp(self.config.hidden_size, self.config.num_labels)

    def forward(self, text_input_ids, visual_inputs, text_input_mask, labels=None, src_row=None, text_repeat=1):
        _, pooled = self.bert(text_input_ids, visual_inputs, text_input_mask, src_row, pooled_dropout=True, text_repeat=text_repeat)
        logits = self._mlp(pooled, self.classifier)
        logits, loss = self.calc_loss(logits, labels)
        return dict(logits=logits, loss=loss)

    def calc_loss(self, logits, labels):
        if labels is None:
            return logits, 0
        if self.config.num_labels == 1:
            loss = head_loss_none(ops.LOSS_MSE, logits.reshape(-1), labels.reshape(-1))
        elif self.config.loss_type == "bce":
            loss = head_loss_none(ops.LOSS_BCE, logits, labels)
        elif self.config.loss_type == "ce":
            loss = cross_entropy_none(logits.view(-1, self.config.num_labels), labels.view(-1))
        else:
            raise ValueError("Invalid option for config.loss_type")
        return logits, loss


class BatchNorm1d(nn.Module):
    """parameter / buffer holder with torch.nn.BatchNorm1d's state-dict keys"""
    def __init__(self, d, eps=1e-5, momentum=0.1):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d))
        self.bias = nn.Parameter(torch.zeros(d))
        self.register_buffer("running_mean", torch.zeros(d))
        self.register_buffer("running_var", torch.ones(d))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.eps, self.momentum = eps, momentum


class _EluBnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, x, rt, bn, training):
        xc = x.contiguous()
        y, sm, si = ops.elu_bn1d_fwd(xc, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, bn.momentum, bn.eps, save=True)
        if training:
            bn.num_batches_tracked += 1
        ctx.rt, ctx.bn, ctx.training, ctx.saved = rt, bn, training, (xc, sm, si)
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, sm, si = ctx.saved
        bank, bn = ctx.rt.bank, ctx.bn
        dx = ops.elu_bn1d_bwd(dy.contiguous(), xc, bn.weight, sm, si, bank.grad_image(bn.weight), bank.grad_image(bn.bias), ctx.training)
        return None, dx, None, None, None


class ClipBertForRegression(_ClipBertHead):
    """src/modeling/modeling.py:454-507: pooled -> dropout -> Linear -> ELU -> BatchNorm1d -> dropout -> Linear(1); MSE loss.
    (No runner of the reference instantiates it -- the TGIF "count" task goes through ClipBertForSequenceClassification with
    num_labels = 1 -- but it is part of the module API.)"""
    def __init__(self, config):
        super().__init__(config)
        d = self.config.hidden_size
        self.regressor = nn.ModuleList([Linear(d, d), nn.Identity(), BatchNorm1d(d), nn.Identity(), Linear(d, 1)])

    def forward(self, text_input_ids, visual_inputs, text_input_mask, labels=None, src_row=None, text_repeat=1):
        rt = self.rt
        _, pooled = self.bert(text_input_ids, visual_inputs, text_input_mask, src_row, pooled_dropout=True, text_repeat=text_repeat)
        reg = self.regressor
        h = _LinearFn.apply(rt.anchor, pooled, rt, reg[0].weight, reg[0].bias, ACT_NONE, False, None)
        h = _EluBnFn.apply(rt.anchor, h, rt, reg[2], self.training)
        p = _drop_p(self, self.training)
        if p > 0:
            h = _DropoutFn.apply(h, rt, p, rt.forward_count)
        logits = _LinearFn.apply(rt.anchor, h, rt, reg[4].weight, reg[4].bias, ACT_NONE, True, None)
        logits, loss = self.calc_loss(logits, labels)
        return dict(logits=logits, loss=loss)

    def calc_loss(self, logits, labels):
        if labels is None:
            return logits, 0
        if self.config.loss_type == "mse":
            return logits, head_loss_none(ops.LOSS_MSE, logits.reshape(-1), labels.reshape(-1))
        raise ValueError(f"Invalid option {self.config.loss_type} for config.loss_type")


_SITE_REG = 6


class _DropoutFn(torch.autograd.Function):
    """nn.Dropout on a small head activation (the regression head's second dropout): stateless hash mask, same in backward"""
    @staticmethod
    def forward(ctx, x, rt, p, fwd_i):
        ctx.rt, ctx.p, ctx.seed = rt, p, _seed(_SITE_REG, 0, fwd_i)
        return ops.dropout(x.contiguous(), p, ctx.seed, rt.seed_dev)

    @staticmethod
    def backward(ctx, dy):
        return ops.dropout(dy.contiguous(), ctx.p, ctx.seed, ctx.rt.seed_dev), None, None, None


class _PredictionHeadTransform(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = LayerNorm(config.hidden_size, config.layer_norm_eps)


class _Decoder(nn.Module):
    def __init__(self, weight, bias):
        super().__init__()
        self.weight = weight          # tied to bert.embeddings.word_embeddings.weight
        self.bias = bias              # same Parameter as predictions.bias (transformers.py:507-510)


class _LMPredictionHead(nn.Module):
    def __init__(self, config, word_weight):
        super().__init__()
        self.transform = _PredictionHeadTransform(config)
        self.bias = nn.Parameter(torch.zeros(config.vocab_size))
        self.decoder = _Decoder(word_weight, self.bias)


class _PreTrainingHeads(nn.Module):
    def __init__(self, config, word_weight):
        super().__init__()
        self.predictions = _LMPredictionHead(config, word_weight)
        self.seq_relationship = Linear(config.hidden_size, 2)


class ClipBertForPreTraining(_ClipBertHead):
    """src/modeling/modeling.py:241-307 with BertPreTrainingHeads (transformers.py:479-547)."""
    def __init__(self, config):
        super().__init__(config)
        self.cls = _PreTrainingHeads(self.config, self.bert.embeddings.word_embeddings.weight)

    def get_output_embeddings(self):
        return self.cls.predictions.decoder

    def forward(self, text_input_ids, visual_inputs, text_input_mask, mlm_labels=None, itm_labels=None, src_row=None):
        rt = self.rt
        seq, pooled = self.bert(text_input_ids, visual_inputs, text_input_mask, src_row)
        b, L, d = seq.shape
        lt = text_input_mask.shape[1]
        pred = self.cls.predictions
        # heads on the TEXT rows only (modeling.py:283-285): gathered inside the GEMM loader
        h = _LinearFn.apply(rt.anchor, seq, rt, pred.transform.dense.weight, pred.transform.dense.bias, ACT_GELU, False,
                            (b, lt, L))
        h = _LayerNormFn.apply(rt.anchor, h, rt, pred.transform.LayerNorm)
        scores = _LinearFn.apply(rt.anchor, h, rt, pred.decoder.weight, pred.bias, ACT_NONE, True, None)
        rel = self.cls.seq_relationship
        itm = _LinearFn.apply(rt.anchor, pooled, rt, rel.weight, rel.bias, ACT_NONE, True, None)
        v = self.config.vocab_size
        mlm_loss = cross_entropy_none(scores, mlm_labels.view(-1)) if mlm_labels is not None else 0
        itm_loss = cross_entropy_none(itm.view(-1, 2), itm_labels.view(-1)) if itm_labels is not None else 0
        return dict(mlm_scores=scores.view(b, lt, v), mlm_loss=mlm_loss, mlm_labels=mlm_labels, itm_scores=itm,
                    itm_loss=itm_loss, itm_labels=itm_labels)


# =================================================================================================
# end-to-end wrapper
# =================================================================================================
class ClipBert(nn.Module):
    """src/modeling/e2e_model.py:14-50."""
    def __init__(self, config, input_format="BGR", detectron2_model_cfg=None, transformer_cls=ClipBertForPreTraining):
        super().__init__()
        config = as_config(config)
        self.config = config
        self.detectron2_model_cfg = detectron2_model_cfg
        self.cnn = GridFeatBackbone(detectron2_model_cfg=detectron2_model_cfg, config=config, input_format=input_format)
        self.transformer = transformer_cls(config)
        self.retrieval = transformer_cls == ClipBertForVideoTextRetrieval
        self.rt: Optional[Runtime] = None
        self._src_cache = {}
        # nn.Module.load_state_dict copies into the fp32 master views of a prepared model: everything derived from them
        # (bf16 compute copies, folded FrozenBN vectors, packed stem filter) is refreshed afterwards -- also when only
        # a sub-module is loaded (load_state_dict_with_mismatch(model.transformer, ...), load_separate_ckpt)
        # (post-hooks fire for the module load_state_dict was CALLED on only: cnn.feature is what load_detectron2_backbone loads)
        for mod in (self, self.cnn, self.cnn.feature, self.transformer, self.transformer.bert):
            mod.register_load_state_dict_post_hook(lambda _m, _keys, owner=self: owner.refresh_compute())

    def refresh_compute(self):
        if self.rt is None:
            return
        self.rt.bank.sync_compute()
        self.rt.stem_w = None
        for m in self.modules():
            if isinstance(m, Conv2d):
                m._ss = None

    # ---- MI355X runtime --------------------------------------------------------------------------------
    def prepare(self, dtype=torch.bfloat16, device=None, transformer_lr_mul_prefix="", cnn_lr_mul_prefix="grid_encoder",
                overlap_wgrad=False):
        """Move parameters into the flat HBM buffers and build compute copies.  Call after loading
        weights / changing requires_grad (freeze_cnn_backbone) and before the first forward."""
        device = torch.device(device) if device is not None else next(self.parameters()).device
        for buf_owner in self.modules():
            if isinstance(buf_owner, FrozenBatchNorm2d):
                buf_owner.to(device)
        rt = Runtime()
        rt.dtype = dtype
        # re-preparing (freeze_cnn_backbone on a prepared model) must rebuild the SAME parameter-group layout
        rt.prepare_args = dict(dtype=dtype, device=device, transformer_lr_mul_prefix=transformer_lr_mul_prefix,
                               cnn_lr_mul_prefix=cnn_lr_mul_prefix, overlap_wgrad=overlap_wgrad)
        rt.bank = ParamBank(self, device, dtype, transformer_lr_mul_prefix, cnn_lr_mul_prefix)
        rt.seed_dev = torch.zeros(1, dtype=torch.int64, device=device)
        rt.anchor = torch.zeros(1, dtype=torch.float32, device=device, requires_grad=True)
        if dtype == torch.bfloat16 and (device.type == "cuda" or ops._ALLOW_HOST_POINTERS):
            ops.splitk_workspace(device)            # scratch of cb_gemm's K-split: allocated here, before any hipGraph capture
        if device.type == "cuda" and overlap_wgrad:
            # overlap_wgrad: True = every convolution's weight gradient on a second stream (round 1); an int = Runtime.overlap bits
            rt.overlap = 4 if overlap_wgrad is True else int(overlap_wgrad)
            rt.side_stream = torch.cuda.Stream(device=device)
            if rt.overlap & 3 and dtype == torch.bfloat16:
                rt.side_ws = ops.new_splitk_workspace(device)
            if rt.overlap & 8:
                rt.fan_streams = [torch.cuda.Stream(device=device) for _ in range(3)]
        for m in self.modules():
            if hasattr(m, "rt"):
                m.rt = rt
            if isinstance(m, (Conv2d,)):
                m._ss = None
        enc = []
        for layer in self.transformer.bert.encoder.layer:
            enc += [layer.attention.self.query.weight, layer.attention.self.key.weight, layer.attention.self.value.weight,
                    layer.attention.output.dense.weight, layer.intermediate.dense.weight, layer.output.dense.weight]
        rt.bank.set_lazy_span(enc)
        # the ResNet's trainable convolution weights + the grid encoder's: one weight-gradient launch each per backward -> first-writer stores
        if dtype == torch.bfloat16:
            rt.bank.set_fresh_params([m.weight for m in self.cnn.modules() if isinstance(m, (Conv2d, _GridConv)) and rt.bank.is_trainable(m.weight)])
        return self

    def _ensure_prepared(self, device):
        if self.rt is None:
            self.prepare(device=device)

    def grid_features(self, visual_inputs):
        """(Bv, T, 3, H, W) frames -> (Bv, T, H', W', hidden) grid features: the CNN half of forward() on its own, so that
        inference can compute each clip's features once and reuse them across text mini-batches (SURVEY 8f N1;
        the reference recomputes them per mini-batch, run_video_retrieval.py:655-666)."""
        self._ensure_prepared(visual_inputs.device)
        return self.cnn(visual_inputs)

    def forward(self, batch):
        vis = batch["visual_inputs"]
        self._ensure_prepared(vis.device)
        batch["visual_inputs"] = self.cnn(vis)
        return self.forward_from_grid(batch)

    def forward_from_grid(self, batch, clip_fold: int = 1):
        """forward() for a batch whose ``visual_inputs`` already are grid features (see grid_features).

        clip_fold = n > 1: the grid holds n clips per video, video-major ((Bv*n, T, H', W', d): row v*n + c is clip c of
        video v -- the plain ``view`` of the reference's (B, n*T, 3, H, W) frame tensor, no copy), and the text batch is
        the reference's batch repeated n times, clip-major (row c*B' + j = pair j looking at clip c).  One forward then
        does what the reference's clip loop does in n (run_video_retrieval.py:396-401); logits come back clip-major, i.e.
        ``logits.view(n, B', C)`` is the stack the loop builds with torch.stack."""
        repeat_counts = batch["n_examples_list"]
        del batch["n_examples_list"]
        vis = batch["visual_inputs"]
        # repeat_tensor_rows (data_utils.py:344-357) is fused into the visual-embedding gather
        src_row = None
        if clip_fold > 1:
            key = (tuple(repeat_counts), clip_fold, str(vis.device))
            src_row = self._src_cache.get(key)
            if src_row is None:
                per_video = [i for i, r in enumerate(repeat_counts) for _ in range(r)]
                src_row = torch.tensor([v * clip_fold + c for c in range(clip_fold) for v in per_video], dtype=torch.int32,
                                       device=vis.device)
                self._src_cache[key] = src_row
            assert vis.shape[0] == len(repeat_counts) * clip_fold, "clip_fold: grid rows != videos x clips"
        elif sum(repeat_counts) != len(repeat_counts):
            key = (tuple(repeat_counts), str(vis.device))
            src_row = self._src_cache.get(key)
            if src_row is None:
                src_row = torch.tensor([i for i, r in enumerate(repeat_counts) for _ in range(r)], dtype=torch.int32,
                                       device=vis.device)
                self._src_cache[key] = src_row
        n_txt = batch["text_input_ids"].shape[0]
        if clip_fold > 1 and src_row is not None and n_txt * clip_fold == src_row.numel():
            batch["text_repeat"] = clip_fold           # the captions of ONE clip: every clip reads them in place (no repeated copies)
        elif src_row is not None and src_row.numel() != n_txt:
            raise ValueError(f"n_examples_list describes {src_row.numel()} (video, text) pairs but the text batch has "
                             f"{n_txt} rows")
        if self.retrieval:
            batch["sample_size"] = len(repeat_counts)
        return self.transformer(src_row=src_row, **batch)

    def load_separate_ckpt(self, cnn_weights_path=None, bert_weights_path=None):
        """e2e_model.py:43-48: detectron2 backbone weights (``grid_feat_R-50.pth`` / a detectron2 ``.pkl`` / a torchvision
        ResNet-50 state dict, see clipbert_amd.checkpoint) into ``cnn.feature`` and a BERT / ClipBERT transformer
        checkpoint into ``transformer``.  Works before or after prepare(); raises if a file matches no key at all."""
        from . import checkpoint as ckpt
        if cnn_weights_path:
            n = ckpt.load_detectron2_backbone(self.cnn, cnn_weights_path)
            if n == 0:
                raise RuntimeError(f"{cnn_weights_path}: no key of the checkpoint matches the ResNet-50 grid backbone")
        if bert_weights_path:
            n = load_state_dict_with_mismatch(self.transformer, bert_weights_path)
            if n == 0:
                raise RuntimeError(f"{bert_weights_path}: no key of the checkpoint matches {type(self.transformer).__name__}")
        self.refresh_compute()                  # (the load hooks already did it; kept explicit: masters -> compute copies)

    def freeze_cnn_backbone(self):
        """e2e_model.py:49-51.  Changes which parameters are trainable, i.e. the layout of the flat buffers: call it
        before prepare() / before building the optimizer."""
        if self.rt is not None and self.rt.bank.clients > 0:
            raise RuntimeError("freeze_cnn_backbone() after an optimizer / GradSync was built on this model's parameter "
                               "bank: freeze first, then prepare() and build the optimizer")
        for _n, p in self.cnn.feature.named_parameters():
            p.requires_grad = False
        if self.rt is not None:
            self.prepare(**self.rt.prepare_args)


def cnn_early_split(model: "ClipBert") -> Optional[int]:
    """Element offset in the flat gradient buffer where res5's parameters start (they are the tail of the CNN range: same module
    order as the reference's parameter groups); None when res5 is frozen or not contiguous at the end.  GradSync.set_cnn_split."""
    bank = model.rt.bank
    ps = [p for p in model.cnn.feature.backbone.res5.parameters() if bank.is_trainable(p)]
    if not ps:
        return None
    start = min(bank.offset[id(p)] for p in ps)
    g6 = bank.group_range[6]
    others = [bank.offset[id(p)] for n, p in model.cnn.feature.backbone.named_parameters()
              if bank.is_trainable(p) and not n.startswith("res5.")]
    if not (g6[0] <= start < g6[1]) or any(o >= start for o in others):
        return None
    return start


def load_state_dict_with_mismatch(model: nn.Module, loaded_state_dict_or_path) -> int:
    """Key/shape tolerant load (src/utils/load_save.py:71-100): accepts a state dict or a path to one; drops
    shape-mismatched and unknown keys (e.g. the reference checkpoint's dead RPN/ROI-head weights), loads the rest
    non-strictly.  Returns the number of tensors loaded (the reference logs the key differences instead)."""
    sd = loaded_state_dict_or_path
    if isinstance(sd, (str, bytes)) or hasattr(sd, "__fspath__"):
        sd = torch.load(sd, map_location="cpu")
    own = model.state_dict()
    ok = {k: v for k, v in sd.items() if k in own and tuple(own[k].shape) == tuple(v.shape)}
    model.load_state_dict(ok, strict=False)
    return len(ok)
