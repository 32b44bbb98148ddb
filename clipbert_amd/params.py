"""Flat parameter storage for the MI355X path.

All trainable parameters live in ONE fp32 master buffer laid out group-by-group in the reference's
8 optimizer groups (src/optimization/utils.py:96-161), with a parallel fp32 gradient buffer, two
AdamW state buffers and -- in bf16 mode -- a bf16 compute copy that the fused AdamW kernel rewrites
in the same pass.  ``nn.Parameter.data`` / ``.grad`` are views into these buffers, so

* the optimizer is a handful of kernel launches over contiguous ranges (no per-tensor loop),
* gradient all-reduce works on large contiguous slices (buckets) of one buffer,
* conv weights keep the reference's logical OIHW shape (state-dict compatible) but a channels_last
  memory image = KRSC, which is exactly what the implicit-GEMM kernels read,
* query/key/value weights (and biases) are adjacent, so the fused QKV GEMM reads them as one
  [3*hidden, hidden] matrix without any copy.

Frozen parameters and FrozenBN buffers get a compute copy once (``sync_compute``).
"""
from typing import Dict, List, Optional

import torch
from torch import nn

from . import ops

NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight")     # src/optimization/utils.py:134
N_GROUPS = 8


def group_of(name: str, transformer_lr_mul_prefix: str = "", cnn_lr_mul_prefix: str = "grid_encoder") -> int:
    """Index into the reference's param-group list: transformer {top decay, top no-decay, rest decay,
    rest no-decay} then the same four for the cnn (run_video_retrieval.py:455-467 indexes them so)."""
    if "transformer" in name:
        base, pre = 0, transformer_lr_mul_prefix
    elif "cnn" in name:
        base, pre = 4, cnn_lr_mul_prefix
    else:
        raise ValueError(f"parameter {name!r} belongs to neither 'transformer' nor 'cnn'")
    top = 0 if (pre != "" and pre in name) else 2
    nd = 1 if any(t in name for t in NO_DECAY) else 0
    return base + top + nd


def _physical(t: torch.Tensor) -> torch.Tensor:
    """1-D view of the memory image of a dense tensor (contiguous or channels_last)."""
    if t.dim() == 4:
        return t.permute(0, 2, 3, 1).reshape(-1)       # KRSC image of an OIHW tensor
    return t.reshape(-1)


class ParamBank:
    ALIGN = 64          # elements; keeps every parameter 256-byte aligned in fp32 and 128-byte in bf16
    GROUP_ALIGN = 512   # every optimizer group starts and ends at a multiple of 512 elements (zero padding behind its last parameter):
                        # ranges made of whole groups -- and, inside the CNN group, of whole convolution weights, whose sizes are
                        # multiples of 512 -- divide into world x 64-element shards for world in {1, 2, 4, 8}: what
                        # GradSync(shard=True) reduce-scatters / all-gathers in place

    def __init__(self, root: nn.Module, device, compute_dtype: torch.dtype, transformer_lr_mul_prefix: str = "",
                 cnn_lr_mul_prefix: str = "grid_encoder", name_prefix: str = ""):
        assert compute_dtype in (torch.bfloat16, torch.float32)
        self.device = torch.device(device)
        self.compute_dtype = compute_dtype
        self.clients = 0            # optimizers / gradient exchangers built on this bank (they hold views of its buffers)
        named = []
        seen = {}
        for name, p in root.named_parameters(remove_duplicate=False):
            if id(p) in seen:               # tied weights (MLM decoder <-> word embeddings, decoder.bias)
                continue
            seen[id(p)] = name
            named.append((name_prefix + name, p))
        self.names: Dict[int, str] = {id(p): n for n, p in named}
        trainable = [(n, p) for n, p in named if p.requires_grad]
        frozen = [(n, p) for n, p in named if not p.requires_grad]
        groups: List[List] = [[] for _ in range(N_GROUPS)]
        for n, p in trainable:
            groups[group_of(n, transformer_lr_mul_prefix, cnn_lr_mul_prefix)].append((n, p))
        self.offset: Dict[int, int] = {}
        self.group_range: List[tuple] = []
        off = 0
        for g in groups:
            start = off
            for _n, p in g:
                self.offset[id(p)] = off
                off += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
            off = (off + self.GROUP_ALIGN - 1) // self.GROUP_ALIGN * self.GROUP_ALIGN      # (the padding stays zero: AdamW leaves it at zero)
            self.group_range.append((start, off))
        self.n_train = off
        dev = self.device
        self.master = torch.zeros(max(off, 1), dtype=torch.float32, device=dev)
        self.grad = torch.zeros(max(off, 1), dtype=torch.float32, device=dev)
        self.exp_avg: Optional[torch.Tensor] = None
        self.exp_avg_sq: Optional[torch.Tensor] = None
        self.w16 = torch.zeros(max(off, 1), dtype=torch.bfloat16, device=dev) if compute_dtype == torch.bfloat16 else None
        self._trainable = trainable
        for _n, p in trainable:
            self._rebind(p, self.master, self.offset[id(p)], grad=True)
        # frozen parameters: own flat fp32 + compute copy
        foff = 0
        self.f_offset: Dict[int, int] = {}
        for _n, p in frozen:
            self.f_offset[id(p)] = foff
            foff += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.f_master = torch.zeros(max(foff, 1), dtype=torch.float32, device=dev)
        self.f_w16 = torch.zeros(max(foff, 1), dtype=torch.bfloat16, device=dev) if compute_dtype == torch.bfloat16 else None
        for _n, p in frozen:
            self._rebind(p, self.f_master, self.f_offset[id(p)], grad=False)
        self.sync_compute()

    def _view(self, flat: torch.Tensor, off: int, like: torch.Tensor) -> torch.Tensor:
        n = like.numel()
        if like.dim() == 4:
            o, i, r, s = like.shape
            return flat[off:off + n].view(o, r, s, i).permute(0, 3, 1, 2)     # logical OIHW, memory KRSC
        return flat[off:off + n].view(like.shape)

    def _rebind(self, p: nn.Parameter, flat: torch.Tensor, off: int, grad: bool):
        v = self._view(flat, off, p)
        v.copy_(p.data.to(self.device))
        p.data = v
        if grad:
            p.grad = self._view(self.grad, off, p)

    # ---- accessors used by the forward/backward code --------------------------------------------
    def compute(self, p: nn.Parameter) -> torch.Tensor:
        """Compute-dtype tensor with the parameter's MEMORY image: 4-D -> (O, R, S, I), else its shape."""
        key = id(p)
        if key in self.offset:
            flat, off = (self.w16 if self.w16 is not None else self.master), self.offset[key]
        else:
            flat, off = (self.f_w16 if self.f_w16 is not None else self.f_master), self.f_offset[key]
        n = p.numel()
        if p.dim() == 4:
            o, i, r, s = p.shape
            return flat[off:off + n].view(o, r, s, i)
        return flat[off:off + n].view(p.shape)

    def compute_span(self, first: nn.Parameter, last: nn.Parameter, shape) -> torch.Tensor:
        """One tensor over several ADJACENT parameters (fused QKV)."""
        flat = self.w16 if self.w16 is not None else self.master
        a, b = self.offset[id(first)], self.offset[id(last)] + last.numel()
        t = flat[a:b]
        assert t.numel() == int(torch.Size(shape).numel()), "parameters are not adjacent in the flat buffer"
        return t.view(shape)

    def master_span(self, first: nn.Parameter, last: nn.Parameter, shape) -> torch.Tensor:
        """fp32 master view over adjacent parameters (fused QKV bias, read by the GEMM epilogue)."""
        a, b = self.offset[id(first)], self.offset[id(last)] + last.numel()
        t = self.master[a:b]
        assert t.numel() == int(torch.Size(shape).numel()), "parameters are not adjacent in the flat buffer"
        return t.view(shape)

    def grad_image(self, p: nn.Parameter) -> torch.Tensor:
        """fp32 gradient in the parameter's memory image (see ``compute``); None if frozen."""
        key = id(p)
        if key not in self.offset:
            return None
        off, n = self.offset[key], p.numel()
        if p.dim() == 4:
            o, i, r, s = p.shape
            return self.grad[off:off + n].view(o, r, s, i)
        return self.grad[off:off + n].view(p.shape)

    def grad_span(self, first: nn.Parameter, last: nn.Parameter, shape) -> torch.Tensor:
        a, b = self.offset[id(first)], self.offset[id(last)] + last.numel()
        return self.grad[a:b].view(shape)

    def is_trainable(self, p: nn.Parameter) -> bool:
        return id(p) in self.offset

    def fp32_read_ranges(self):
        """(lo, hi) element ranges of the trainable parameters that kernels read from the fp32 MASTER buffer rather than the bf16
        compute copy: every 1-D parameter (biases, LayerNorm and BatchNorm1d vectors are handed to the kernels as fp32 pointers)."""
        return [(self.offset[id(p)], self.offset[id(p)] + p.numel()) for _n, p in self._trainable if p.dim() <= 1]

    # ---- maintenance -----------------------------------------------------------------------------
    def sync_compute(self):
        """Refresh the bf16 compute copies from the fp32 masters (after loading / editing weights).  After owner-only updates
        (GradSync(shard=True)) the masters of the decay groups are current on their owners only: copying them over the all-gathered
        compute weights would silently roll the other ranks' pieces back -- gather_state() first."""
        self.assert_whole("ParamBank.sync_compute()")
        if self.w16 is not None and self.n_train > 0:
            ops.cast(self.master, self.w16)
        if self.f_w16 is not None:
            ops.cast(self.f_master, self.f_w16)

    def set_lazy_span(self, params):
        """``params``: parameters whose gradients are produced by launches that can STORE instead of accumulate (the strided-
        batched weight-gradient GEMMs of the encoder layers).  If they tile one contiguous range of the flat gradient buffer,
        ``zero_grad(lazy=True)`` skips that range and the first producer after it overwrites (first-writer stores: no memset
        and no fp32 read-modify-write for ~57 % of the gradient bytes)."""
        self.lazy_span, self.lazy_fresh = None, False
        ps = [p for p in params if id(p) in self.offset]
        if not ps or len(ps) != len(list(params)):
            return
        spans = sorted((self.offset[id(p)], self.offset[id(p)] + (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN) for p in ps)
        for (a0, b0), (a1, _b1) in zip(spans, spans[1:]):
            if b0 != a1:
                return                                  # another parameter sits in between
        self.lazy_span = (spans[0][0], spans[-1][1])

    def set_fresh_params(self, params):
        """``params``: parameters whose weight gradient is produced by ONE launch per backward that can store instead of accumulate
        (the ResNet's convolution weights: cb_gemm accumulate = 2, first writer).  If they tile one contiguous range of the flat gradient
        buffer, ``zero_grad(lazy=True)`` skips it too; ``take_fresh_param`` tells each producer whether it is the first writer of this
        step, and ``finish_fresh`` zeroes what no producer wrote (a partial backward) before the gradients are read."""
        self.fresh_span, self.fresh_ids, self.fresh_left = None, {}, set()
        ps = [p for p in params if id(p) in self.offset]
        if not ps:
            return
        spans = sorted((self.offset[id(p)], self.offset[id(p)] + (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN, id(p)) for p in ps)
        for (_a0, b0, _i0), (a1, _b1, _i1) in zip(spans, spans[1:]):
            if b0 != a1:
                return                                  # another parameter sits in between
        if any(p.numel() % self.ALIGN for p in ps):
            return                                      # (padding inside the range would never be written)
        self.fresh_span = (spans[0][0], spans[-1][1])
        self.fresh_ids = {i: (a, b) for a, b, i in spans}

    def take_fresh_param(self, p: nn.Parameter) -> bool:
        """True exactly once per gradient epoch for a parameter of the fresh range: its gradient holds garbage and must be STORED now"""
        left = getattr(self, "fresh_left", None)
        if left and id(p) in left:
            left.discard(id(p))
            return True
        return False

    def finish_fresh(self):
        """zero the gradients of fresh-range parameters no producer wrote since zero_grad(lazy=True) (called before anything reads them)"""
        left = getattr(self, "fresh_left", None)
        if left:
            for i in sorted(left):
                a, b = self.fresh_ids[i]
                ops.zero_(self.grad[a:b])
            left.clear()
            self.fold_invalidate()

    # ---- squared-norm shares (cb_gemm_desc.sq_slots): the weight-gradient launches of a step leave sum(dW^2) per output tile in slots ----
    SQ_SLOTS = 1 << 16

    def enable_norm_fold(self):
        """allocate the accumulator + slots (zeroed by zero_grad(lazy=True) from then on); idempotent"""
        if getattr(self, "sq_buf", None) is None:
            self.sq_buf = torch.zeros(1 + self.SQ_SLOTS, dtype=torch.float32, device=self.device)
            self.fold = None
        return self.sq_buf

    def fold_take(self, slots: int, covers: str):
        """``slots`` slots for one weight-gradient launch of the running step (None: no share wanted / possible); ``covers``: what the
        launch writes completely -- one of the four encoder kinds ("enc:<kind>") or one fresh-range parameter ("cnn")"""
        f = getattr(self, "fold", None)
        if f is None or not f["valid"] or getattr(self, "norm_fold_blocked", False):
            return None
        if f["next"] + slots > self.SQ_SLOTS:
            f["valid"] = False
            return None
        view = self.sq_buf[1 + f["next"]:1 + f["next"] + slots]
        f["next"] += slots
        f["covers"].append(covers)
        return view

    def fold_invalidate(self):
        if getattr(self, "fold", None) is not None:
            self.fold["valid"] = False

    def fold_result(self):
        """(segments, slots) for ops.sq_sum_fold if the shares of this step are usable: every encoder kind stored once, every fresh-range
        parameter stored once with a share -- else None (the caller runs the full pass)"""
        f = getattr(self, "fold", None)
        if f is None or not f["valid"] or getattr(self, "lazy_fresh", False) or getattr(self, "fresh_left", None) or getattr(self, "norm_fold_blocked", False):
            return None
        covered = []
        enc = [c for c in f["covers"] if c.startswith("enc:")]
        if self.lazy_span is not None and sorted(enc) == sorted(f"enc:{k}" for k in ("out", "ffn", "att", "qkv")):
            covered.append(self.lazy_span)
        elif enc:
            return None                                  # (a partial set of shares inside the lazy span cannot be subtracted)
        ncnn = sum(1 for c in f["covers"] if c == "cnn")
        if getattr(self, "fresh_span", None) is not None and ncnn == len(self.fresh_ids):
            covered.append(self.fresh_span)
        elif ncnn:
            return None
        if not covered:
            return None
        segs, lo = [], 0
        for a, b in sorted(covered):
            if a > lo:
                segs.append((lo, a))
            lo = b
        if lo < self.n_train:
            segs.append((lo, self.n_train))
        return segs, self.sq_buf[1:1 + f["next"]]

    def assert_whole(self, what: str):
        """Owner-only updates (GradSync(shard=True)) leave every rank with ITS pieces of the fp32 masters and AdamW moments: anything
        that reads the whole state (state_dict(), a checkpoint) needs GradSync.gather_state() on all ranks first."""
        if getattr(self, "owner_only_dirty", False):
            raise RuntimeError(f"{what}: the parameters / optimizer moments are sharded over the ranks (owner-only update) -- "
                               "call GradSync.gather_state(optimizer) on every rank first")

    def zero_grad(self, lazy: bool = False):
        """lazy=True: the caller guarantees that a full backward follows before the gradients are read; the lazy span (see
        set_lazy_span) is then left as it is and marked fresh -- its producer stores instead of accumulating."""
        self.grad_epoch = getattr(self, "grad_epoch", 0) + 1       # one per gradient group: GradSync.wait() tells a repeated wait()
        span = getattr(self, "lazy_span", None)                   # (nothing to do) from a step whose exchange was never issued
        if lazy and span is not None:
            skip = sorted([span] + ([self.fresh_span] if getattr(self, "fresh_span", None) is not None else []))
            lo = 0
            fills = []                                         # (the gaps around the first-writer ranges + the norm slots: ONE launch, cb_zero_ranges)
            for a, b in skip:
                if a > lo:
                    fills.append(self.grad[lo:a])
                lo = max(lo, b)
            if lo < self.grad.numel():
                fills.append(self.grad[lo:])
            self.lazy_fresh = True
            self.fresh_left = set(getattr(self, "fresh_ids", {}))
            if getattr(self, "sq_buf", None) is not None:      # the norm accumulator and this step's share slots start at zero
                fills.append(self.sq_buf)
                self.fold = dict(valid=True, next=0, covers=[])
            ops.zero_many(fills)
        else:
            ops.zero_(self.grad)
            self.lazy_fresh = False
            self.fresh_left = set()
            self.fold = None

    def zero_grad_range(self, lo: int, hi: int, lazy: bool = False):
        """zero_grad restricted to [lo, hi) of the flat gradient buffer (a step whose halves are zeroed at different points, see
        FusedAdamW.launch(groups=...)); ``lazy`` as in zero_grad: the lazy span is skipped.  Does not start a new gradient epoch."""
        self.fresh_left = set()                                   # (the half-step plans zero every range they own: no first writers)
        self.fold = None
        span = getattr(self, "lazy_span", None) if lazy else None
        if span is None or span[1] <= lo or span[0] >= hi:
            if hi > lo:
                ops.zero_(self.grad[lo:hi])
            return
        a, b = max(span[0], lo), min(span[1], hi)
        if a > lo:
            ops.zero_(self.grad[lo:a])
        if hi > b:
            ops.zero_(self.grad[b:hi])

    def take_fresh(self) -> bool:
        """True once after zero_grad(lazy=True): the lazy span holds stale values and must be overwritten (or zeroed) now"""
        fresh = getattr(self, "lazy_fresh", False)
        self.lazy_fresh = False
        return fresh

    def ensure_state(self):
        if self.exp_avg is None:
            self.exp_avg = torch.zeros_like(self.master)
            self.exp_avg_sq = torch.zeros_like(self.master)
