"""Data-parallel gradient exchange over RCCL (torch.distributed backend "nccl" on ROCm; "gloo" on CPU).

The reference averages every parameter gradient with Horovod/NCCL, one tensor at a time
(run_video_retrieval.py:298-301,432).  Here the gradients already live in ONE flat fp32 buffer laid
out transformer-groups-first, so the exchange is two large bucket all-reduces:

  bucket 0 = transformer parameters: issued from the end of the encoder backward, i.e. it travels over
             xGMI on RCCL's side stream WHILE the ResNet backward (the larger half of the step) runs;
  bucket 1 = CNN parameters: issued after the CNN backward.

Sums are averaged inside the fused AdamW (grad_scale = 1/world), so no extra pass touches the buffer.
The dead detectron2 RPN/ROI parameters the reference also all-reduces do not exist here.
"""
from typing import List, Optional

import torch
import torch.distributed as dist

from .params import ParamBank


class NativeComm:
    """The C ABI's communicator (include/clipbert_hip.h: cb_comm_init / cb_allreduce_bucket): RCCL called by libclipbert_hip
    itself on a stream the caller chooses, instead of torch.distributed's process-group call.  torch.distributed (any backend)
    is only the channel that carries rank 0's 128-byte unique id to the other ranks.  One per process."""
    _instance = None

    def __init__(self, rank: int, world: int, unique_id: bytes):
        import ctypes
        from . import _lib
        assert len(unique_id) == 128
        self._buf = ctypes.create_string_buffer(unique_id, 128)
        _lib.check(_lib.get().cb_comm_init(rank, world, ctypes.cast(self._buf, ctypes.c_void_p)), "cb_comm_init")
        self.rank, self.world = rank, world

    @staticmethod
    def unique_id() -> bytes:
        import ctypes
        from . import _lib
        buf = ctypes.create_string_buffer(128)
        _lib.check(_lib.get().cb_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p)), "cb_comm_unique_id")
        return buf.raw

    @classmethod
    def from_process_group(cls, group=None) -> "NativeComm":
        """every rank calls this with its GPU current (torch.cuda.set_device)"""
        if cls._instance is not None:
            if dist.is_initialized():
                assert (cls._instance.rank, cls._instance.world) == (dist.get_rank(group), dist.get_world_size(group)), \
                    "NativeComm: one communicator per process -- destroy() it before creating one for another group"
            return cls._instance
        if dist.is_initialized():
            rank, world = dist.get_rank(group), dist.get_world_size(group)
            # rank 0 ALWAYS enters the broadcast: if it cannot even create the id (library without RCCL, dlopen failure) it sends the
            # error instead, and every rank raises together -- nobody is left waiting in the collective
            box = [None]
            if rank == 0:
                try:
                    box = [cls.unique_id()]
                except Exception as e:                           # noqa: BLE001
                    box = [("error", repr(e))]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            if not isinstance(box[0], (bytes, bytearray)):
                raise RuntimeError(f"NativeComm: rank 0 could not create the RCCL unique id: {box[0][1] if box[0] else 'nothing received'}")
        else:
            rank, world, box = 0, 1, [cls.unique_id()]
        if world == 1:
            cls._instance = cls(rank, world, box[0])
            return cls._instance
        # Several ranks: ncclCommInitRank and a first collective of THIS library's communicator run in a helper thread with a deadline,
        # the result (sum of rank + 1 over the ranks) is checked, and the ranks agree over the process group whether everybody passed:
        # a communicator that cannot be created, hangs or adds wrongly makes every rank raise together (GradSync(comm="auto") then lets
        # torch.distributed carry the buckets) instead of some ranks proceeding and the job dead-locking in its first bucket.
        dev = torch.cuda.current_device()

        def bring_up():
            torch.cuda.set_device(dev)                         # (the current device is per thread)
            comm = cls(rank, world, box[0])
            probe = torch.full((1024,), float(rank + 1), dtype=torch.float32, device=f"cuda:{dev}")
            comm.all_reduce_(probe)
            torch.cuda.synchronize(dev)
            return comm, bool((probe == world * (world + 1) / 2).all().item())

        def agree(ok: bool) -> bool:
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=f"cuda:{dev}" if dist.get_backend(group) == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            return int(flag.item()) == 1

        import os
        try:
            cls._instance = cls._bring_up_guarded(bring_up, agree, float(os.environ.get("CB_COMM_INIT_TIMEOUT", "120")), rank)
        except RuntimeError:
            # ranks whose own bring-up succeeded still hold a communicator inside the library: drop it, so that a later
            # cb_comm_init (another attempt, another group) starts clean
            try:
                from . import _lib
                _lib.get().cb_comm_destroy()
            except Exception:                                    # noqa: BLE001
                pass
            raise
        return cls._instance

    @staticmethod
    def _bring_up_guarded(bring_up, agree, timeout_s: float, rank: int = 0):
        """``bring_up() -> (communicator, probe_ok)`` in a helper thread with a deadline; ``agree(ok) -> bool`` = AND over the ranks
        (a collective of the process group: every rank calls it exactly once, whatever happened locally).  Returns the communicator
        when every rank succeeded; raises RuntimeError on every rank otherwise."""
        import threading
        out = {}

        def run():
            try:
                out["comm"], out["ok"] = bring_up()
            except Exception as e:                               # noqa: BLE001
                out["err"] = e

        th = threading.Thread(target=run, daemon=True, name="cb_comm_init")
        th.start()
        th.join(timeout=timeout_s)
        ok = bool(out.get("ok", False))
        if not agree(ok):
            if ok:
                why = "failed on another rank"
            elif "err" in out:
                why = repr(out["err"])
            elif th.is_alive():
                why = f"no answer within {timeout_s:.0f} s"
            else:
                why = "the probe all-reduce returned a wrong sum"
            raise RuntimeError(f"NativeComm: bring-up of the library's RCCL communicator failed on at least one rank (rank {rank}: {why})")
        return out["comm"]

    def all_reduce_(self, t: torch.Tensor, stream=None):
        """in-place sum over the ranks, enqueued on `stream` (default: the current stream)"""
        from . import _lib
        from .ops import dtype_code
        assert t.is_cuda and t.is_contiguous() and t.dtype in (torch.float32, torch.bfloat16)
        st = stream if stream is not None else torch.cuda.current_stream(t.device)
        _lib.check(_lib.get().cb_allreduce_bucket(t.data_ptr(), t.numel(), dtype_code(t.dtype), st.cuda_stream), "cb_allreduce_bucket")
        return t

    def reduce_scatter_(self, t: torch.Tensor, stream=None) -> torch.Tensor:
        """in place: t holds world equal shards; returns the view of THIS rank's shard, which receives the sum over the ranks"""
        from . import _lib
        from .ops import dtype_code
        assert t.is_cuda and t.is_contiguous() and t.numel() % self.world == 0
        n = t.numel() // self.world
        mine = t[self.rank * n:(self.rank + 1) * n]
        st = stream if stream is not None else torch.cuda.current_stream(t.device)
        _lib.check(_lib.get().cb_reduce_scatter_bucket(t.data_ptr(), mine.data_ptr(), n, dtype_code(t.dtype), st.cuda_stream), "cb_reduce_scatter_bucket")
        return mine

    def all_gather_(self, t: torch.Tensor, stream=None) -> torch.Tensor:
        """in place: every rank's shard of t (its 1/world slice) is delivered to all ranks"""
        from . import _lib
        from .ops import dtype_code
        assert t.is_cuda and t.is_contiguous() and t.numel() % self.world == 0
        n = t.numel() // self.world
        st = stream if stream is not None else torch.cuda.current_stream(t.device)
        _lib.check(_lib.get().cb_allgather_bucket(t.data_ptr() + self.rank * n * t.element_size(), t.data_ptr(), n, dtype_code(t.dtype),
                                                  st.cuda_stream), "cb_allgather_bucket")
        return t

    def broadcast_(self, t: torch.Tensor, root: int = 0, stream=None) -> torch.Tensor:
        from . import _lib
        from .ops import dtype_code
        assert t.is_cuda and t.is_contiguous()
        st = stream if stream is not None else torch.cuda.current_stream(t.device)
        _lib.check(_lib.get().cb_broadcast_bucket(t.data_ptr(), t.numel(), dtype_code(t.dtype), root, st.cuda_stream), "cb_broadcast_bucket")
        return t

    @classmethod
    def destroy(cls):
        from . import _lib
        if cls._instance is not None:
            _lib.check(_lib.get().cb_comm_destroy(), "cb_comm_destroy")
            cls._instance = None


class GradSync:
    """``compress="bf16"`` sends the gradients as bf16 (half the xGMI payload: 297 MB instead of 594 MB per step; the
    reference's apex-O2 gradients are fp16 on the wire too): cast -> all-reduce -> cast back, three passes over the flat
    buffer (~0.3 ms) against ~1.5 ms of link time saved on 8 GPUs."""

    def __init__(self, bank: ParamBank, group=None, bucket_bytes: int = 64 << 20, compress: Optional[str] = None, comm: str = "auto",
                 pretend_world: int = 0, loopback: bool = False, shard: bool = False):
        """bucket_bytes: fp32 gradient bytes per all-reduce (0 = one collective per range).  The default 64 MiB (32 MiB on
        the wire in bf16) lets the cast of bucket i+1 run while bucket i is on the links, and keeps each collective well
        past the ~8 MiB where RCCL's ring reaches its link bandwidth."""
        assert compress in (None, "bf16") and comm in ("auto", "torch", "native")
        # comm="native": the buckets go through cb_allreduce_bucket (RCCL called by the library) on this object's own HIP stream,
        # ordered against the compute stream by events; "torch": torch.distributed all_reduce(async_op=True) (also the CPU / gloo path).
        # "auto" (default): native when the process group is RCCL ("nccl") and the gradients live on a GPU -- if the library's
        # communicator cannot be created the process group carries the buckets instead (self.carrier says which one runs).
        self.native = None
        self.carrier = "torch"
        self.bank = bank
        bank.clients += 1
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # pretend_world > 1 (single process, no process group): everything a rank of an N-rank job does EXCEPT the collectives --
        # wire casts, bucket bookkeeping, 1/N scaling, bf16-direct AdamW -- so the non-link overhead of the DP plan can be timed on one GPU
        self.dry = pretend_world > 1 and self.world == 1
        if self.dry:
            self.world = pretend_world
        # loopback (single process, one GPU): the complete N-rank plan with REAL collectives on a world-size-1 communicator of the
        # library -- every cb_allreduce_bucket is issued (and can be captured into a hipGraph) although it adds nothing
        self.loopback = bool(loopback) and self.world == 1 and not self.dry
        if self.world > 1 or self.loopback or self.dry:
            # the gradient norm of a data-parallel step is the norm of the EXCHANGED gradients: the per-launch shares of the local ones
            # (ParamBank.enable_norm_fold) say nothing about it -- FusedAdamW.launch runs its pass over the reduced gradients
            bank.norm_fold_blocked = True
        t_end = bank.group_range[3][1]
        self.t_range = (0, t_end)
        self.c_range = (t_end, bank.n_train)
        self.bucket_elems = bucket_bytes // 4 if bucket_bytes > 0 else 0
        if shard:
            q = 64 * max(self.world, 1)
            assert ParamBank.GROUP_ALIGN % q == 0, f"shard=True needs world in (1, 2, 4, 8): {self.world}"
            self.bucket_elems = max(q, self.bucket_elems // ParamBank.GROUP_ALIGN * ParamBank.GROUP_ALIGN) if self.bucket_elems > 0 else 0
        self.compress = compress
        self._wire = None
        self._work: List = []
        self._pending: List = []            # (start, end) ranges whose bf16 wire image must be cast back after wait()
        self._inflight: List = []           # ranges handed to _reduce since the last wait()
        self.c_early: List = []             # set_cnn_split(): sub-ranges of the CNN range whose gradients are final early
        self.late_ranges = 0                # ranges wait() had to send itself because nobody reduced them (diagnostic)
        # shard=True ("owner-only update", the two halves of the all-reduce with the optimizer in between): every bucket
        # is REDUCE-SCATTERED -- rank r ends up with the sum of the r-th 1/world of each bucket --, FusedAdamW.launch(pieces=
        # sync.owned_pieces()) updates only those pieces, gather_updated() ALL-GATHERS the new compute weights (and the fp32 masters
        # of the groups kernels read in fp32).  Same bytes on the links as the all-reduce; the optimizer streams 1/world of the state.
        self.shard = bool(shard)
        # mute (diagnostic, set by bench.py AFTER its timed region): every collective of the plan is skipped while everything else a
        # rank does stays -- the step time with and without the links gives the EXPOSED communication time of the plan
        self.mute = False
        self._owned_done: List = []         # ... of the exchange wait() completed last (what the optimizer / gather_updated use)
        self._buckets_done: List = []
        self._owned: List = []              # (lo, hi) pieces of the flat buffers this rank owns, from the buckets reduced since wait()
        self._buckets: List = []            # (s, e) of every bucket of the current exchange
        self._epoch_done = getattr(bank, "grad_epoch", 0)    # bank.grad_epoch (one per zero_grad) of the last completed exchange
        self._grid = None                   # bucket_grid(): fixed at first use
        self._grid_used = False
        want_native = comm == "native" or (comm == "auto" and dist.is_initialized() and bank.grad.is_cuda and dist.get_backend(group) == "nccl")
        want_native = want_native or self.loopback
        if want_native and (self.world > 1 or self.loopback) and not self.dry:
            assert bank.grad.is_cuda, "comm='native' needs the gradients on a GPU"
            try:
                self.native = NativeComm.from_process_group(group)
                self.comm_stream = torch.cuda.Stream(device=bank.grad.device)
                self.carrier = "native"
            except Exception as e:                        # noqa: BLE001  (comm="native" was asked for explicitly: fail loudly)
                if comm == "native" or self.loopback:
                    raise
                import warnings
                warnings.warn(f"GradSync: the library's RCCL communicator could not be created ({e}); torch.distributed carries the buckets")
                self.native = None

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world

    def _ensure_wire(self):
        if self._wire is None and self.compress == "bf16":
            self._wire = torch.empty(self.bank.n_train, dtype=torch.bfloat16, device=self.bank.grad.device)

    def cast_range(self, a: int, b: int):
        """fp32 gradients [a, b) -> the bf16 wire image (no communication: capturable into a hipGraph, so that a replay plan can
        end each of its graphs with the cast of the range it just finished and issue only the collectives eagerly)."""
        if not self.active or b <= a or self.compress != "bf16":
            return
        from . import ops
        self._ensure_wire()
        if self.bank.grad.is_cuda:
            ops.cast(self.bank.grad[a:b], self._wire[a:b])
        else:
            self._wire[a:b].copy_(self.bank.grad[a:b])

    def set_cnn_split(self, res5_start: Optional[int]):
        """The CNN range is [grid_encoder | res3 | res4 | res5] (the reference's parameter-group order); its two ends -- grid_encoder
        and res5, ~3/4 of the bytes -- are final when the ResNet backward has passed res5 (modeling.cnn_backward_steps yields there).
        ``res5_start`` = modeling.cnn_early_split(model); None switches the split off."""
        new = []
        if res5_start is not None:
            lo, hi = self.c_range
            mid = self.bank.group_range[6][0]           # end of the grid_encoder groups = start of the backbone group
            assert lo <= mid <= res5_start <= hi, (lo, mid, res5_start, hi)
            new = [r for r in ((lo, mid), (res5_start, hi)) if r[1] > r[0]]
        if new != self.c_early:
            # the split moves the bucket grid, i.e. (shard=True) which rank owns which elements: the AdamW moments and fp32 masters
            # never travel between steps, so the grid may only change while no owner-only update has happened on it
            assert not (self.shard and self._grid_used), \
                "GradSync.set_cnn_split after an owner-only exchange: the ownership of optimizer state is fixed by the first one"
            self.c_early = new
            self._grid = None

    def bucket_grid(self):
        """The FIXED list of (start, end) buckets of the flat gradient buffer: the canonical ranges (transformer; the CNN range cut at
        the set_cnn_split points) each divided from ITS start in steps of bucket_elems.  Every exchange -- the overlap hooks, reduce_cnn
        with or without the early part, the safety net of wait() -- is made of whole buckets of this grid, whatever range it was
        issued for: with shard=True the owner of an element (and of its AdamW moments / fp32 master) therefore never changes."""
        if self._grid is None:
            cuts = sorted({self.t_range[0], self.t_range[1], self.c_range[1]} | {x for r in self.c_early for x in r})
            grid = []
            for a, b in zip(cuts, cuts[1:]):
                step = self.bucket_elems if self.bucket_elems > 0 else (b - a)
                for s0 in range(a, b, max(step, 1)):
                    grid.append((s0, min(b, s0 + step)))
            self._grid = grid
        return self._grid

    def attach(self, model):
        """Arm the overlap hooks of a prepared ClipBert: the transformer buckets leave from the end of the last encoder backward of a
        step, grid_encoder + res5 from inside the last ResNet backward (what INTEGRATION.md spells out line by line)."""
        from .modeling import cnn_early_split
        rt = model.rt
        assert rt is not None and rt.bank is self.bank, "GradSync.attach: the model is not prepared on this sync's parameter bank"
        rt.after_encoder_backward = self.reduce_transformer
        self.set_cnn_split(cnn_early_split(model))
        rt.after_res5_backward = self.reduce_cnn_early if self.c_early else None
        return self

    def _cnn_late(self):
        """the part of the CNN range that set_cnn_split() does not declare early"""
        out, cur = [], self.c_range[0]
        for a, b in sorted(self.c_early):
            if a > cur:
                out.append((cur, a))
            cur = max(cur, b)
        if cur < self.c_range[1]:
            out.append((cur, self.c_range[1]))
        return out

    def cast_cnn_early(self):
        for a, b in self.c_early:
            self.cast_range(a, b)

    def reduce_cnn_early(self, cast: bool = True):
        """issue the exchange of grid_encoder + res5 (call where cnn_backward_steps yields / from rt.after_res5_backward)"""
        for a, b in self.c_early:
            self._reduce(a, b, cast=cast)

    def cast_transformer(self):
        self.cast_range(*self.t_range)

    def cast_cnn(self, late_only: bool = False):
        for a, b in (self._cnn_late() if late_only else [self.c_range]):
            self.cast_range(a, b)

    @property
    def active(self) -> bool:
        """an exchange takes place (several ranks, the dry run of their plan, or the one-GPU loopback)"""
        return self.world > 1 or self.loopback

    def _reduce(self, a: int, b: int, cast: bool = True):
        if not self.active or b <= a:
            return
        for s0, e0 in self._inflight:
            assert e0 <= a or b <= s0, (f"GradSync: gradients [{a}, {b}) are already being reduced ([{s0}, {e0})): "
                                        "wait() before reducing a range again (one exchange per optimizer step)")
        self._inflight.append((a, b))
        buckets = [(s, e) for s, e in self.bucket_grid() if a <= s and e <= b]
        covered = sum(e - s for s, e in buckets)
        if covered != b - a or (buckets and (buckets[0][0] != a or buckets[-1][1] != b)):
            raise RuntimeError(f"GradSync: [{a}, {b}) is not a union of whole buckets of the fixed grid (canonical cuts "
                               f"{sorted({x for r in [self.t_range, self.c_range] + list(self.c_early) for x in r})}): reduce the transformer / CNN ranges "
                               "(or their set_cnn_split parts), not arbitrary slices")
        self._grid_used = True
        for s, e in buckets:
            if self.shard:
                assert (e - s) % (64 * self.world) == 0, "shard=True: ranges must start / end at multiples of world x 64 elements (ParamBank.GROUP_ALIGN)"
                n = (e - s) // self.world
                self._owned.append((s + self.rank * n, s + (self.rank + 1) * n))
                self._buckets.append((s, e))
            if self.compress == "bf16":
                if cast:
                    self.cast_range(s, e)                          # (per bucket: the cast of bucket i+1 overlaps bucket i's transfer)
                self._ensure_wire()
                self._work.append(self._exchange(self._wire[s:e]))
                self._pending.append((s, e))
            else:
                self._work.append(self._exchange(self.bank.grad[s:e]))

    def _uncovered(self):
        """parts of [0, n_train) not handed to _reduce since the last wait()"""
        out, cur = [], 0
        for a, b in sorted(self._inflight):
            if a > cur:
                out.append((cur, a))
            cur = max(cur, b)
        if cur < self.bank.n_train:
            out.append((cur, self.bank.n_train))
        return out

    def _exchange(self, t: torch.Tensor):
        """one bucket: all-reduce, or (shard=True) reduce-scatter in place -- this rank's 1/world slice receives the sum"""
        if not self.shard:
            return self._all_reduce(t)
        if self.dry or self.mute:
            return None
        if self.native is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream(t.device))
            self.native.reduce_scatter_(t, self.comm_stream)
            return None
        if self.world == 1:
            return None
        n = t.numel() // self.world
        if t.is_cuda and dist.get_backend(self.group) == "nccl":
            return dist.reduce_scatter_tensor(t[self.rank * n:(self.rank + 1) * n], t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)   # gloo has no reduce-scatter: the owned slice is what is read

    def owned_pieces(self):
        """shard=True, after wait(): the (lo, hi) pieces of the flat buffers whose gradients THIS rank holds reduced"""
        assert self.shard
        return list(self._owned_done)

    def norm_all_reduce(self, sq: torch.Tensor):
        """sum over the ranks of the squared-gradient partial of the owned pieces (every rank then derives the same clip coefficient)"""
        if not self.active or self.dry or self.mute or (self.world == 1 and not self.loopback):
            return sq
        if self.native is not None:
            self.native.all_reduce_(sq)                   # (on the current stream, between the partial sums and the update)
        else:
            dist.all_reduce(sq, op=dist.ReduceOp.SUM, group=self.group)
        return sq

    def gather_updated(self, masters: bool = False):
        """shard=True, after the optimizer ran on the owned pieces: every bucket's new bf16 compute weights (and the fp32 masters of
        the no-decay groups -- biases and LayerNorm vectors, which kernels read in fp32; ``masters=True``: of everything, e.g. before
        state_dict()) are all-gathered in place."""
        assert self.shard
        if not self.active or self.dry:
            return
        bank = self.bank
        # what kernels read in fp32: every 1-D parameter (biases, LayerNorm / BatchNorm vectors -- the regression head's
        # BatchNorm1d.weight sits in a DECAY group) plus, as before, whole no-decay groups
        nd = [bank.group_range[g] for g in (1, 3, 5, 7)] + bank.fp32_read_ranges()
        for s, e in self._buckets_done:
            if bank.w16 is not None:
                self._gather(bank.w16[s:e])
            fp32_read = masters or bank.w16 is None or any(a < e and s < b for a, b in nd)
            if fp32_read:
                self._gather(bank.master[s:e])
        if self.native is not None:
            torch.cuda.current_stream(bank.grad.device).wait_stream(self.comm_stream)

    def gather_state(self, optimizer=None):
        """shard=True, before anything reads the WHOLE state (state_dict(), ModelSaver / E2E_TrainingRestorer.save() on rank 0): every
        rank calls this; the fp32 masters and (``optimizer``: its) AdamW moments of every bucket are all-gathered from their owners."""
        assert self.shard
        bank = self.bank
        if self.active and not self.dry:
            for s, e in self._buckets_done:
                self._gather(bank.master[s:e])
                if optimizer is not None:
                    self._gather(bank.exp_avg[s:e])
                    self._gather(bank.exp_avg_sq[s:e])
            if self.native is not None:
                torch.cuda.current_stream(bank.grad.device).wait_stream(self.comm_stream)
        if optimizer is not None or not self.active:
            bank.owner_only_dirty = False

    def _gather(self, t: torch.Tensor):
        if self.mute:
            return
        if self.native is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream(t.device))
            self.native.all_gather_(t, self.comm_stream)
            return
        if self.world == 1:
            return
        n = t.numel() // self.world
        if t.is_cuda and dist.get_backend(self.group) != "nccl":           # gloo carries GPU tensors for all_reduce / broadcast only
            mine = t[self.rank * n:(self.rank + 1) * n].clone()
            t.zero_()
            t[self.rank * n:(self.rank + 1) * n].copy_(mine)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            return
        if t.is_cuda:
            dist.all_gather_into_tensor(t, t[self.rank * n:(self.rank + 1) * n].clone(), group=self.group)
            return
        parts = [torch.empty_like(t[:n]) for _ in range(self.world)]
        dist.all_gather(parts, t[self.rank * n:(self.rank + 1) * n].contiguous(), group=self.group)
        for r, p in enumerate(parts):
            t[r * n:(r + 1) * n].copy_(p)

    def _all_reduce(self, t: torch.Tensor):
        if self.dry or self.mute:
            return None
        if self.native is None:
            return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        # the bucket was produced on the current stream: the comm stream waits for exactly that point, then carries the collective
        self.comm_stream.wait_stream(torch.cuda.current_stream(t.device))
        self.native.all_reduce_(t, self.comm_stream)
        return None

    def reduce_transformer(self, cast: bool = True):
        """cast=False: the wire image of the range was already produced by cast_transformer() (e.g. inside a captured graph)"""
        self._reduce(*self.t_range, cast=cast)

    def reduce_cnn(self, cast: bool = True):
        """whatever of the CNN range is not already in flight (reduce_cnn_early may have sent its two ends)"""
        early_sent = bool(self.c_early) and all(r in self._inflight for r in self.c_early)
        for a, b in (self._cnn_late() if early_sent else [self.c_range]):
            self._reduce(a, b, cast=cast)

    def wire_gradients(self) -> Optional[torch.Tensor]:
        """The flat bf16 gradient image (valid after wait(cast_back=False) once BOTH ranges were reduced): hand it to
        FusedAdamW.step / launch(grad16=...) and the two cast-back passes over the 594 MB fp32 buffer disappear."""
        return self._wire if (self.compress == "bf16" and self.active) else None

    def wait(self, cast_back: bool = True):
        """cast_back=False (bf16 wire only): leave the reduced gradients in the wire buffer for an optimizer that reads bf16
        (wire_gradients()); bank.grad then still holds THIS rank's un-reduced fp32 gradients."""
        # safety net: a hook that did not fire (e.g. a forward whose backward never ran left a pending-node count behind) must not
        # leave part of the gradient buffer un-exchanged -- whatever is missing goes out now, late but correct
        # -- including a step for which NO reduce_* was called at all (hooks not armed, caller forgot): the gradient epoch of the
        # bank (one per zero_grad) tells that case from a repeated wait() with nothing left to do
        epoch = getattr(self.bank, "grad_epoch", 0)
        if self.active and (self._inflight or epoch != self._epoch_done):
            for a, b in self._uncovered():
                self.late_ranges += 1
                self._reduce(a, b)
        self._epoch_done = epoch
        for w in self._work:
            if w is not None:
                w.wait()
        if self.native is not None and self._work:
            torch.cuda.current_stream(self.bank.grad.device).wait_stream(self.comm_stream)
        self._work = []
        self._inflight = []
        if self.shard and self._buckets:               # (a repeated wait() of the same step keeps the pieces of the exchange it completed)
            # one exchange = every bucket of the fixed grid exactly once: the owned pieces are the same set every step
            if sorted(self._buckets) != self.bucket_grid():
                raise RuntimeError("GradSync(shard=True): the exchange of this step did not cover the bucket grid exactly once "
                                   f"({len(self._buckets)} buckets issued, {len(self.bucket_grid())} in the grid)")
            self._owned_done, self._buckets_done, self._owned, self._buckets = sorted(self._owned), sorted(self._buckets), [], []
        if not cast_back and self.compress == "bf16":
            self._pending = []
            return
        for s, e in self._pending:
            if self.bank.grad.is_cuda:
                from . import ops
                ops.cast(self._wire[s:e], self.bank.grad[s:e])
            else:
                self.bank.grad[s:e].copy_(self._wire[s:e])
        self._pending = []

    def _broadcast(self, t: torch.Tensor, src: int):
        if self.native is not None:                      # cb_broadcast_bucket on the current stream
            self.native.broadcast_(t, src)
        else:
            dist.broadcast(t, dist.get_global_rank(self.group, src) if self.group is not None else src, group=self.group)

    def broadcast_parameters(self, src: int = 0):
        """hvd.broadcast_parameters equivalent (run_video_retrieval.py:304): one flat buffer per kind."""
        if not self.active or self.dry:
            return
        self.bank.assert_whole("GradSync.broadcast_parameters()")
        self._broadcast(self.bank.master, src)
        self._broadcast(self.bank.f_master, src)
        self.bank.sync_compute()

    def broadcast_state(self, optimizer, src: int = 0):
        """Parameters AND optimizer state of rank ``src`` on every rank (hvd.broadcast_parameters + hvd.broadcast_optimizer_state,
        run_video_retrieval.py:304-305 / the restorer's restore on every rank): fp32 masters, both AdamW moments and the step count
        (the bias corrections depend on it) -- ranks that resume from different states would otherwise drift apart."""
        self.broadcast_parameters(src)
        if not self.active or self.dry:
            return
        bank = self.bank
        bank.ensure_state()
        self._broadcast(bank.exp_avg, src)
        self._broadcast(bank.exp_avg_sq, src)
        box = [int(optimizer.step_count) if self.rank == src else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(self.group, src) if self.group is not None else src, group=self.group)
        optimizer.step_count = int(box[0])
