"""Fused AdamW over the flat parameter buffers + the reference's LR schedule.

Semantics follow src/optimization/adamw.py:40-103 (eps 1e-6, bias correction, decoupled weight decay
applied after the Adam update with the un-corrected lr), the 8 param groups of
src/optimization/utils.py:96-161 and the global-norm clipping of run_video_retrieval.py:477-482.
One ``cb_sq_sum`` + one ``cb_adamw`` launch per non-empty group; hyper-parameters travel through a
small DEVICE array so a captured hipGraph replays with fresh lr / step / clip values.
"""
from typing import List, Optional, Sequence

import torch

from . import ops
from ._lib import HP_COUNT, HP_SKIP
from .params import N_GROUPS, ParamBank


def warmup_linear(step: int, warmup_step: int, tot_step: int) -> float:
    """src/optimization/sched.py:14-17."""
    if step < warmup_step:
        return step / warmup_step
    return max(0, (tot_step - step) / (tot_step - warmup_step))


def multi_step_schedule(n_epoch: int, milestones, gamma: float = 0.5) -> float:
    """src/optimization/sched.py:20-25: gamma**i before the i-th (sorted) milestone, gamma**(len+1) after the last one."""
    ms = sorted(milestones)
    for i, m in enumerate(ms):
        if n_epoch < m:
            return gamma ** i
    return gamma ** (len(ms) + 1)


def get_lr_sched(global_step: int, decay: str, learning_rate: float, num_train_steps: int, warmup_ratio: float = 0.1,
                 decay_epochs=(), multi_step_epoch: int = -1) -> float:
    """src/optimization/sched.py:28-47 ('linear' | 'invsqrt' | 'constant' | 'multi_step'); floored to 1e-8."""
    warmup_steps = int(warmup_ratio * num_train_steps)
    if decay == "linear":
        lr = learning_rate * warmup_linear(global_step, warmup_steps, num_train_steps)
    elif decay == "invsqrt":
        lr = learning_rate * (global_step / warmup_steps if global_step <= warmup_steps
                              else (warmup_steps ** 0.5) * (global_step ** -0.5))
    elif decay == "constant":
        lr = learning_rate
    elif decay == "multi_step":
        assert multi_step_epoch >= 0
        lr = learning_rate * multi_step_schedule(multi_step_epoch, decay_epochs)
    else:
        raise ValueError(f"unsupported decay {decay}")
    return lr if lr > 0 else 1e-8


class FusedAdamW:
    """``param_groups`` mirrors the reference list (8 dicts with 'lr' and 'weight_decay') so runner code
    that assigns ``optimizer.param_groups[i]['lr']`` (run_video_retrieval.py:455-467) works unchanged."""

    def __init__(self, bank: ParamBank, lr: float = 5e-5, betas=(0.9, 0.98), eps: float = 1e-6, weight_decay: float = 1e-3,
                 cnn_lr: Optional[float] = None, cnn_weight_decay: Optional[float] = None, transformer_lr_mul: float = 1.0,
                 cnn_lr_mul: float = 1.0, max_grad_norm: float = -1.0, fold_norm: bool = True):
        """fold_norm: after ``zero_grad(lazy=True)`` the weight-gradient launches leave their shares of the squared gradient norm in slots
        (ParamBank.enable_norm_fold) and ``launch()`` adds them up instead of reading every gradient again -- single-process steps only
        (with a gradient exchange the norm is taken of the reduced gradients: ``launch(grad16=...)`` / ``pieces`` ignore the shares)."""
        self.bank = bank
        bank.clients += 1
        bank.ensure_state()
        self.betas, self.eps = betas, eps
        self.max_grad_norm = max_grad_norm
        cnn_lr = lr if cnn_lr is None else cnn_lr
        cnn_wd = weight_decay if cnn_weight_decay is None else cnn_weight_decay
        self.param_groups: List[dict] = []
        for g in range(N_GROUPS):
            is_cnn, top, nd = g >= 4, (g % 4) < 2, g % 2 == 1
            base = cnn_lr if is_cnn else lr
            mul = (cnn_lr_mul if is_cnn else transformer_lr_mul) if top else 1.0
            wd = 0.0 if nd else (cnn_wd if is_cnn else weight_decay)
            self.param_groups.append(dict(lr=base * mul, weight_decay=wd, range=bank.group_range[g]))
        self.step_count = 0
        dev = bank.device
        # hyper-parameter staging: TWO pinned host slots used alternately, each guarded by an event recorded after its
        # H2D copy -- the host may run ahead of the GPU by a step without overwriting values a pending copy still reads
        self._hp_host = torch.zeros(2, N_GROUPS, 16, dtype=torch.float32)
        if dev.type == "cuda":
            self._hp_host = self._hp_host.pin_memory()
        self._hp_events = [None, None]
        self._hp_dev = torch.zeros(N_GROUPS, 16, dtype=torch.float32, device=dev)
        # hyper-parameters of the PREVIOUS step (launch(prev=True): an update deferred into the next step's graph); until a step
        # exists behind it the skip flag makes that launch a no-op
        self._hp_dev_prev = torch.zeros(N_GROUPS, 16, dtype=torch.float32, device=dev)
        self._hp_dev_prev[:, HP_SKIP] = 1.0
        self._prev_live = False              # _hp_dev_prev holds a step's values (not the initial "skip")
        self.deferred_pending = False        # an update of some groups has been left to the next step (flush with launch(..., reuse_norm=True))
        self.fold_norm = bool(fold_norm) and max_grad_norm > 0 and bank.compute_dtype == torch.bfloat16
        if self.fold_norm:
            bank.enable_norm_fold()
        self._sq_own = torch.zeros(1, dtype=torch.float32, device=dev)
        self._sq = self._sq_own
        self._sq_ws = torch.zeros(1024, dtype=torch.float32, device=dev)     # partials of the order-independent norm reduction
        self._last_scale = 1.0

    def zero_grad(self, lazy: bool = False):
        """lazy=True (training loops whose backward always follows, e.g. tasks.train_step / bench.py): skip the memset of the
        encoder weight gradients; the batched weight-gradient launches overwrite them (ParamBank.set_lazy_span)."""
        self.bank.zero_grad(lazy=lazy)

    @torch.no_grad()
    def prepare_step(self, grad_scale: float = 1.0):
        """Host half of a step: advance the step count, pack lr / bias corrections / clip norm / grad_scale of the 8 groups
        and enqueue their (tiny) H2D copy on the current stream.  Never captured into a hipGraph: a training loop that
        replays a captured step calls ``prepare_step()`` eagerly before each replay and captures only ``launch()``."""
        if self._hp_dev.is_cuda and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("FusedAdamW.prepare_step() inside a hipGraph capture: capture launch() only and call "
                               "prepare_step() eagerly before each replay (see INTEGRATION.md)")
        if self.deferred_pending and self.step_count > 0:      # an update was left behind by the step before: it needs that step's values
            self._hp_dev_prev.copy_(self._hp_dev)              # D2D, stream-ordered before the upload below overwrites _hp_dev
            self._prev_live = True
        elif self._prev_live:                                  # nothing left behind: a launch(prev=True) is a no-op again
            self._hp_dev_prev[:, HP_SKIP] = 1.0
            self._prev_live = False
        self.step_count += 1
        self._last_scale = grad_scale
        slot = self.step_count & 1
        ev = self._hp_events[slot]
        if ev is not None:
            ev.synchronize()                       # the copy that last read this slot has executed
        host = self._hp_host[slot]
        for g, pg in enumerate(self.param_groups):
            hp = ops.adamw_hyper(pg["lr"], self.betas[0], self.betas[1], self.eps, pg["weight_decay"], self.step_count,
                                 self.max_grad_norm, grad_scale)
            host[g, :HP_COUNT] = torch.tensor(hp[:HP_COUNT])
        self._hp_dev.copy_(host, non_blocking=True)
        if self._hp_dev.is_cuda:
            ev = torch.cuda.Event()
            ev.record()
            self._hp_events[slot] = ev

    @torch.no_grad()
    def launch(self, grad16: Optional[torch.Tensor] = None, groups: Optional[Sequence[int]] = None, prev: bool = False,
               reuse_norm: bool = False, pieces=None, norm_reduce=None):
        """Device half of a step (capturable): global grad-norm reduction, then clip + AdamW + bf16 weight refresh, reading
        the hyper-parameters from the device array prepare_step() filled.  ``grad16``: consume these bf16 gradients (flat, same
        layout as bank.grad -- GradSync's reduced wire image, ``sync.wire_gradients()``) instead of the fp32 buffer.

        Software-pipelined update (bench.py's one-GPU plan): ``groups`` restricts the launch to some of the 8 parameter groups --
        e.g. the CNN groups (4-7) at the end of step i, and the transformer groups (0-3) at the START of step i+1's graph with
        ``prev=True`` (hyper-parameters of step i) and ``reuse_norm=True`` (the norm step i computed over ALL gradients), on a side
        stream beside the ResNet forward, which only reads CNN weights.  Same arithmetic, same order per parameter; a no-op until a
        step exists behind it (CB_HP_SKIP).  The caller zeroes each half of the gradients after its update (ParamBank.zero_grad_range).

        Owner-only update (GradSync(shard=True)): ``pieces`` = ``sync.owned_pieces()``, the (lo, hi) slices of the flat buffers whose
        reduced gradients this rank holds after the reduce-scatter; only they are updated, ``norm_reduce`` (``sync.norm_all_reduce``)
        sums the squared-norm partials over the ranks, and ``sync.gather_updated()`` afterwards distributes the new weights."""
        bank = self.bank
        if getattr(bank, "lazy_fresh", False) and not prev:
            raise RuntimeError("FusedAdamW: zero_grad(lazy=True) was not followed by an encoder backward -- the encoder weight "
                               "gradients were never written")
        bank.finish_fresh()                          # first-writer ranges nobody wrote (a partial backward) are zeroed before anything reads them
        sq = None
        if self.max_grad_norm > 0:
            fold = bank.fold_result() if (self.fold_norm and not reuse_norm and grad16 is None and pieces is None) else None
            if fold is not None:
                # the weight-gradient launches left their shares of the squared norm in slots (cb_gemm_desc.sq_slots): what remains is the
                # ranges they do not cover + the slots, added in a fixed order -- no second pass over ~5/6 of the gradient bytes.  The
                # accumulator was zeroed with the slots by zero_grad(lazy=True).
                self._sq = bank.sq_buf[:1]
                ops.sq_sum_fold(bank.grad, fold[0], fold[1], self._sq, self._sq_ws)
            elif not reuse_norm:
                self._sq = self._sq_own
                ops.zero_(self._sq)
                src = bank.grad if grad16 is None else grad16
                if pieces is None:
                    ops.sq_sum(src[:bank.n_train], self._sq, self._sq_ws)   # deterministic: ranks must derive the same clip coefficient
                else:
                    for lo, hi in pieces:                                   # partial of the owned slices (accumulated), then summed over ranks
                        ops.sq_sum(src[lo:hi], self._sq, self._sq_ws)
                    if norm_reduce is not None:
                        norm_reduce(self._sq)
            sq = self._sq
        hp_dev = self._hp_dev_prev if prev else self._hp_dev
        if pieces is not None:
            bank.owner_only_dirty = True          # masters / moments outside the owned pieces are stale until GradSync.gather_state()
        for g, pg in enumerate(self.param_groups):
            a, b = pg["range"]
            if b <= a or (groups is not None and g not in groups):
                continue
            gsrc = bank.grad if grad16 is None else grad16
            spans = [(a, b)] if pieces is None else [(max(a, lo), min(b, hi)) for lo, hi in pieces if lo < b and a < hi]
            for x, y in spans:
                w16 = bank.w16[x:y] if bank.w16 is not None else None
                ops.adamw(bank.master[x:y], gsrc[x:y], bank.exp_avg[x:y], bank.exp_avg_sq[x:y], w16, hp_dev[g], sq)

    def step(self, grad_scale: float = 1.0, grad16: Optional[torch.Tensor] = None, pieces=None, norm_reduce=None):
        """grad_scale multiplies the gradients first (1/world_size after a SUM all-reduce)."""
        self.prepare_step(grad_scale)
        self.launch(grad16=grad16, pieces=pieces, norm_reduce=norm_reduce)

    def grad_norm(self) -> float:
        """Host-visible global norm of the (averaged) gradient of the last step (syncs)."""
        return float(self._sq.sqrt().item()) * self._last_scale

    # ---- checkpointing (the reference saves optimizer.state_dict() in *_train_state.pt / restore.pt) ------------------
    def state_dict(self) -> dict:
        """{"state": {parameter name: {"step", "exp_avg", "exp_avg_sq"}}, "param_groups": [...], "step": n}: moments in the
        parameters' LOGICAL shapes (OIHW for convs), keyed by name so that the file does not depend on the flat layout."""
        bank = self.bank
        bank.assert_whole("FusedAdamW.state_dict()")
        if self.deferred_pending:
            raise RuntimeError("FusedAdamW.state_dict(): an update deferred to the next step is pending -- flush it first "
                               "(launch(groups=..., reuse_norm=True); deferred_pending = False)")
        state = {}
        for name, p in bank._trainable:
            off = bank.offset[id(p)]
            state[name] = dict(step=self.step_count, exp_avg=bank._view(bank.exp_avg, off, p).detach().clone().contiguous(),
                               exp_avg_sq=bank._view(bank.exp_avg_sq, off, p).detach().clone().contiguous())
        groups = [dict(lr=pg["lr"], weight_decay=pg["weight_decay"], betas=self.betas, eps=self.eps) for pg in self.param_groups]
        return dict(state=state, param_groups=groups, step=self.step_count)

    @torch.no_grad()
    def load_state_dict(self, sd: dict):
        bank = self.bank
        missing = [n for n, _p in bank._trainable if n not in sd["state"]]
        if missing:
            raise KeyError(f"optimizer state lacks {len(missing)} parameters, e.g. {missing[:3]}")
        for name, p in bank._trainable:
            off = bank.offset[id(p)]
            st = sd["state"][name]
            bank._view(bank.exp_avg, off, p).copy_(st["exp_avg"].to(bank.device, torch.float32))
            bank._view(bank.exp_avg_sq, off, p).copy_(st["exp_avg_sq"].to(bank.device, torch.float32))
        self.step_count = int(sd.get("step", 0))
        for pg, src in zip(self.param_groups, sd.get("param_groups", [])):
            pg["lr"], pg["weight_decay"] = src["lr"], src["weight_decay"]
