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
from ._lib import HP_COUNT
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
                 cnn_lr_mul: float = 1.0, max_grad_norm: float = -1.0):
        self.bank = bank
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
        self._hp_host = torch.zeros(N_GROUPS, 16, dtype=torch.float32)
        if dev.type == "cuda":
            self._hp_host = self._hp_host.pin_memory()
        self._hp_dev = torch.zeros(N_GROUPS, 16, dtype=torch.float32, device=dev)
        self._sq = torch.zeros(1, dtype=torch.float32, device=dev)

    def zero_grad(self):
        self.bank.zero_grad()

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0):
        """grad_scale multiplies the gradients first (1/world_size after a SUM all-reduce)."""
        bank = self.bank
        self.step_count += 1
        self._last_scale = grad_scale
        for g, pg in enumerate(self.param_groups):
            hp = ops.adamw_hyper(pg["lr"], self.betas[0], self.betas[1], self.eps, pg["weight_decay"], self.step_count,
                                 self.max_grad_norm, grad_scale)
            self._hp_host[g, :HP_COUNT + 1] = torch.tensor(hp[:HP_COUNT + 1])
        self._hp_dev.copy_(self._hp_host, non_blocking=True)
        sq = None
        if self.max_grad_norm > 0:
            self._sq.zero_()
            ops.sq_sum(bank.grad[:bank.n_train], self._sq)
            sq = self._sq
        for g, pg in enumerate(self.param_groups):
            a, b = pg["range"]
            if b <= a:
                continue
            w16 = bank.w16[a:b] if bank.w16 is not None else None
            ops.adamw(bank.master[a:b], bank.grad[a:b], bank.exp_avg[a:b], bank.exp_avg_sq[a:b], w16, self._hp_dev[g], sq)

    def grad_norm(self) -> float:
        """Host-visible global norm of the (averaged) gradient of the last step (syncs)."""
        return float(self._sq.sqrt().item()) * getattr(self, "_last_scale", 1.0)
