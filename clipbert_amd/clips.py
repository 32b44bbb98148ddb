"""Clip aggregation of the per-clip logits -- row a20 of the hot path.

The reference computes N_clip independent forwards and pools their logits in its task loops
(src/tasks/run_video_retrieval.py:396-418 training, :664-682 inference; src/tasks/run_video_qa.py:241-275).  Same names,
same argument meaning, same errors; the arithmetic runs in libclipbert_hip (cb_clip_aggregate_*, cb_lse_loss)."""
from typing import List

import torch

from . import ops

_MODES = {"mean": ops.AGG_MEAN, "max": ops.AGG_MAX}


class _AggFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, stack, mode):
        out, am = ops.clip_aggregate_fwd(stack, mode)
        ctx.mode, ctx.n = mode, stack.shape[0]
        ctx.save_for_backward(*(t for t in (stack, out, am) if t is not None))
        ctx.has_am = am is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        saved = ctx.saved_tensors
        stack, out = saved[0], saved[1]
        am = saved[2] if ctx.has_am else None
        return ops.clip_aggregate_bwd(dout.float(), stack, out, am, ctx.n, ctx.mode), None


class _LseLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, stack, labels):
        loss, _ = ops.lse_loss(stack, labels)
        ctx.save_for_backward(stack, labels)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        stack, labels = ctx.saved_tensors
        _, dx = ops.lse_loss(stack, labels, want_loss=False, dloss=dloss.float().contiguous(), want_grad=True)
        return dx, None


class _MeanFn(torch.autograd.Function):
    """loss.mean() of the runners (run_video_retrieval.py:422) as a library kernel (a captured step then holds no framework-side
    reduction / fill kernels); the upstream gradient stays on the device (a scalar tensor)."""
    @staticmethod
    def forward(ctx, x):
        ctx.n, ctx.shape = x.numel(), x.shape
        return ops.mean_fwd(x)

    @staticmethod
    def backward(ctx, dmean):
        return ops.mean_bwd(dmean.float().contiguous(), ctx.n, dmean).view(ctx.shape)


def mean_loss(per_example: torch.Tensor) -> torch.Tensor:
    """scalar mean of the per-example losses (cb_mean_fwd / cb_mean_bwd)"""
    x = per_example.float().contiguous()
    if x.numel() == 0:
        return x.sum() * float("nan")                   # mean of nothing, as torch defines it
    return _MeanFn.apply(x)


def _stack(logits) -> torch.Tensor:
    """list of per-clip logits, or their (n_clips, B, C) stack (a folded forward returns it directly), as fp32"""
    if torch.is_tensor(logits):
        return logits.float().contiguous()
    return torch.stack([t.float() for t in logits]).contiguous()          # (n_clips, B, C) fp32: layout only


def aggregate_clip_logits(logits, pool_method: str) -> torch.Tensor:
    """run_video_retrieval.py:402-411: "mean" / "max" over the clips -> (B, C); "lse" -> (B, n_clips, C), pooled inside
    lse_train_loss / lse_inference_logits."""
    st = _stack(logits)
    if pool_method in _MODES:
        return _AggFn.apply(st, _MODES[pool_method])
    if pool_method == "lse":
        return st.permute(1, 0, 2).contiguous()
    raise ValueError(f"Invalid value for pool_method, got {pool_method}, expect one of [`mean`, `max`, `lse`]")


def lse_train_loss(logits_bnc: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """run_video_retrieval.py:415-418 on the (B, n_clips, C) tensor of aggregate_clip_logits(..., "lse"): per-pair loss
    (B, 1) = logsumexp over all (clip, class) - logsumexp over clips of the labelled class."""
    st = logits_bnc.permute(1, 0, 2).contiguous().float()
    return _LseLossFn.apply(st, labels.view(-1)).view(-1, 1)


def lse_stack_train_loss(logits, labels: torch.Tensor) -> torch.Tensor:
    """aggregate_clip_logits(logits, "lse") followed by lse_train_loss, without materialising the (B, n_clips, C) transpose and
    its inverse (two copies forward, two backward): the kernel reads the (n_clips, B, C) stack as the forwards produced it."""
    return _LseLossFn.apply(_stack(logits), labels.view(-1)).view(-1, 1)


def lse_inference_logits(logits_bnc: torch.Tensor) -> torch.Tensor:
    """run_video_retrieval.py:674-676: logsumexp over the clips -> (B, C)."""
    st = logits_bnc.permute(1, 0, 2).contiguous().float()
    with torch.no_grad():
        out, _ = ops.clip_aggregate_fwd(st, ops.AGG_LSE)
    return out


def retrieval_scores(logits_bc: torch.Tensor) -> List[float]:
    """run_video_retrieval.py:681-690: softmax[:, 1] (2-way) or sigmoid (1 logit), rounded to 4 places."""
    probs = ops.retrieval_scores(logits_bc.float().contiguous()).tolist()          # softmax[:, 1] (two logits) or sigmoid (one): cb_retrieval_scores
    return [round(s, 4) for s in probs]
