"""Task-loop pieces around the hot path: the training step and the retrieval inference of
src/tasks/run_video_retrieval.py, and its metrics, on top of clipbert_amd.modeling.  Host logic only -- every tensor op is
a libclipbert_hip kernel reached through the model / clips / optimizer objects.

``cfg`` is any object with the reference's config attributes (src/configs/*.json): num_frm, train_n_clips,
inference_n_clips, score_agg_func, inference_batch_size, learning_rate, decay, cnn_learning_rate, cnn_lr_decay,
num_train_steps, warmup_ratio, step_decay_epochs, cnn_step_decay_epochs, transformer_lr_mul, cnn_lr_mul."""
import math
from collections import defaultdict
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import clips
from .optim import get_lr_sched


def _get(cfg, name, default=None):
    return cfg.get(name, default) if isinstance(cfg, dict) else getattr(cfg, name, default)


# ---- training (run_video_retrieval.py:380-494) ------------------------------------------------------------------------
def forward_clips(model, batch: Dict, num_clips: int, num_frm: int) -> List[torch.Tensor]:
    """The clip loop :391-401: (B, num_clips*num_frm, 3, H, W) frames -> list of per-clip logits."""
    vis = batch["visual_inputs"]
    bsz = vis.shape[0]
    vis = vis.view(bsz, num_clips, num_frm, *vis.shape[2:])
    logits = []
    for c in range(num_clips):
        mini = {k: v for k, v in batch.items() if k not in ("visual_inputs", "caption_ids", "vid_id")}
        mini["visual_inputs"] = vis[:, c].contiguous() if num_clips > 1 else vis[:, 0]
        mini["n_examples_list"] = list(batch["n_examples_list"])
        logits.append(model(mini)["logits"])
    return logits


def training_loss(model, logits: List[torch.Tensor], labels, n_examples_list, pool_method: str) -> torch.Tensor:
    """:402-419: pool the clips and take the mean per-pair loss."""
    pooled = clips.aggregate_clip_logits(logits, pool_method)
    if pool_method == "lse":
        loss = clips.lse_train_loss(pooled, labels)
    elif getattr(model, "retrieval", False):
        _, loss = model.transformer.calc_loss(pooled, labels, sample_size=len(n_examples_list))
    else:                                                   # QA heads (run_video_qa.py:417-419)
        _, loss = model.transformer.calc_loss(pooled, labels)
    return loss.mean()


def set_learning_rates(optimizer, cfg, global_step: int, n_epoch: int = 0):
    """:438-467: transformer / CNN schedules onto the 8 parameter groups (0,1 new transformer; 2,3 transformer; 4,5 new
    CNN; 6,7 CNN)."""
    lr_t = get_lr_sched(global_step, _get(cfg, "decay", "linear"), _get(cfg, "learning_rate"), _get(cfg, "num_train_steps"),
                        warmup_ratio=_get(cfg, "warmup_ratio", 0.1), decay_epochs=_get(cfg, "step_decay_epochs", ()),
                        multi_step_epoch=n_epoch)
    lr_c = get_lr_sched(global_step, _get(cfg, "cnn_lr_decay", "linear"), _get(cfg, "cnn_learning_rate"), _get(cfg, "num_train_steps"),
                        warmup_ratio=_get(cfg, "warmup_ratio", 0.1), decay_epochs=_get(cfg, "cnn_step_decay_epochs", ()),
                        multi_step_epoch=n_epoch)
    assert len(optimizer.param_groups) == 8
    for i, pg in enumerate(optimizer.param_groups):
        if i in (0, 1):
            pg["lr"] = _get(cfg, "transformer_lr_mul", 1.0) * lr_t
        elif i in (2, 3):
            pg["lr"] = lr_t
        elif i in (4, 5):
            pg["lr"] = _get(cfg, "cnn_lr_mul", 1.0) * lr_c
        else:
            pg["lr"] = lr_c
    return lr_t, lr_c


def train_step(model, optimizer, batch: Dict, cfg, global_step: int, sync=None, n_epoch: int = 0, micro_step: int = 0) -> torch.Tensor:
    """One micro-step of start_training (:380-494): forward over the clips, pooled loss, backward; on the last micro-step of
    a gradient-accumulation group ((micro_step + 1) % cfg.gradient_accumulation_steps == 0, :426-436) also the gradient
    all-reduce (``sync`` = clipbert_amd.dist.GradSync or None), the LR schedule and clip + AdamW.  Gradients of the
    micro-steps add up un-scaled, as in the reference; the exchange happens once per group (the sum is linear)."""
    acc = max(1, int(_get(cfg, "gradient_accumulation_steps", 1) or 1))
    first, last = micro_step % acc == 0, (micro_step + 1) % acc == 0
    if first:
        optimizer.zero_grad()
    hook = model.rt.after_encoder_backward
    if not last:
        model.rt.after_encoder_backward = None              # no exchange before the group is complete
    try:
        logits = forward_clips(model, batch, _get(cfg, "train_n_clips", 1), _get(cfg, "num_frm"))
        loss = training_loss(model, logits, batch["labels"], batch["n_examples_list"], _get(cfg, "score_agg_func", "mean"))
        loss.backward()
    finally:
        model.rt.after_encoder_backward = hook
    model.rt.seed_dev.add_(1)                               # fresh dropout masks for the next forward
    if not last:
        return loss.detach()
    scale = 1.0
    if sync is not None:
        if hook is None:
            sync.reduce_transformer()
        sync.reduce_cnn()
        sync.wait()
        scale = sync.grad_scale
    set_learning_rates(optimizer, cfg, global_step + 1, n_epoch)
    optimizer.step(grad_scale=scale)
    return loss.detach()


# ---- retrieval inference (:628-734) ------------------------------------------------------------------------------------
@torch.no_grad()
def inference_retrieval_video(model, visual_inputs: torch.Tensor, text_input_ids: torch.Tensor, text_input_mask: torch.Tensor,
                              cfg, cache_cnn: bool = True) -> List[float]:
    """Scores of ONE video (1, inference_n_clips*num_frm, 3, H, W) against all its candidate captions (:640-690).

    cache_cnn=True (row N1): the grid features of all clips are computed once, in one CNN batch, and every text
    mini-batch runs only the cross-modal encoder -- on all clips at once (pairs = clips x captions).  cache_cnn=False is
    the reference's order of evaluation (full forward per clip per mini-batch).  Both give the same scores."""
    n_clips, num_frm = _get(cfg, "inference_n_clips", 1), _get(cfg, "num_frm")
    pool, eval_bsz = _get(cfg, "score_agg_func", "mean"), _get(cfg, "inference_batch_size", 64)
    vis = visual_inputs.view(n_clips, num_frm, *visual_inputs.shape[2:])
    n_txt = text_input_ids.shape[0]
    grid = model.grid_features(vis) if cache_cnn else None          # (n_clips, T, H', W', hidden)
    scores: List[float] = []
    for i0 in range(0, n_txt, eval_bsz):
        ids, mask = text_input_ids[i0:i0 + eval_bsz], text_input_mask[i0:i0 + eval_bsz]
        nb = ids.shape[0]
        if cache_cnn:
            out = model.forward_from_grid(dict(visual_inputs=grid, text_input_ids=ids.repeat(n_clips, 1),
                                               text_input_mask=mask.repeat(n_clips, 1), labels=None,
                                               n_examples_list=[nb] * n_clips))
            per_clip = list(out["logits"].view(n_clips, nb, -1).unbind(0))
        else:
            per_clip = []
            for c in range(n_clips):
                out = model(dict(visual_inputs=vis[c:c + 1], text_input_ids=ids, text_input_mask=mask, labels=None,
                                 n_examples_list=[nb]))
                per_clip.append(out["logits"])
        pooled = clips.aggregate_clip_logits(per_clip, pool)
        if pool == "lse":
            pooled = clips.lse_inference_logits(pooled)
        scores.extend(clips.retrieval_scores(pooled))
    return scores


# ---- video QA inference (run_video_qa.py:216-300) ------------------------------------------------------------------------
@torch.no_grad()
def qa_predict(model, batch: Dict, cfg, fold_clips: bool = False) -> List[int]:
    """Predicted answer ids of one batch: clip loop over inference_n_clips, pooling, then argmax (classification tasks
    action / transition / frameqa / msrvtt_qa) or round-and-clamp to 1..10 (the regression task "count").

    fold_clips=True (row N1 for QA): all clips of all videos go through the CNN and the encoder as ONE batch -- (clip, video)
    pairs become the "videos" of a single forward -- instead of inference_n_clips separate forwards; same predictions."""
    n_clips, num_frm = _get(cfg, "inference_n_clips", 1), _get(cfg, "num_frm")
    if fold_clips and n_clips > 1:
        vis = batch["visual_inputs"]
        bsz = vis.shape[0]
        vis = vis.view(bsz, n_clips, num_frm, *vis.shape[2:]).transpose(0, 1).reshape(n_clips * bsz, num_frm, *vis.shape[2:])
        counts = list(batch["n_examples_list"])
        out = model(dict(visual_inputs=vis.contiguous(), text_input_ids=batch["text_input_ids"].repeat(n_clips, 1),
                         text_input_mask=batch["text_input_mask"].repeat(n_clips, 1), labels=None, n_examples_list=counts * n_clips))
        lg = out["logits"]
        logits = list(lg.view(n_clips, lg.shape[0] // n_clips, *lg.shape[1:]).unbind(0))
    else:
        logits = forward_clips(model, dict(batch, labels=None), n_clips, num_frm)
    pool = _get(cfg, "score_agg_func", "mean")
    pooled = clips.aggregate_clip_logits(logits, pool)
    if pool == "lse":
        pooled = clips.lse_inference_logits(pooled)
    if _get(cfg, "task", "action") in ("action", "transition", "frameqa", "msrvtt_qa"):
        return pooled.max(dim=-1)[1].tolist()
    return (pooled + 0.5).long().clamp(min=1, max=10).reshape(-1).tolist()


# ---- metrics (:519-625) ------------------------------------------------------------------------------------------------
def retrieval_metrics_from_scores(score_matrix, gt_cols: Sequence[int]) -> Dict[str, float]:
    """score_matrix (#queries, #candidates), gt_cols[i] = index of query i's ground-truth candidate -> recall@{1,5,10} in %,
    median and mean rank (1-indexed); ties keep the order of a stable descending sort like torch.sort."""
    sm = torch.as_tensor(score_matrix, dtype=torch.float32)
    order = torch.sort(sm, dim=1, descending=True)[1]
    gt = torch.as_tensor(list(gt_cols)).view(-1, 1)
    ranks = (order == gt).float().argmax(dim=1).numpy() + 1
    n = float(len(ranks))
    return dict(r1=100.0 * float((ranks <= 1).sum()) / n, r5=100.0 * float((ranks <= 5).sum()) / n,
                r10=100.0 * float((ranks <= 10).sum()) / n, medianR=float(np.median(ranks)), meanR=float(np.mean(ranks)))


def eval_retrieval(vid_txt_score_dicts: List[Dict], gt_txt_id2vid_id: Dict) -> Dict[str, Dict[str, float]]:
    """Same contract as the reference's eval_retrieval (:563-625): rows of dict(vid_id, txt_id, score) -> text2video and
    video2text metric dicts (first occurrence of a (txt, vid) pair wins; every caption must see the same videos)."""
    by_txt = defaultdict(dict)
    for d in vid_txt_score_dicts:
        by_txt[d["txt_id"]].setdefault(d["vid_id"], d["score"])
    txt_ids = list(by_txt)
    vid_ids = list(by_txt[txt_ids[0]])
    for t in txt_ids:
        assert len(by_txt[t]) == len(vid_ids), "each captions should be compared with the same #videos."
    vid_idx = {v: i for i, v in enumerate(vid_ids)}
    sm = torch.zeros(len(txt_ids), len(vid_ids))
    for r, t in enumerate(txt_ids):
        for v, s in by_txt[t].items():
            sm[r, vid_idx[v]] = s
    t2v = retrieval_metrics_from_scores(sm, [vid_idx[gt_txt_id2vid_id[t]] for t in txt_ids])
    txt_idx = {t: i for i, t in enumerate(txt_ids)}
    gt_v2t = {v: t for t, v in gt_txt_id2vid_id.items()}
    v2t = retrieval_metrics_from_scores(sm.t(), [txt_idx[gt_v2t[v]] for v in vid_ids])
    return dict(text2video=t2v, video2text=v2t)


def qa_accuracy(pred_ids: Sequence[int], gt_ids: Sequence[int]) -> float:
    """answer id = argmax of the pooled logits (run_video_qa.py:273-275); accuracy in %."""
    p, g = np.asarray(list(pred_ids)), np.asarray(list(gt_ids))
    return 100.0 * float((p == g).mean()) if len(g) else math.nan
