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
_SKIP_KEYS = ("visual_inputs", "caption_ids", "vid_id", "question_ids", "n_examples_list")


def _pair_counts(cfg, n_examples_list):
    """forward_step of run_video_qa.py:206-210: the multiple-choice tasks present every (question, option) pair as its own
    text row, so each video's count is multiplied by cfg.num_labels before the model call."""
    counts = list(n_examples_list)
    if cfg is not None and _get(cfg, "task") in ("action", "transition"):
        nl = int(_get(cfg, "num_labels"))
        counts = [e * nl for e in counts]
    return counts


def forward_clips_stack(model, batch: Dict, num_clips: int, num_frm: int, fold: bool = True, cfg=None,
                        with_labels: bool = False) -> torch.Tensor:
    """The clip loop of the reference's training / validation steps (run_video_retrieval.py:391-401, run_video_qa.py:
    470-482): (B, num_clips*num_frm, 3, H, W) frames -> the (num_clips, B', C) stack of per-clip logits.

    fold=True (default) runs ALL clips as one forward: the frame tensor is viewed as (B*num_clips, num_frm, ...) -- no
    copy -- for one CNN batch, the text rows are repeated clip-major for one encoder batch of num_clips*B' pairs
    (ClipBert.forward_from_grid(clip_fold=...)).  The reference pools at logit level, so this is results-identical to its
    loop while every GEMM sees num_clips x the rows.  fold=False keeps the loop (one forward per clip)."""
    vis = batch["visual_inputs"]
    bsz = vis.shape[0]
    counts = _pair_counts(cfg, batch["n_examples_list"])
    labels = batch.get("labels") if with_labels else None
    extra = {k: v for k, v in batch.items() if k not in _SKIP_KEYS + ("labels", "text_input_ids", "text_input_mask")}
    ids, mask = batch["text_input_ids"], batch["text_input_mask"]
    if fold and num_clips > 1:
        grid = model.grid_features(vis.view(bsz * num_clips, num_frm, *vis.shape[2:]))
        # (the text rows are NOT repeated: forward_from_grid(clip_fold=n) lets every clip read the one caption batch in place)
        mini = dict(extra, visual_inputs=grid, text_input_ids=ids, text_input_mask=mask, labels=None, n_examples_list=counts)
        lg = model.forward_from_grid(mini, clip_fold=num_clips)["logits"]
        return lg.reshape(num_clips, lg.shape[0] // num_clips, *lg.shape[1:])
    vis = vis.view(bsz, num_clips, num_frm, *vis.shape[2:])
    logits = []
    for c in range(num_clips):
        mini = dict(extra, visual_inputs=vis[:, c].contiguous() if num_clips > 1 else vis[:, 0], text_input_ids=ids,
                    text_input_mask=mask, labels=labels, n_examples_list=list(counts))
        logits.append(model(mini)["logits"])
    return torch.stack(logits)


def forward_clips(model, batch: Dict, num_clips: int, num_frm: int, fold: bool = True, cfg=None) -> List[torch.Tensor]:
    """forward_clips_stack as the list of per-clip logits the reference's loop builds."""
    return list(forward_clips_stack(model, batch, num_clips, num_frm, fold=fold, cfg=cfg).unbind(0))


def training_loss(model, logits, labels, n_examples_list, pool_method: str) -> torch.Tensor:
    """:402-419: pool the clips (``logits``: list of per-clip logits or their (n_clips, B', C) stack) and take the mean
    per-pair loss."""
    if pool_method == "lse":
        return clips.mean_loss(clips.lse_stack_train_loss(logits, labels))
    pooled = clips.aggregate_clip_logits(logits, pool_method)
    if getattr(model, "retrieval", False):
        _, loss = model.transformer.calc_loss(pooled, labels, sample_size=len(n_examples_list))
    else:                                                   # QA heads (run_video_qa.py:417-419)
        _, loss = model.transformer.calc_loss(pooled, labels)
    return clips.mean_loss(loss)


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


def train_step(model, optimizer, batch: Dict, cfg, global_step: int, sync=None, n_epoch: int = 0, micro_step: int = 0,
               fold_clips: bool = True) -> torch.Tensor:
    """One micro-step of start_training (:380-494): forward over the clips, pooled loss, backward; on the last micro-step of
    a gradient-accumulation group ((micro_step + 1) % cfg.gradient_accumulation_steps == 0, :426-436) also the gradient
    all-reduce (``sync`` = clipbert_amd.dist.GradSync or None), the LR schedule and clip + AdamW.  Gradients of the
    micro-steps add up un-scaled, as in the reference; the exchange happens once per group (the sum is linear).

    With ``rt.after_encoder_backward = sync.reduce_transformer`` the transformer buckets are issued from inside the LAST encoder
    backward of the group (the model counts its pending encoder nodes, so an un-folded clip loop does not fire early) and travel
    during the ResNet backward; with ``rt.after_res5_backward = sync.reduce_cnn_early`` (after ``sync.set_cnn_split``) the
    grid_encoder + res5 part of the CNN range follows from inside the last ResNet backward, while res4 / res3 still run."""
    acc = max(1, int(_get(cfg, "gradient_accumulation_steps", 1) or 1))
    first, last = micro_step % acc == 0, (micro_step + 1) % acc == 0
    if first:
        optimizer.zero_grad(lazy=True)                      # a backward always follows: the encoder weight gradients are overwritten
    rt = model.rt
    rt.pending_encoder_nodes = 0
    rt.pending_cnn_nodes = 0
    hook, hook5 = rt.after_encoder_backward, rt.after_res5_backward
    if not last:
        rt.after_encoder_backward = None                    # no exchange before the group is complete
        rt.after_res5_backward = None
    try:
        stack = forward_clips_stack(model, batch, _get(cfg, "train_n_clips", 1), _get(cfg, "num_frm"), fold=fold_clips, cfg=cfg)
        loss = training_loss(model, stack, batch["labels"], batch["n_examples_list"], _get(cfg, "score_agg_func", "mean"))
        loss.backward()
    finally:
        rt.after_encoder_backward, rt.after_res5_backward = hook, hook5
    if not last:
        return loss.detach()
    scale, g16 = 1.0, None
    if sync is not None:
        if hook is None:
            sync.reduce_transformer()
        sync.reduce_cnn()
        g16 = sync.wire_gradients()                         # bf16 wire: the optimizer reads the reduced image directly
        sync.wait(cast_back=g16 is None)
        scale = sync.grad_scale
    set_learning_rates(optimizer, cfg, global_step + 1, n_epoch)
    if sync is not None and getattr(sync, "shard", False) and sync.active:
        # owner-only update: this rank holds the reduced gradients of 1/world of every bucket (reduce-scatter), updates those pieces,
        # and the all-gather hands every rank the new compute weights
        optimizer.step(grad_scale=scale, grad16=g16, pieces=sync.owned_pieces(), norm_reduce=sync.norm_all_reduce)
        sync.gather_updated()
    else:
        optimizer.step(grad_scale=scale, grad16=g16)
    return loss.detach()


def _all_gather_scalar(x, group=None) -> list:
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [x]
    parts = [None] * dist.get_world_size(group)
    dist.all_gather_object(parts, x, group=group)
    return parts


def _all_sum(x: float, group=None) -> float:
    """sum of a host scalar over the ranks (the reference's sum(all_gather_list(x)), :254-256)"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return x
    parts = [None] * dist.get_world_size(group)
    dist.all_gather_object(parts, float(x), group=group)
    return float(sum(parts))


@torch.no_grad()
def validate_retrieval(model, val_loader, eval_videos, cfg, gt_txt_id2vid_id=None, group=None) -> Dict[str, float]:
    """validate of run_video_retrieval.py:224-276: mean ITM loss / accuracy over ``val_loader`` (single-clip batches as the
    training set yields them) and the retrieval metrics of inference_retrieval over ``eval_videos``; sums over the ranks."""
    was_training = model.training
    model.eval()
    loss, n_ex, n_correct = 0.0, 0, 0
    for batch in val_loader:
        batch = {k: v for k, v in batch.items() if k not in ("caption_ids", "vid_id")}
        targets = batch["labels"]
        out = model(dict(batch))
        if torch.is_tensor(out["loss"]):
            loss += float(out["loss"].sum().item())
        n_ex += len(targets)
        logits = out["logits"].float()
        if logits.shape[1] == 2:
            n_correct += int((logits.max(dim=-1)[1] == targets).sum().item())
        else:                                                  # rank loss: first score of each group is the positive (:244-250)
            pred = (logits > 0).long().view(out["loss"].shape[0], -1)          # sigmoid(x) > 0.5  <=>  x > 0
            n_correct += int((pred[:, 0] == targets.view(out["loss"].shape[0], -1)[:, 0]).sum().item())
    loss, n_ex, n_correct = _all_sum(loss, group), _all_sum(n_ex, group), _all_sum(n_correct, group)
    _rows, metrics = inference_retrieval(model, eval_videos, cfg, gt_txt_id2vid_id, group=group)
    model.train(was_training)
    log = {"valid/loss": loss / max(n_ex, 1), "valid/acc": n_correct / max(n_ex, 1)}
    for kind, m in (metrics or {}).items():
        log.update({f"valid/{kind}_{k}": round(v, 4) for k, v in m.items()})
    return log


def start_training(model, optimizer, train_loader, cfg, sync=None, validate_fn=None, model_saver=None, restorer=None,
                   total_n_examples: Optional[int] = None, fold_clips: bool = True, log_fn=None, overlap: bool = True) -> int:
    """The loop of start_training (run_video_retrieval.py:379-516 / run_video_qa.py:457-560) around train_step: infinite
    iteration over ``train_loader`` until cfg.num_train_steps optimizer steps, gradient accumulation, LR schedules with the
    multi-step epoch counter, validation + ``model_step_N.pt`` every cfg.valid_steps (and once at the end), restorer.step()
    after every optimizer step.  ``train_loader`` yields collated batches (clipbert_amd.data.PrefetchLoader delivers them with
    uint8 frames already in HBM).  With several ranks and ``overlap`` the gradient exchange is armed to leave from inside the
    backward (GradSync.attach).  Returns the final global step.

    Ranks: pass ``model_saver`` / ``restorer`` on EVERY rank (clipbert_amd.checkpoint: both write on rank 0 only, the restorer
    restores on all ranks -- the reference's order, run_video_retrieval.py:329-346).  All ranks must run the same number of
    optimizer steps: the resumed global step is agreed on across the ranks (max), and when the ranks did not all restore the
    same step, the most advanced rank's parameters, AdamW moments and optimizer step count are broadcast before the first step."""
    from .data import InfiniteIterator
    if sync is not None and sync.world > 1 and overlap and model.rt is not None and model.rt.after_encoder_backward is None:
        sync.attach(model)
    acc = max(1, int(_get(cfg, "gradient_accumulation_steps", 1) or 1))
    global_step = restorer.global_step if restorer is not None else 0
    if sync is not None and sync.world > 1 and not sync.dry:
        steps = _all_gather_scalar(global_step)
        if len(set(steps)) > 1:
            # the ranks resumed from different states (e.g. restore.pt readable on one rank only): ALL of them continue from the most
            # advanced one -- its global step, parameters, AdamW moments and optimizer step count (the reference broadcasts both
            # model and optimizer state from rank 0 after its restore, run_video_retrieval.py:304-305,329-346)
            global_step = int(max(steps))
            src = steps.index(global_step)
            if restorer is not None:
                restorer.global_step = global_step
            sync.broadcast_state(optimizer, src)
    num_train_steps, valid_steps = int(_get(cfg, "num_train_steps")), int(_get(cfg, "valid_steps", 0) or 0)
    n_gpu = sync.world if sync is not None else 1
    total_bsz = n_gpu * int(_get(cfg, "train_batch_size", 1)) * acc * int(_get(cfg, "max_n_example_per_group", 1) or 1)
    model.train()

    sharded = sync is not None and getattr(sync, "shard", False) and sync.active

    def run_validation(step):
        if validate_fn is not None:
            log = validate_fn(model, step)
            if log_fn is not None and log is not None:
                log_fn(step, log)
        if model_saver is not None:
            if sharded:
                sync.gather_state(optimizer)                # (every rank: the fp32 masters and moments return from their owners)
            model_saver.save(step=step, model=model)

    if global_step >= num_train_steps:
        return global_step
    for micro, batch in enumerate(InfiniteIterator(train_loader)):
        n_epoch = int(1.0 * total_bsz * (global_step + 1) / total_n_examples) if total_n_examples else 0
        loss = train_step(model, optimizer, batch, cfg, global_step, sync=sync, n_epoch=n_epoch, micro_step=micro, fold_clips=fold_clips)
        if (micro + 1) % acc != 0:
            continue
        global_step += 1
        if log_fn is not None:
            log_fn(global_step, {"train/loss": float(loss.item()) if global_step % 50 == 0 else None})
        if restorer is not None:
            if sharded and (restorer.global_step + 1) % restorer.save_steps == 0:
                sync.gather_state(optimizer)                # restore.pt is written by this step()
            restorer.step()
        if valid_steps and global_step % valid_steps == 0:
            run_validation(global_step)
        if global_step >= num_train_steps:
            break
    if not valid_steps or global_step % valid_steps != 0:
        run_validation(global_step)
    return global_step


# ---- retrieval inference (:628-734) ------------------------------------------------------------------------------------
@torch.no_grad()
def inference_retrieval_video(model, visual_inputs: torch.Tensor, text_input_ids: torch.Tensor, text_input_mask: torch.Tensor,
                              cfg, cache_cnn: bool = True, max_pairs_per_pass: int = 256) -> List[float]:
    """Scores of ONE video (1, inference_n_clips*num_frm, 3, H, W) against all its candidate captions (:640-690).

    cache_cnn=True (row N1): the grid features of all clips are computed once, in one CNN batch, and every text
    mini-batch runs only the cross-modal encoder -- on several clips at once (pairs = clips x captions, at most
    ``max_pairs_per_pass`` per encoder pass: measured on MI355X, an encoder batch whose activations outgrow the 256 MB
    Infinity Cache (1024 pairs = 42 k token rows: 258 MB per FFN activation) runs its GEMMs 3-5x slower per row than a
    batch of a few thousand rows).  cache_cnn=False is the reference's order of evaluation (full forward per clip per
    mini-batch).  Both give the same scores."""
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
            cpp = max(1, min(n_clips, max_pairs_per_pass // max(nb, 1)))          # clips per encoder pass
            per_clip = []
            for c0 in range(0, n_clips, cpp):
                nc = min(cpp, n_clips - c0)
                out = model.forward_from_grid(dict(visual_inputs=grid[c0:c0 + nc], text_input_ids=ids.repeat(nc, 1),
                                                   text_input_mask=mask.repeat(nc, 1), labels=None,
                                                   n_examples_list=[nb] * nc))
                per_clip.extend(out["logits"].view(nc, nb, -1).unbind(0))
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


def shard_for_rank(n_items: int, rank: int, world: int) -> range:
    """Items (videos) of this rank: a strided partition, what DistributedSampler(shuffle=False) hands out without padding
    (no duplicated videos: the reference's eval_retrieval drops duplicates anyway, :574-577)."""
    return range(rank, n_items, world)


def gather_retrieval_rows(rows: List[Dict], group=None) -> List[Dict]:
    """All ranks' dict(vid_id, txt_id, score) rows on every rank, in rank order -- the exchange the reference does through
    per-rank JSON files on shared storage (:696-724); here ONE all-gather of a compact (ids, float32 scores) payload.
    Works on RCCL ("nccl") and gloo groups; a single process returns its rows unchanged."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return rows
    payload = ([r["vid_id"] for r in rows], [r["txt_id"] for r in rows], np.asarray([r["score"] for r in rows], dtype=np.float64))
    parts = [None] * dist.get_world_size(group)
    dist.all_gather_object(parts, payload, group=group)
    out: List[Dict] = []
    for vids, txts, sc in parts:
        out.extend(dict(vid_id=v, txt_id=t, score=float(x)) for v, t, x in zip(vids, txts, sc))
    return out


@torch.no_grad()
def inference_retrieval(model, videos, cfg, gt_txt_id2vid_id: Optional[Dict] = None, group=None, cache_cnn: bool = True):
    """inference_retrieval of the reference (:628-734) for the videos of THIS rank: ``videos`` yields dict(vid_id,
    visual_inputs (1, n_clips*num_frm, 3, H, W), text_input_ids, text_input_mask, caption_ids) -- one video against all its
    candidate captions.  Returns (rows of all ranks, metrics or None); videos are independent units, the only exchange is
    the final gather of the score rows."""
    was_training = model.training
    model.eval()
    rows: List[Dict] = []
    for b in videos:
        scores = inference_retrieval_video(model, b["visual_inputs"], b["text_input_ids"], b["text_input_mask"], cfg, cache_cnn=cache_cnn)
        rows.extend(dict(vid_id=b["vid_id"], txt_id=c, score=s) for c, s in zip(b["caption_ids"], scores))
    rows = gather_retrieval_rows(rows, group)
    metrics = eval_retrieval(rows, gt_txt_id2vid_id) if gt_txt_id2vid_id is not None else None
    model.train(was_training)
    return rows, metrics


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
        counts = _pair_counts(cfg, batch["n_examples_list"])
        out = model(dict(visual_inputs=vis.contiguous(), text_input_ids=batch["text_input_ids"].repeat(n_clips, 1),
                         text_input_mask=batch["text_input_mask"].repeat(n_clips, 1), labels=None, n_examples_list=counts * n_clips))
        lg = out["logits"]
        logits = list(lg.view(n_clips, lg.shape[0] // n_clips, *lg.shape[1:]).unbind(0))
    else:
        logits = forward_clips(model, dict(batch, labels=None), n_clips, num_frm, fold=False, cfg=cfg)
    pool = _get(cfg, "score_agg_func", "mean")
    pooled = clips.aggregate_clip_logits(logits, pool)
    if pool == "lse":
        pooled = clips.lse_inference_logits(pooled)
    if _get(cfg, "task", "action") in ("action", "transition", "frameqa", "msrvtt_qa"):
        return pooled.max(dim=-1)[1].tolist()
    return (pooled + 0.5).long().clamp(min=1, max=10).reshape(-1).tolist()


@torch.no_grad()
def validate_qa(model, val_loader, cfg, group=None, fold_clips: bool = True):
    """validate of run_video_qa.py:216-362 without the dataset-specific score tables: per batch the pooled prediction of
    qa_predict over cfg.inference_n_clips clips -> rows dict(question_id, answer); with labels in the batch also the overall
    accuracy (classification tasks) / mean absolute error rounded as the reference reports it (count), summed over ranks."""
    was_training = model.training
    model.eval()
    rows, n_ex, n_hit, abs_err = [], 0, 0, 0.0
    for batch in val_loader:
        qids = batch.get("question_ids", list(range(n_ex, n_ex + len(batch["n_examples_list"]))))
        b = {k: v for k, v in batch.items() if k != "question_ids"}
        pred = qa_predict(model, b, cfg, fold_clips=fold_clips)
        rows.extend(dict(question_id=q, answer=a) for q, a in zip(qids, pred))
        if batch.get("labels") is not None:
            gt = batch["labels"].view(-1).tolist()
            n_ex += len(gt)
            n_hit += sum(int(a == int(g)) for a, g in zip(pred, gt))
            abs_err += sum(abs(a - g) for a, g in zip(pred, gt))
    n_ex, n_hit, abs_err = _all_sum(n_ex, group), _all_sum(n_hit, group), _all_sum(abs_err, group)
    model.train(was_training)
    log = {}
    if n_ex:
        log = {"valid/overall_acc": round(100.0 * n_hit / n_ex, 2)} if _get(cfg, "task", "action") != "count" else {"valid/mae": round(abs_err / n_ex, 2)}
    return rows, log


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
