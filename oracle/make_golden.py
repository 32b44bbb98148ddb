"""Generate tests/golden/*.npz from the reference itself.  TEST INFRASTRUCTURE ONLY.

Run HERE (needs /root/reference):   python -m oracle.make_golden

For every case the transformer half is evaluated by the REFERENCE's own
``src/modeling/modeling.py`` classes (imported verbatim through oracle/ref_shim.py) in fp32 /
eval mode; the CNN half (detectron2, absent) by oracle.clipbert_oracle.grid_feat_backbone, whose
output is handed to the reference classes exactly as ``ClipBert.forward`` does
(src/modeling/e2e_model.py:29-39).  Weights / inputs are the deterministic synthetic ones of
clipbert_amd/synthetic.py (seeded per key), so the GPU box can rebuild them bit-identically
without the reference.  Stored: final outputs, and for each intermediate stage a fingerprint
(mean, mean |x|, and 32 values at fixed flat indices).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from clipbert_amd import synthetic as S          # noqa: E402
from oracle import clipbert_oracle as O          # noqa: E402
from oracle import ref_shim                      # noqa: E402

GOLDEN_DIR = os.path.join(os.path.dirname(HERE), "tests", "golden")

# name -> case description.  Shapes follow BASELINE.json configs / SURVEY.md 8(d), scaled in
# batch only (the per-example arithmetic is batch-independent).
CASES = {
    # BASELINE config 1: single 224x224 image + 20-token caption, ITM+MLM forward, batch 2
    "pretrain_cfg1": dict(head="pretraining", n_videos=2, n_frames=1, size=224, lt=20, repeat=1,
                          cfg=dict()),
    # headline / config 2 shape: 2 frames 224, Lt=32, retrieval CE, pos+neg text per video
    "retrieval_ce": dict(head="retrieval", n_videos=2, n_frames=2, size=224, lt=32, repeat=2,
                         cfg=dict(num_labels=2, loss_type="ce", margin=0.1)),
    "retrieval_rank": dict(head="retrieval", n_videos=2, n_frames=2, size=224, lt=32, repeat=2,
                           cfg=dict(num_labels=1, loss_type="rank", margin=0.1)),
    # config 4: TGIF-QA action, multiple choice with 5 options, Lt=25 (4 questions; the classifier is TRAINED, see HEAD_TRAIN)
    "tgif_mc": dict(head="multiple_choice", n_videos=4, n_frames=2, size=224, lt=25, repeat=5,
                    cfg=dict(num_labels=5, loss_type="ce")),
    # frame-QA style open-ended classifier (a18); small label space to keep the fixture small
    "seqcls_ce": dict(head="sequence_classification", n_videos=2, n_frames=1, size=224, lt=16,
                      repeat=1, cfg=dict(num_labels=37, loss_type="ce")),
}


# Multi-clip cases: the reference's task LOOPS around the model (src/tasks/run_video_retrieval.py:387-422 training,
# :655-690 inference; src/tasks/run_video_qa.py:470-501): N_clip forwards, pooling of the logits, loss / scores.
CLIP_CASES = {
    # BASELINE configs[2]: MSRVTT retrieval at the JSON's native sizes (448 px, L_txt 20 -> L = 69), N_clip = 4, LSE pooling
    "msrvtt_lse_c4_448": dict(head="retrieval", n_videos=2, n_clips=4, n_frames=2, size=448, lt=20, repeat=2, pool="lse", mode="train",
                              cfg=dict(num_labels=2, loss_type="ce", margin=0.1)),
    # BASELINE configs[3]: TGIF-QA action at the JSON's native sizes (768 px, L_txt 25 -> L = 169), N_clip = 2, mean pooling
    "tgif_mc_c2_768": dict(head="multiple_choice", n_videos=4, n_clips=2, n_frames=2, size=768, lt=25, repeat=5, pool="mean", mode="train",
                           cfg=dict(num_labels=5, loss_type="ce")),
    # BASELINE configs[4]: retrieval inference, one video x 16 clips against 8 captions, LSE pooling, scores rounded to 4 places
    "msrvtt_infer_c16": dict(head="retrieval", n_videos=1, n_clips=16, n_frames=2, size=224, lt=32, repeat=8, pool="lse", mode="infer",
                             cfg=dict(num_labels=2, loss_type="ce", margin=0.1)),
}

# Heads with DECIDED margins.  A random-init model separates the five answer options of a question (and the
# 30522 entries of the MLM head) by less than bf16 storage resolves, so "argmax-exact answer ids" could not be asserted in the
# arithmetic that is timed.  For these cases the HEAD parameters listed here are trained for a few hundred steps of the REFERENCE's
# own AdamW (src/optimization/adamw.py:40-103) through the reference's own head modules on the case's fixed batch (upstream features
# frozen), and shipped as tests/golden/<case>_head.npz; build_case() overlays them on the synthetic state dict, so the GPU box rebuilds
# the exact weights without the reference.  The golden outputs are then produced by the reference classes with those weights, as for
# every other case.  Target: every top-1 / top-2 margin > 10x the bf16 error of the logits where the features allow it (oracle/make_bf16_yardstick.py records
# margin and error per case).
HEAD_TRAIN = {
    "tgif_mc": dict(params=("transformer.classifier.2.",), steps=400, lr=2e-3),
    # 768 px / L = 169: the five answer texts move the pooled features by rms 0.013 against 0.0044 of bf16 noise, so no head reaches
    # 10x here (measured: the fully trained classifier, either layer or both, with or without noise injection, ends at a margin of
    # 4-5x the bf16 error of the pooled logits); 2000 steps give 5.3x -- still decided: flipping an answer needs 2 x the error > margin
    "tgif_mc_c2_768": dict(params=("transformer.classifier.2.",), steps=2000, lr=5e-3),
    "pretrain_cfg1": dict(params=("transformer.cls.predictions.transform.",), steps=300, lr=1e-3),
}

_SD_CACHE = {}


def head_override_path(name: str) -> str:
    return os.path.join(GOLDEN_DIR, name + "_head.npz")


def mlm_train_targets(ids: torch.Tensor, seed: int = 42) -> torch.Tensor:
    """targets of the pretraining head's training (HEAD_TRAIN): one fixed random token per TEXT POSITION, padding included (the
    golden compares the MLM arg-max at every position)"""
    return torch.randint(1000, 30522, ids.shape, generator=S._gen(seed, "mlm_target"), dtype=torch.long)


def build_case(name: str, seed: int = 42, trained_head: bool = True):
    c = CASES[name] if name in CASES else dict(CLIP_CASES[name], n_frames=CLIP_CASES[name]["n_clips"] * CLIP_CASES[name]["n_frames"])
    cfg = dict(O.BASE_CONFIG)
    cfg.update(c["cfg"])
    head = c["head"]
    key = (head, seed, tuple(sorted((k, str(v)) for k, v in cfg.items())))
    if key not in _SD_CACHE:          # ~150 M random values: generate once per process (callers never mutate it)
        _SD_CACHE.clear()
        _SD_CACHE[key] = S.full_state_dict(cfg, head, seed)
    sd = _SD_CACHE[key]
    if trained_head and name in HEAD_TRAIN:
        over = np.load(head_override_path(name))      # missing file = the fixture was not generated: fail loudly
        sd = dict(sd)
        for k in over.files:
            assert k in sd and tuple(sd[k].shape) == over[k].shape, k
            sd[k] = torch.from_numpy(over[k].astype(np.float32))
        if "transformer.cls.predictions.bias" in sd:   # aliased keys of the tied decoder stay aliased
            sd["transformer.cls.predictions.decoder.bias"] = sd["transformer.cls.predictions.bias"]
    frames = S.synthetic_frames(c["n_videos"], c["n_frames"], c["size"], seed)
    n_pairs = c["n_videos"] * c["repeat"]
    ids, mask = S.synthetic_text(n_pairs, c["lt"], seed)
    batch = dict(visual_inputs=O.image_norm(frames, S.PIXEL_MEAN, S.PIXEL_STD),
                 text_input_ids=ids, text_input_mask=mask,
                 n_examples_list=[c["repeat"]] * c["n_videos"])
    if head == "pretraining":
        mlm = ids.clone()
        sel = torch.rand(ids.shape, generator=S._gen(seed, "mlm")) < 0.15
        mlm[~(sel & mask.bool())] = -100
        batch["mlm_labels"] = mlm
        batch["itm_labels"] = S.synthetic_labels(n_pairs, 2, seed)
    elif head == "multiple_choice":
        batch["labels"] = S.synthetic_labels(c["n_videos"], cfg["num_labels"], seed)
    elif head == "retrieval" and name in CLIP_CASES:
        batch["labels"] = torch.tensor(([1] + [0] * (c["repeat"] - 1)) * c["n_videos"])      # 1 positive + negatives per video
    elif head == "retrieval" and cfg["loss_type"] == "rank":
        batch["labels"] = torch.zeros(n_pairs, dtype=torch.long)  # unused by the rank loss
    else:
        batch["labels"] = S.synthetic_labels(n_pairs, cfg["num_labels"], seed)
    return cfg, head, sd, batch


def fingerprint(t: torch.Tensor, n: int = 32):
    f = t.detach().float().reshape(-1)
    idx = torch.linspace(0, f.numel() - 1, n).long()
    return np.concatenate([[f.mean().item(), f.abs().mean().item()], f[idx].numpy()]).astype(np.float32)


REF_CLASS = dict(retrieval="ClipBertForVideoTextRetrieval", multiple_choice="ClipBertForMultipleChoice",
                 sequence_classification="ClipBertForSequenceClassification",
                 pretraining="ClipBertForPreTraining")


def train_head(name: str, verbose: bool = True):
    """HEAD_TRAIN: train the listed head parameters with the reference's AdamW through the reference's head modules on the frozen
    upstream features of the case's batch; writes tests/golden/<name>_head.npz."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_ref_adamw", os.path.join(ref_shim.REFERENCE_ROOT, "src", "optimization", "adamw.py"))
    ref_adamw = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_adamw)
    ht = HEAD_TRAIN[name]
    cfg, head, sd, batch = build_case(name, trained_head=False)
    mo, _tr = ref_shim.load_reference_modeling()
    model = getattr(mo, REF_CLASS[head])(ref_shim.make_config(cfg)).eval()
    tsd = {k[len("transformer."):]: v for k, v in sd.items() if k.startswith("transformer.")}
    missing, unexpected = model.load_state_dict(tsd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    c = CASES[name] if name in CASES else CLIP_CASES[name]
    n_clips = c.get("n_clips", 1)
    vis = batch["visual_inputs"]
    vis = vis.view(c["n_videos"], n_clips, vis.shape[1] // n_clips, *vis.shape[2:])
    feats = []                                       # frozen upstream features, one entry per clip
    with torch.no_grad():
        for k in range(n_clips):
            grid = O.repeat_rows(O.grid_feat_backbone(sd, vis[:, k], "cnn."), batch["n_examples_list"])
            seq, pooled = model.bert(text_input_ids=batch["text_input_ids"], visual_inputs=grid,
                                     attention_mask=batch["text_input_mask"])[:2]
            feats.append((seq, pooled))
    named = {"transformer." + n: p for n, p in model.named_parameters()}
    train = {n: p for n, p in named.items() if n.startswith(ht["params"])}
    assert train, ht
    for p in model.parameters():
        p.requires_grad_(False)
    for p in train.values():
        p.requires_grad_(True)
    opt = ref_adamw.AdamW(list(train.values()), lr=ht["lr"], betas=(0.9, 0.98), weight_decay=0.0)
    lt = batch["text_input_mask"].shape[1]
    import warnings
    for step in range(ht["steps"]):
        if head == "pretraining":
            scores, _ = model.cls(feats[0][0][:, :lt], feats[0][1])
            loss = torch.nn.functional.cross_entropy(scores.view(-1, cfg["vocab_size"]), mlm_train_targets(batch["text_input_ids"]).view(-1))
        else:                                        # run_video_qa.py:484-501: mean of the clips' logits, then calc_loss
            logits = torch.stack([model.classifier(pooled) for _seq, pooled in feats]).mean(0)
            _, per = model.calc_loss(logits, batch["labels"])
            loss = per.mean()
        opt.zero_grad()
        loss.backward()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")           # the reference's AdamW uses the deprecated add_(Number, Tensor) overloads
            opt.step()
        if verbose and (step % 50 == 0 or step == ht["steps"] - 1):
            print(f"[train_head {name}] step {step} loss {float(loss):.5f}", flush=True)
    np.savez_compressed(head_override_path(name), **{n: p.detach().numpy() for n, p in train.items()})
    _SD_CACHE.clear()


@torch.no_grad()
def run_reference(name: str):
    """Reference transformer classes on top of the oracle CNN.  Returns dict of np arrays."""
    cfg, head, sd, batch = build_case(name)
    mo, _tr = ref_shim.load_reference_modeling()
    model = getattr(mo, REF_CLASS[head])(ref_shim.make_config(cfg)).eval()
    tsd = {k[len("transformer."):]: v for k, v in sd.items() if k.startswith("transformer.")}
    missing, unexpected = model.load_state_dict(tsd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    out = {}
    taps = {}
    grid = O.grid_feat_backbone(sd, batch["visual_inputs"], "cnn.", taps)
    for k, v in taps.items():
        out["fp_" + k] = fingerprint(v)
    out["fp_grid"] = fingerprint(grid)
    out["grid"] = grid.numpy()
    grid_r = O.repeat_rows(grid, batch["n_examples_list"])
    hooks = []
    bert = model.bert
    hooks.append(bert.embeddings.register_forward_hook(
        lambda m, i, o: out.__setitem__("fp_text_emb", fingerprint(o))))
    hooks.append(bert.visual_embeddings.register_forward_hook(
        lambda m, i, o: out.__setitem__("fp_vis_emb", fingerprint(o))))
    for li, layer in enumerate(bert.encoder.layer):
        hooks.append(layer.register_forward_hook(
            lambda m, i, o, li=li: out.__setitem__(f"fp_layer{li}", fingerprint(o[0]))))
    hooks.append(bert.pooler.register_forward_hook(
        lambda m, i, o: out.__setitem__("pooled", o.numpy().copy())))
    kw = {k: v for k, v in batch.items() if k not in ("visual_inputs", "n_examples_list")}
    kw["visual_inputs"] = grid_r
    if head == "retrieval":
        kw["sample_size"] = len(batch["n_examples_list"])
    res = model(**kw)
    for h in hooks:
        h.remove()
    if head == "pretraining":
        out["itm_scores"] = res["itm_scores"].numpy()
        out["itm_loss"] = res["itm_loss"].numpy()
        out["mlm_loss"] = res["mlm_loss"].numpy()
        out["mlm_argmax"] = res["mlm_scores"].argmax(-1).numpy()
        out["mlm_scores_strided"] = res["mlm_scores"][..., ::509].numpy()
        top2 = res["mlm_scores"].topk(2, dim=-1)[0]
        out["mlm_margin_min"] = np.float32((top2[..., 0] - top2[..., 1]).min().item())      # how decided the arg-max is
    else:
        out["logits"] = res["logits"].numpy()
        out["loss"] = res["loss"].numpy()
    return out


@torch.no_grad()
def run_reference_clips(name: str):
    """The reference's clip loop around its own transformer classes (CNN half: the oracle), then the runner's pooling and
    loss / score arithmetic exactly as written in src/tasks/run_video_retrieval.py:402-419 (training), :669-690
    (inference) and src/tasks/run_video_qa.py:484-501."""
    c = CLIP_CASES[name]
    cfg, head, sd, batch = build_case(name)
    mo, _tr = ref_shim.load_reference_modeling()
    model = getattr(mo, REF_CLASS[head])(ref_shim.make_config(cfg)).eval()
    tsd = {k[len("transformer."):]: v for k, v in sd.items() if k.startswith("transformer.")}
    missing, unexpected = model.load_state_dict(tsd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    bsz, num_clips, num_frm = c["n_videos"], c["n_clips"], c["n_frames"]
    vis = batch["visual_inputs"]
    visual_inputs = vis.view(bsz, num_clips, num_frm, *vis.shape[2:])           # :394-395
    logits = []
    out = {}
    for clip_idx in range(num_clips):
        grid = O.grid_feat_backbone(sd, visual_inputs[:, clip_idx], "cnn.")
        if clip_idx == 0:
            out["fp_grid_clip0"] = fingerprint(grid)
        kw = dict(text_input_ids=batch["text_input_ids"], text_input_mask=batch["text_input_mask"], labels=None,
                  visual_inputs=O.repeat_rows(grid, batch["n_examples_list"]))
        if head == "retrieval":
            kw["sample_size"] = bsz
        logits.append(model(**kw)["logits"])
    logits = torch.stack(logits)                                                # :402
    out["stack"] = logits.numpy().copy()
    pool_method = c["pool"]
    if pool_method == "mean":
        pooled = logits.mean(0)
    elif pool_method == "max":
        pooled = logits.max(0)[0]
    else:
        pooled = logits.permute(1, 0, 2).contiguous()
    labels = batch["labels"]
    if c["mode"] == "train":
        if pool_method == "lse":                                                # :413-417
            o = torch.logsumexp(pooled.view(pooled.shape[0], -1), dim=-1, keepdim=True) - torch.logsumexp(pooled, dim=1)
            loss = torch.gather(o, -1, labels.view(-1, 1))
        elif head == "retrieval":
            _, loss = model.calc_loss(pooled, labels, sample_size=bsz)
        else:
            pooled, loss = model.calc_loss(pooled, labels)                      # (B, 5) for the multiple-choice head
        out["pooled"] = pooled.numpy().copy()
        out["loss"] = loss.numpy().copy()
        if head == "multiple_choice":
            out["answer_ids"] = pooled.max(dim=-1)[1].numpy()                   # run_video_qa.py:273-275
    else:
        if pool_method == "lse":
            pooled = torch.logsumexp(pooled, dim=1)                             # :674-676
        probs = torch.nn.functional.softmax(pooled, dim=1)[:, 1].tolist()       # :681
        out["pooled"] = pooled.numpy().copy()
        out["scores"] = np.asarray([round(s_, 4) for s_ in probs], dtype=np.float64)      # :687
    return out


def main():
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    only = set(sys.argv[1:])
    for name in list(CASES) + list(CLIP_CASES):
        if only and name not in only:
            continue
        if name in HEAD_TRAIN:
            train_head(name)
        out = run_reference(name) if name in CASES else run_reference_clips(name)
        path = os.path.join(GOLDEN_DIR, name + ".npz")
        np.savez_compressed(path, **out)
        print(name, {k: v.shape for k, v in out.items() if not k.startswith("fp_")},
              os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
