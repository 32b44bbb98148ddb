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
    # config 4: TGIF-QA action, multiple choice with 5 options, Lt=25
    "tgif_mc": dict(head="multiple_choice", n_videos=2, n_frames=2, size=224, lt=25, repeat=5,
                    cfg=dict(num_labels=5, loss_type="ce")),
    # frame-QA style open-ended classifier (a18); small label space to keep the fixture small
    "seqcls_ce": dict(head="sequence_classification", n_videos=2, n_frames=1, size=224, lt=16,
                      repeat=1, cfg=dict(num_labels=37, loss_type="ce")),
}


_SD_CACHE = {}


def build_case(name: str, seed: int = 42):
    c = CASES[name]
    cfg = dict(O.BASE_CONFIG)
    cfg.update(c["cfg"])
    head = c["head"]
    key = (head, seed, tuple(sorted((k, str(v)) for k, v in cfg.items())))
    if key not in _SD_CACHE:          # ~150 M random values: generate once per process (callers never mutate it)
        _SD_CACHE.clear()
        _SD_CACHE[key] = S.full_state_dict(cfg, head, seed)
    sd = _SD_CACHE[key]
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
    else:
        out["logits"] = res["logits"].numpy()
        out["loss"] = res["loss"].numpy()
    return out


def main():
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    for name in CASES:
        out = run_reference(name)
        path = os.path.join(GOLDEN_DIR, name + ".npz")
        np.savez_compressed(path, **out)
        print(name, {k: v.shape for k, v in out.items() if not k.startswith("fp_")},
              os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
