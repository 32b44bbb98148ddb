"""The oracle must reproduce the committed golden vectors (tests/golden/*.npz), which were produced
by the REFERENCE's own transformer classes (oracle/make_golden.py).  Runs on CPU, on any host."""
import os

import numpy as np
import pytest
import torch

from oracle import clipbert_oracle as O
from oracle import make_golden as G

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name", list(G.CASES))
def test_oracle_matches_reference_golden(name):
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    gold = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg, head, sd, batch = G.build_case(name)
    taps = {}
    with torch.no_grad():
        out = O.clipbert_forward(sd, batch, cfg, head, taps)
    # CNN half is the same oracle code (self-consistency across hosts / torch builds)
    np.testing.assert_allclose(taps["grid"].numpy(), gold["grid"], rtol=1e-4, atol=1e-4)
    for i in range(cfg["num_hidden_layers"]):
        np.testing.assert_allclose(G.fingerprint(taps[f"layer{i}"]), gold[f"fp_layer{i}"],
                                   rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(taps["pooled"].numpy(), gold["pooled"], rtol=1e-4, atol=1e-4)
    if head == "pretraining":
        np.testing.assert_allclose(out["itm_scores"].numpy(), gold["itm_scores"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(out["itm_loss"].numpy(), gold["itm_loss"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(out["mlm_loss"].numpy(), gold["mlm_loss"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(out["mlm_scores"][..., ::509].numpy(), gold["mlm_scores_strided"],
                                   rtol=1e-4, atol=1e-4)
        assert (out["mlm_scores"].argmax(-1).numpy() == gold["mlm_argmax"]).all()
    else:
        np.testing.assert_allclose(out["logits"].numpy(), gold["logits"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(out["loss"].numpy(), gold["loss"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name", ["msrvtt_lse_c4_448", "msrvtt_infer_c16"])
def test_oracle_clip_loop_matches_reference_golden(name):
    """The oracle's clip loop + pooling + LSE loss / rounded scores (what bench.py's cpu_baseline and the emulator-sized
    task tests use as their checker) against the goldens produced by looping the REFERENCE's classes."""
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    c = G.CLIP_CASES[name]
    gold = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg, head, sd, batch = G.build_case(name)
    vis = batch["visual_inputs"].view(c["n_videos"], c["n_clips"], c["n_frames"], *batch["visual_inputs"].shape[2:])
    per_clip = []
    with torch.no_grad():
        for k in range(c["n_clips"]):
            b = dict(visual_inputs=vis[:, k], text_input_ids=batch["text_input_ids"], text_input_mask=batch["text_input_mask"],
                     n_examples_list=list(batch["n_examples_list"]))
            per_clip.append(O.clipbert_forward(sd, b, cfg, head)["logits"])
    np.testing.assert_allclose(torch.stack(per_clip).numpy(), gold["stack"], rtol=1e-4, atol=2e-5)
    pooled = O.aggregate_clip_logits(per_clip, c["pool"])
    if c["mode"] == "train":
        np.testing.assert_allclose(O.lse_train_loss(pooled, batch["labels"]).numpy(), gold["loss"], rtol=1e-4, atol=2e-5)
    else:
        probs = torch.softmax(torch.logsumexp(pooled, dim=1), dim=1)[:, 1].tolist()
        assert max(abs(round(p, 4) - g) for p, g in zip(probs, gold["scores"].tolist())) <= 1.01e-4
