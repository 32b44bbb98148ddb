"""Live check of the oracle against the reference's own code (only where /root/reference exists).
Covers what the fixtures do not: train-mode-free paths with random grid inputs for every head, the
pixel sub-sampling branch, and the LSE clip aggregation arithmetic of the runner."""
import pytest
import torch

from clipbert_amd import synthetic as S
from oracle import clipbert_oracle as O
from oracle import ref_shim

pytestmark = pytest.mark.skipif(not ref_shim.available(), reason="reference tree not present")

SMALL = dict(O.BASE_CONFIG, num_hidden_layers=2, vocab_size=2000, max_position_embeddings=64)


def _ref(head, cfg):
    mo, _ = ref_shim.load_reference_modeling()
    cls = dict(retrieval=mo.ClipBertForVideoTextRetrieval, multiple_choice=mo.ClipBertForMultipleChoice,
               sequence_classification=mo.ClipBertForSequenceClassification, regression=mo.ClipBertForRegression,
               pretraining=mo.ClipBertForPreTraining)[head]
    model = cls(ref_shim.make_config(cfg)).eval()
    sd = S.transformer_state_dict(cfg, head, 7, "")
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    return model, {"transformer." + k: v for k, v in sd.items()}


@pytest.mark.parametrize("head,extra", [
    ("retrieval", dict(num_labels=2, loss_type="ce", margin=0.1)),
    ("retrieval", dict(num_labels=1, loss_type="rank", margin=0.2)),
    ("multiple_choice", dict(num_labels=5, loss_type="ce")),
    ("sequence_classification", dict(num_labels=11, loss_type="ce")),
    ("sequence_classification", dict(num_labels=11, loss_type="bce")),
    ("sequence_classification", dict(num_labels=1, loss_type="ce")),
    ("regression", dict(num_labels=1, loss_type="mse")),
])
def test_heads_bit_close(head, extra):
    cfg = dict(SMALL, **extra)
    model, sd = _ref(head, cfg)
    n = 10 if head == "multiple_choice" else 4
    ids, mask = S.synthetic_text(n, 12, 3, cfg["vocab_size"])
    grid = torch.randn(n, 2, 3, 4, 768, generator=S._gen(3, "grid"))
    if head == "multiple_choice":
        labels = S.synthetic_labels(2, 5, 3)
    elif extra["loss_type"] == "bce":
        labels = torch.rand(n, 11, generator=S._gen(3, "bce"))
    elif extra["num_labels"] == 1:
        labels = torch.randn(n, generator=S._gen(3, "mse"))
    else:
        labels = S.synthetic_labels(n, max(2, extra["num_labels"]), 3)
    kw = dict(sample_size=2) if head == "retrieval" else {}
    with torch.no_grad():
        r = model(ids, grid, mask, labels=labels, **kw)
        o = O.HEADS[head](sd, ids, grid, mask, cfg, labels=labels, **kw)
    torch.testing.assert_close(o["logits"], r["logits"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(o["loss"], r["loss"], rtol=1e-5, atol=1e-6)


def test_pretraining_with_pixel_subsampling():
    import numpy as np
    cfg = dict(SMALL, pixel_random_sampling_size=5)
    model, sd = _ref("pretraining", cfg)
    model.train()          # sub-sampling only fires in training mode (modeling.py:80-81)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    ids, mask = S.synthetic_text(3, 10, 5, cfg["vocab_size"])
    grid = torch.randn(3, 1, 3, 3, 768, generator=S._gen(5, "grid"))
    mlm = ids.clone()
    mlm[:, ::2] = -100
    itm = S.synthetic_labels(3, 2, 5)
    np.random.seed(123)
    idx = torch.from_numpy(np.sort(np.random.choice(9, size=5, replace=False))).long()
    np.random.seed(123)    # the reference draws the same indices from numpy's global RNG
    with torch.no_grad():
        r = model(ids, grid, mask, mlm_labels=mlm, itm_labels=itm)
        o = O.pretraining_forward(sd, ids, grid, mask, cfg, mlm, itm, sample_idx=idx)
    torch.testing.assert_close(o["mlm_scores"], r["mlm_scores"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(o["itm_scores"], r["itm_scores"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(o["mlm_loss"], r["mlm_loss"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(o["itm_loss"], r["itm_loss"], rtol=1e-5, atol=1e-6)
