"""The oracle's bf16 YARDSTICK modes (oracle/clipbert_oracle.py ``precision``) and the committed constants derived from them
(tests/golden/bf16_yardstick.json, tests/parity_bounds.py).  CPU only; nothing here touches the product."""
import json
import os

import numpy as np
import pytest
import torch

import parity_bounds as PB
from clipbert_amd import synthetic as S
from oracle import clipbert_oracle as O
from oracle import make_bf16_yardstick as Y
from oracle import make_golden as G

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _small_case(layers=2, size=64):
    cfg = dict(O.BASE_CONFIG, num_hidden_layers=layers, num_labels=2, loss_type="ce", margin=0.1)
    sd = S.full_state_dict(cfg, "retrieval", 5)
    frames = S.synthetic_frames(2, 2, size, 5)
    ids, mask = S.synthetic_text(4, 12, 5)
    batch = dict(visual_inputs=O.image_norm(frames, S.PIXEL_MEAN, S.PIXEL_STD), text_input_ids=ids, text_input_mask=mask,
                 n_examples_list=[2, 2], labels=torch.tensor([1, 0, 1, 0]))
    return cfg, sd, batch


def test_fp32_mode_is_untouched_and_bf16_modes_round_storage():
    cfg, sd, batch = _small_case()
    with torch.no_grad():
        base = O.clipbert_forward(sd, batch, cfg, "retrieval")["logits"]
        with O.precision("fp32"):
            same = O.clipbert_forward(sd, batch, cfg, "retrieval")["logits"]
        assert torch.equal(base, same)                                   # the oracle proper is bit-identical with the switch in place
        outs = {}
        for mode in ("bf16", "bf16_fused"):
            taps = {}
            with O.precision(mode):
                outs[mode] = O.clipbert_forward(sd, batch, cfg, "retrieval", taps)["logits"]
            assert O.PRECISION == "fp32"                                 # the context restores the mode
            for k in ("grid_conv", "embeddings", "layer0", "pooled"):     # stored activations are bf16-representable in both modes
                assert torch.equal(taps[k], taps[k].bfloat16().float()), (mode, k)
            err = (outs[mode] - base).abs().max().item()
            assert 0 < err < 0.05 * max(1.0, base.abs().max().item()), (mode, err)
        assert not torch.equal(outs["bf16"], outs["bf16_fused"])         # two different rounding granularities


def test_bf16_mode_rounds_activation_gradients_but_not_weight_gradients():
    x = torch.randn(8, 16, requires_grad=True)
    w = torch.randn(4, 16, requires_grad=True)
    with O.precision("bf16"):
        y = O._r(torch.nn.functional.linear(O._r(x), O._w(w)))
        (y * torch.randn(8, 4)).sum().backward()
    assert torch.equal(x.grad, x.grad.bfloat16().float())               # activation gradient: stored in bf16
    assert not torch.equal(w.grad, w.grad.bfloat16().float())           # weight gradient: accumulated in fp32
    # value path: the product of ROUNDED operands
    with torch.no_grad():
        ref = torch.nn.functional.linear(x.bfloat16().float(), w.bfloat16().float()).bfloat16().float()
    assert torch.equal(y.detach(), ref)


def test_committed_yardstick_reproduces_on_this_host():
    """one cheap case re-run here: the committed figure is what the script gives (to within what another thread count's fp32
    summation order moves a bf16 rounding decision)"""
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    for mode in PB.MODES:
        rec = Y.forward_errors("retrieval_rank", mode)
        com = PB.YARDSTICK["retrieval_rank"][mode]
        assert abs(rec["logits"] - com["logits"]) <= 0.25 * com["logits"], (mode, rec, com)
        assert rec["logit_scale"] == pytest.approx(com["logit_scale"], rel=1e-4)


def test_yardstick_covers_every_golden_and_its_margins_are_decided():
    for name in list(G.CASES) + list(G.CLIP_CASES):
        assert name in PB.YARDSTICK, name
        for mode in PB.MODES:
            assert PB.YARDSTICK[name][mode], (name, mode)
    # configs[3] / pretraining: the goldens' arg-max must be decided against bf16 noise, else asserting it in bf16 is vacuous:
    # margin > 2 x bound = 3 x yardstick error is the logical requirement (tests/test_parity_record.py), 10 x the comfortable one
    assert PB.margin_over_error("tgif_mc") >= 10
    assert PB.margin_over_error("pretrain_cfg1") >= 10
    assert PB.margin_over_error("tgif_mc_c2_768") > 3.3                   # 768 px / L = 169: see oracle/make_golden.py HEAD_TRAIN
    for name in G.HEAD_TRAIN:
        for mode in PB.MODES:
            y = PB.YARDSTICK[name][mode]
            assert y.get("answer_ids_agree", 1.0) == 1.0 and y.get("mlm_argmax_agreement", 1.0) == 1.0, (name, mode, y)
    g = PB.grad_yardstick()
    assert 0 < g["median_tensor_rel_l2"] < 0.5 and 0 < g["one_minus_cosine"] < 0.05 and len(g["per_tensor"]) > 200


def test_trained_heads_overlay_the_synthetic_state_dict():
    for name, ht in G.HEAD_TRAIN.items():
        over = np.load(G.head_override_path(name))
        assert over.files and all(k.startswith(ht["params"]) for k in over.files), (name, over.files)
    cfg, head, sd, batch = G.build_case("tgif_mc")
    raw = G.build_case("tgif_mc", trained_head=False)[2]
    k = "transformer.classifier.2.weight"
    assert not torch.equal(sd[k], raw[k]) and torch.equal(sd["transformer.classifier.0.weight"], raw["transformer.classifier.0.weight"])
    assert json.dumps(sorted(sd)) == json.dumps(sorted(raw))
