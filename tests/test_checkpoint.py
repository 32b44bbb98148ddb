"""Checkpoint compatibility: reference key layout, save -> load round trip after an optimiser
step (parameters live in the flat HBM buffers; state_dict() must still show logical OIHW / (out, in) tensors), tolerant
loading of reference checkpoints that carry the dead detectron2 RPN / ROI-head weights (src/utils/load_save.py:71-100)."""
import io

import pytest
import torch

from clipbert_amd import modeling as M
from clipbert_amd import optim
from clipbert_amd import synthetic as S
from oracle import ref_shim
from test_model_small import HEAD_CLS, SMALL, build, make_batch, to_dev

RET = dict(num_labels=2, loss_type="ce", margin=0.1)


@pytest.mark.skipif(not ref_shim.available(), reason="needs /root/reference")
@pytest.mark.parametrize("head,ref_cls", [("retrieval", "ClipBertForVideoTextRetrieval"), ("pretraining", "ClipBertForPreTraining"),
                                          ("multiple_choice", "ClipBertForMultipleChoice")])
def test_transformer_keys_and_shapes_match_reference(head, ref_cls):
    cfg = dict(SMALL, **(RET if head == "retrieval" else dict(num_labels=5, loss_type="ce") if head == "multiple_choice" else {}))
    mo, _ = ref_shim.load_reference_modeling()
    ref = getattr(mo, ref_cls)(ref_shim.make_config(cfg)).state_dict()
    ours = M.ClipBert(cfg, detectron2_model_cfg="R-50-grid.yaml", transformer_cls=HEAD_CLS[head]).transformer.state_dict()
    assert set(ours) == set(ref), (sorted(set(ours) ^ set(ref))[:10])
    for k in ref:
        assert tuple(ours[k].shape) == tuple(ref[k].shape), k


def test_save_load_round_trip_after_a_step(hw):
    cfg, sd, model = build("retrieval", RET, torch.float32, hw.dev)
    batch = to_dev(make_batch(cfg, "retrieval", 2, 2, 6), hw.dev)
    batch["labels"] = S.synthetic_labels(4, 2, 5).to(hw.dev)
    opt = optim.FusedAdamW(model.rt.bank, lr=1e-2, betas=(0.9, 0.98), weight_decay=1e-3, cnn_lr=1e-2, max_grad_norm=5.0)
    opt.zero_grad()
    model(dict(batch, n_examples_list=[2, 2]))["loss"].mean().backward()
    opt.step()
    with torch.no_grad():
        after = model(dict(batch, n_examples_list=[2, 2]))["logits"].clone()
    buf = io.BytesIO()
    torch.save(model.state_dict(), buf)                      # what save_checkpoint would write (model_step_N.pt)
    buf.seek(0)
    loaded = torch.load(buf, map_location="cpu")
    w = loaded["cnn.feature.backbone.res4.0.conv2.weight"]
    assert w.dim() == 4 and w.shape[2:] == (3, 3)            # logical OIHW, whatever the memory image is
    assert any((loaded[k] - sd[k]).abs().max() > 0 for k in loaded if k.endswith("conv2.weight") and "res4" in k)   # trained
    fresh = M.ClipBert(cfg, detectron2_model_cfg="R-50-grid.yaml", transformer_cls=HEAD_CLS["retrieval"])
    fresh.load_state_dict(loaded, strict=True)
    fresh.to(hw.dev).eval()
    fresh.prepare(dtype=torch.float32, device=hw.dev)
    with torch.no_grad():
        again = fresh(dict(batch, n_examples_list=[2, 2]))["logits"]
    torch.testing.assert_close(again, after, rtol=0, atol=0)


def test_tolerant_load_drops_dead_and_mismatched_keys(hw):
    cfg, sd, model = build("retrieval", RET, torch.float32, hw.dev)
    ckpt = {k: v.clone() + 0.5 for k, v in sd.items() if v.is_floating_point()}
    ckpt["cnn.feature.proposal_generator.rpn_head.conv.weight"] = torch.zeros(4, 4, 3, 3)     # dead detectron2 weights
    ckpt["cnn.feature.roi_heads.box_predictor.cls_score.bias"] = torch.zeros(7)
    ckpt["transformer.classifier.2.weight"] = torch.zeros(17, 3)                              # head of another task
    M.load_state_dict_with_mismatch(model, ckpt)
    now = model.state_dict()
    key = "transformer.bert.encoder.layer.0.output.dense.weight"
    torch.testing.assert_close(now[key].cpu(), sd[key] + 0.5)
    assert tuple(now["transformer.classifier.2.weight"].shape) != (17, 3)                 # shape-mismatched entry ignored
