"""The JSON config surface: the reference's own src/configs files load, the model config carries the downstream keys the
way setup_model adds them, and the optimizer gets the 8 groups with the configured rates."""
import json
import os

import pytest
import torch

from clipbert_amd import config as C

REF_CFG = "/root/reference/src/configs"
needs_ref = pytest.mark.skipif(not os.path.isdir(REF_CFG), reason="needs /root/reference")


def test_config_attribute_access_and_nesting():
    c = C.Config(dict(a=1, train_datasets=[dict(name="msrvtt", txt="/t")]))
    assert c.a == 1 and c.train_datasets[0].name == "msrvtt"
    c.b = dict(x=2)
    assert c["b"].x == 2
    with pytest.raises(AttributeError):
        c.missing


@needs_ref
@pytest.mark.parametrize("fname,task,expect", [
    ("msrvtt_ret_base_resnet50.json", "video_retrieval", dict(num_labels=2, score_agg_func="lse", train_n_clips=8, num_frm=2, max_txt_len=20)),
    ("tgif_qa_action_base_resnet50.json", "video_qa", dict(num_labels=5)),
    ("pretrain_image_text_base_resnet50_mlm_itm.json", "pretraining", dict()),
])
def test_reference_json_files_load(fname, task, expect):
    cfg = C.load_task_config(os.path.join(REF_CFG, fname), task=task)
    raw = json.load(open(os.path.join(REF_CFG, fname)))
    for k, v in raw.items():
        assert cfg[k] == v                                     # the file wins over the defaults
    for k, v in expect.items():
        assert cfg[k] == v, (k, cfg[k])
    assert cfg.cnn_lr_mul_prefix == "grid_encoder" and cfg.warmup_ratio == 0.1          # argparse defaults fill the rest
    mc = C.build_model_config(cfg, config_root="/root/reference")
    assert mc.hidden_size == 768 and mc.num_hidden_layers == 12 and mc.vocab_size == 30522 and mc.layer_norm_eps == 1e-12
    if task == "video_retrieval":
        assert (mc.num_labels, mc.loss_type, mc.classifier, mc.cls_hidden_scale) == (2, "ce", "mlp", 2)
    ranked = C.load_task_config(os.path.join(REF_CFG, fname), task=task, loss_type="rank") if task == "video_retrieval" else None
    if ranked is not None:
        assert ranked.num_labels == 1


def test_setup_model_and_optimizer_small(tmp_path, emul):
    """setup_model / setup_optimizer on a tiny model config written in the reference's format."""
    model_json = dict(max_temporal_position_embeddings=100, backbone_channel_in_size=2048, max_grid_row_position_embeddings=100,
                      max_grid_col_position_embeddings=100, attention_probs_dropout_prob=0.1, hidden_act="gelu", hidden_dropout_prob=0.1,
                      hidden_size=128, initializer_range=0.02, intermediate_size=256, layer_norm_eps=1e-12, max_position_embeddings=32,
                      model_type="bert", num_attention_heads=2, num_hidden_layers=2, pad_token_id=0, type_vocab_size=2, vocab_size=200)
    (tmp_path / "model.json").write_text(json.dumps(model_json))
    task_json = dict(model_config=str(tmp_path / "model.json"), detectron2_model_cfg="R-50-grid.yaml", num_frm=2, train_n_clips=1,
                     learning_rate=1e-4, cnn_learning_rate=2e-5, weight_decay=1e-3, cnn_weight_decay=1e-4, grad_norm=5.0, loss_type="ce",
                     transformer_lr_mul=2.0, cnn_lr_mul=3.0)
    (tmp_path / "task.json").write_text(json.dumps(task_json))
    cfg = C.load_task_config(str(tmp_path / "task.json"), task="video_retrieval")
    model = C.setup_model(cfg, device=torch.device("cpu"), dtype=torch.float32)
    assert type(model.transformer).__name__ == "ClipBertForVideoTextRetrieval"
    opt = C.setup_optimizer(model, cfg)
    assert len(opt.param_groups) == 8
    lrs = [g["lr"] for g in opt.param_groups]
    wds = [g["weight_decay"] for g in opt.param_groups]
    assert lrs[2] == pytest.approx(1e-4) and lrs[6] == pytest.approx(2e-5)
    assert lrs[4] == pytest.approx(3.0 * 2e-5)                 # grid_encoder (cnn_lr_mul_prefix) gets the multiplier
    assert wds[0] in (1e-3, 0.0) and 1e-4 in wds and 0.0 in wds
    cfg2 = C.load_task_config(str(tmp_path / "task.json"), task="video_retrieval", freeze_cnn=1)
    frozen = C.setup_model(cfg2, device=torch.device("cpu"), dtype=torch.float32)
    assert all(not p.requires_grad for _n, p in frozen.cnn.feature.named_parameters())
