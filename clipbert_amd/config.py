"""The reference's JSON config surface for the hot path (src/configs/*.json + the defaults of src/configs/config.py):
load a task config, build the model config the way setup_model does, build the model and the 8-group optimizer.

    cfg = load_task_config("src/configs/msrvtt_ret_base_resnet50.json", task="video_retrieval", train_n_clips=1)
    model = setup_model(cfg, device)                    # run_video_retrieval.py:181-216
    optimizer = setup_optimizer(model, cfg)             # src/optimization/utils.py:96-161 (setup_e2e_optimizer)

Dataset / tokenizer / output keys of the JSON files are carried along untouched (nothing here reads them)."""
import json
import os
from typing import Optional

import torch


class Config(dict):
    """dict with attribute access, nested (the reference uses EasyDict): cfg.train_datasets[0].name works."""
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, Config):
            return Config(v)
        if isinstance(v, (list, tuple)):
            return type(v)(Config._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, Config._wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    __setattr__ = __setitem__


# defaults of src/configs/config.py for the keys the hot path reads (:66-211, 285-344)
SHARED_DEFAULTS = dict(
    max_txt_len=20, max_img_size=448, img_pixel_mean=None, img_pixel_std=None, img_input_format="BGR", num_frm=3, train_n_clips=3,
    score_agg_func="mean", gradient_accumulation_steps=1, learning_rate=5e-5, betas=[0.9, 0.98], decay="linear", dropout=0.1,
    weight_decay=1e-3, grad_norm=2.0, warmup_ratio=0.1, transformer_lr_mul=1.0, transformer_lr_mul_prefix="", step_decay_epochs=None,
    optim="adamw", cnn_optim="adamw", cnn_learning_rate=5e-5, cnn_weight_decay=1e-3, cnn_lr_mul=1.0, cnn_lr_mul_prefix="grid_encoder",
    cnn_lr_decay="linear", cnn_step_decay_epochs=None, freeze_cnn=0, inference_batch_size=64, inference_n_clips=1,
    e2e_weights_path=None, detectron2_weights_path=None, bert_weights_path=None, detectron2_model_cfg="", model_config=None, seed=42)
TASK_DEFAULTS = {
    "pretraining": dict(pixel_random_sampling_size=0, use_itm=1, use_mlm=1),
    "video_retrieval": dict(itm_neg_size=1, classifier="mlp", cls_hidden_scale=2, margin=0.2, loss_type="ce"),
    "video_qa": dict(classifier="mlp", cls_hidden_scale=2, loss_type="ce", task="action"),
    "vqa": dict(classifier="mlp", cls_hidden_scale=2, loss_type="bce", num_labels=3129),
}


def load_task_config(path: str, task: str = "video_retrieval", **overrides) -> Config:
    """JSON file over the argparse defaults, keyword overrides over both (parse_with_config, config.py:12-30), then the
    task's derived keys (config.py:311-312, 349-362)."""
    assert task in TASK_DEFAULTS, task
    cfg = Config(SHARED_DEFAULTS)
    for k, v in TASK_DEFAULTS[task].items():
        cfg[k] = v
    with open(path) as fh:
        for k, v in json.load(fh).items():
            cfg[k] = v
    for k, v in overrides.items():
        cfg[k] = v
    if task == "video_retrieval":
        cfg["num_labels"] = 1 if cfg.loss_type == "rank" else 2
    elif task == "video_qa" and "num_labels" not in cfg:
        cfg["num_labels"] = 5 if cfg.task in ("action", "transition") else None     # frameqa / msrvtt_qa: len(ans2label)
    cfg["task_kind"] = task
    return cfg


def build_model_config(cfg: Config, config_root: Optional[str] = None) -> Config:
    """BertConfig(**load_json(cfg.model_config)) + the downstream keys (run_video_retrieval.py:184-193,
    run_video_qa.py: same list, run_pretrain.py: pixel_random_sampling_size)."""
    path = cfg.model_config
    if config_root is not None and not os.path.isabs(path):
        path = os.path.join(config_root, path)
    with open(path) as fh:
        mc = Config(json.load(fh))
    for k in ("num_labels", "classifier", "cls_hidden_scale", "loss_type", "margin", "pixel_random_sampling_size"):
        if k in cfg and cfg[k] is not None:
            mc[k] = cfg[k]
    return mc


_TRANSFORMER_CLS = {"pretraining": "ClipBertForPreTraining", "video_retrieval": "ClipBertForVideoTextRetrieval",
                    "vqa": "ClipBertForSequenceClassification"}


def setup_model(cfg: Config, device=None, dtype=torch.bfloat16, config_root: Optional[str] = None):
    """setup_model of the task runners (run_video_retrieval.py:181-216): model config, ClipBert, weights, freeze, device --
    plus prepare(): the move into the flat HBM parameter buffers that replaces amp.initialize."""
    from . import modeling as M
    kind = cfg.task_kind
    if kind == "video_qa":
        cls_name = "ClipBertForMultipleChoice" if cfg.task in ("action", "transition") else "ClipBertForSequenceClassification"
    else:
        cls_name = _TRANSFORMER_CLS[kind]
    mc = build_model_config(cfg, config_root)
    model = M.ClipBert(mc, input_format=cfg.img_input_format, detectron2_model_cfg=cfg.detectron2_model_cfg,
                       transformer_cls=getattr(M, cls_name))
    if cfg.e2e_weights_path:
        sd = torch.load(cfg.e2e_weights_path, map_location="cpu")
        M.load_state_dict_with_mismatch(model, sd.get("model", sd) if isinstance(sd, dict) else sd)
    elif cfg.detectron2_weights_path or cfg.bert_weights_path:
        model.load_separate_ckpt(cnn_weights_path=cfg.detectron2_weights_path, bert_weights_path=cfg.bert_weights_path)
    if cfg.freeze_cnn:
        model.freeze_cnn_backbone()
    if device is not None:
        model.to(device)
        model.prepare(dtype=dtype, device=device, transformer_lr_mul_prefix=cfg.transformer_lr_mul_prefix,
                      cnn_lr_mul_prefix=cfg.cnn_lr_mul_prefix)
    return model


def setup_optimizer(model, cfg: Config):
    """setup_e2e_optimizer (src/optimization/utils.py:96-128): AdamW over the 8 groups with the config's rates."""
    from .optim import FusedAdamW
    if cfg.optim != "adamw":
        raise ValueError("invalid optimizer" if cfg.optim not in ("adam", "adamax") else f"optimizer {cfg.optim} is not built (adamw is)")
    return FusedAdamW(model.rt.bank, lr=cfg.learning_rate, betas=tuple(cfg.betas), weight_decay=cfg.weight_decay,
                      cnn_lr=cfg.cnn_learning_rate, cnn_weight_decay=cfg.cnn_weight_decay, transformer_lr_mul=cfg.transformer_lr_mul,
                      cnn_lr_mul=cfg.cnn_lr_mul, max_grad_norm=cfg.grad_norm)
