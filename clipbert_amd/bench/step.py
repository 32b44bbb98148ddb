"""The 1-GPU training step of the metric configuration, built exactly as bench.py's default path builds it (same model, batch,
optimizer, closures), for diagnostics and tests that need the step without bench.py's launcher / plans:

    st = build()                       # model + batch + optimizer on cuda:0
    st.host_prepare(); loss = st.device_step()         # one eager step
    g, loss = st.capture()             # the step as a hipGraph;  st.host_prepare(); g.replay()

bench.py's `device_step_single` / `host_prepare` / `forward_loss` are these functions (bench.py imports this module), so a test on
this step is a test on the thing that is timed.
"""
import os
import sys
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BASE_CONFIG = dict(
    max_temporal_position_embeddings=100, backbone_channel_in_size=2048, max_grid_row_position_embeddings=100,
    max_grid_col_position_embeddings=100, attention_probs_dropout_prob=0.1, hidden_act="gelu", hidden_dropout_prob=0.1,
    hidden_size=768, initializer_range=0.02, intermediate_size=3072, layer_norm_eps=1e-12, max_position_embeddings=512,
    model_type="bert", num_attention_heads=12, num_hidden_layers=12, pad_token_id=0, type_vocab_size=2, vocab_size=30522,
    num_labels=2, loss_type="ce", margin=0.1)


def build(videos=16, n_clips=2, frames=2, size=224, txt_len=32, repeat=2, pool="lse", seed=42, dropout=True, dev=None, cfg_over=None):
    from clipbert_amd import modeling as M
    from clipbert_amd import ops
    from clipbert_amd import synthetic as S
    from clipbert_amd import tasks
    from clipbert_amd.dist import GradSync
    from clipbert_amd.optim import FusedAdamW

    dev = dev or torch.device("cuda", 0)
    cfg = dict(BASE_CONFIG)
    if not dropout:
        cfg.update(attention_probs_dropout_prob=0.0, hidden_dropout_prob=0.0)
    if cfg_over:
        cfg.update(cfg_over)
    model = M.ClipBert(cfg, detectron2_model_cfg="R-50-grid.yaml", transformer_cls=M.ClipBertForVideoTextRetrieval)
    sd = S.full_state_dict(cfg, "retrieval", seed)
    model.load_state_dict(sd, strict=True)
    model.to(dev)
    model.train(True)
    model.prepare(dtype=torch.bfloat16, device=dev, overlap_wgrad=int(os.environ.get("CB_OVERLAP_WGRAD", "0")))
    bank = model.rt.bank
    fr = S.synthetic_frames(videos, n_clips * frames, size, seed).to(dev)
    ids, mask = S.synthetic_text(videos * repeat, txt_len, seed)
    ids, mask = ids.to(dev), mask.to(dev)
    tcfg = SimpleNamespace(train_n_clips=n_clips, inference_n_clips=n_clips, num_frm=frames, score_agg_func=pool, task=None,
                           num_labels=cfg["num_labels"], inference_batch_size=repeat, gradient_accumulation_steps=1, learning_rate=5e-5,
                           cnn_learning_rate=5e-5, decay="linear", cnn_lr_decay="linear", num_train_steps=100000, warmup_ratio=0.1)
    labels = torch.tensor(([1] + [0] * (repeat - 1)) * videos, dtype=torch.long, device=dev)
    counts = [repeat] * videos
    batch = dict(visual_inputs=fr, text_input_ids=ids, text_input_mask=mask, labels=labels, n_examples_list=counts)
    sync = GradSync(bank, compress="bf16", comm="auto")
    opt = FusedAdamW(bank, lr=5e-5, betas=(0.9, 0.98), weight_decay=1e-3, max_grad_norm=5.0, fold_norm=os.environ.get("CB_BENCH_NO_FOLD") is None)
    fns = make_step(model, batch, tcfg, opt, sync, labels, counts, n_clips, frames, pool)

    def capture(fn=None):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = (fn or fns.device_step)()
        return g, out

    return SimpleNamespace(model=model, cfg=cfg, state_dict=sd, batch=batch, tcfg=tcfg, labels=labels, counts=counts, opt=opt, sync=sync, bank=bank,
                           forward_loss=fns.forward_loss, host_prepare=fns.host_prepare, device_step=fns.device_step, capture=capture, state=fns.state,
                           clips_per_step=videos * n_clips, dev=dev)


def make_step(model, batch, tcfg, opt, sync, labels, counts, n_clips, frames, pool, fold=True):
    """The closures of one training step on prepared objects -- bench.py's default path calls THIS (its `forward_loss`, `host_prepare`
    and `device_step_single` are these functions), so tests/test_bench_step.py tests what is timed."""
    from clipbert_amd import ops
    from clipbert_amd import tasks
    state = {"global_step": 0}
    one = torch.ones((), dtype=torch.float32, device=labels.device)          # d(loss)/d(loss): persistent, so that backward() launches no fill

    def forward_loss():
        stack = tasks.forward_clips_stack(model, batch, n_clips, frames, fold=fold, cfg=tcfg)       # (n_clips, pairs, C) logits
        return tasks.training_loss(model, stack, labels, counts, pool)                              # clip pooling (a20) + loss

    def host_prepare():
        """per-step host work of a real training loop: LR schedule onto the 8 groups, hyper-parameter upload"""
        state["global_step"] += 1
        tasks.set_learning_rates(opt, tcfg, state["global_step"])
        opt.prepare_step(grad_scale=sync.grad_scale)

    def device_step():
        """everything a 1-GPU step enqueues (capturable)"""
        opt.zero_grad(lazy=True)
        model.rt.pending_encoder_nodes = model.rt.pending_cnn_nodes = 0
        loss = forward_loss()
        loss.backward(one)
        ops.counter_add(model.rt.seed_dev)
        opt.launch()
        return loss

    return SimpleNamespace(forward_loss=forward_loss, host_prepare=host_prepare, device_step=device_step, state=state, one=one)
