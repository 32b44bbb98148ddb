"""Task-loop pieces: feature-cached retrieval inference == the reference's order of evaluation, the
training step over clips with each pooling method == the oracle's composition, metrics == their definition."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from clipbert_amd import optim, tasks
from clipbert_amd import synthetic as S
from oracle import clipbert_oracle as O
from test_model_small import build, to_dev

RET = dict(num_labels=2, loss_type="ce", margin=0.1)


def _frames(n_videos, n_frames, seed):
    f = S.synthetic_frames(n_videos, n_frames, 64, seed)[..., :64, :].repeat(1, 1, 1, 1, 2).contiguous()   # (Bv, n, 3, 64, 128)
    return O.image_norm(f, S.PIXEL_MEAN, S.PIXEL_STD)


@pytest.mark.parametrize("pool", ["lse"])
def test_cached_inference_equals_per_clip_loop(hw, pool):
    cfg, sd, model = build("retrieval", RET, torch.float32, hw.dev)
    model.eval()
    icfg = SimpleNamespace(inference_n_clips=2, num_frm=2, score_agg_func=pool, inference_batch_size=2)
    vis = hw(_frames(1, 4, 11))
    ids, mask = S.synthetic_text(3, 6, 11, cfg["vocab_size"])
    ids = hw(ids.clamp(max=cfg["vocab_size"] - 1))
    mask = hw(mask)
    fast = tasks.inference_retrieval_video(model, vis, ids, mask, icfg, cache_cnn=True)
    slow = tasks.inference_retrieval_video(model, vis, ids, mask, icfg, cache_cnn=False)
    assert len(fast) == len(slow) == 3
    assert max(abs(a - b) for a, b in zip(fast, slow)) <= 1.01e-4
    # and both equal the oracle's composition of the reference loop (run_video_retrieval.py:655-690)
    per_clip = []
    visc = vis.cpu().view(2, 2, *vis.shape[2:])
    for c in range(2):
        batch = dict(visual_inputs=visc[c:c + 1], text_input_ids=ids.cpu(), text_input_mask=mask.cpu(), n_examples_list=[3])
        with torch.no_grad():
            per_clip.append(O.clipbert_forward(sd, batch, cfg, "retrieval")["logits"])
    pooled = O.aggregate_clip_logits(per_clip, pool)
    if pool == "lse":
        pooled = torch.logsumexp(pooled, dim=1)
    ref = [round(s, 4) for s in torch.softmax(pooled, dim=1)[:, 1].tolist()]
    tol = 2e-4 if hw.name == "emul" else 2e-3
    assert max(abs(a - b) for a, b in zip(fast, ref)) <= tol


@pytest.mark.parametrize("pool", ["lse"])
def test_train_step_over_clips(hw, pool):
    cfg, sd, model = build("retrieval", RET, torch.float32, hw.dev)
    model.eval()                                            # dropout off: the loss is comparable with the oracle
    opt = optim.FusedAdamW(model.rt.bank, lr=1e-3, betas=(0.9, 0.98), weight_decay=1e-3, cnn_lr=1e-3, max_grad_norm=5.0)
    tcfg = SimpleNamespace(train_n_clips=2, num_frm=2, score_agg_func=pool, learning_rate=1e-3, cnn_learning_rate=1e-3, decay="linear",
                           cnn_lr_decay="multi_step", cnn_step_decay_epochs=[2, 4], num_train_steps=10, warmup_ratio=0.1,
                           transformer_lr_mul=2.0, cnn_lr_mul=3.0)
    vis = _frames(2, 4, 5)
    ids, mask = S.synthetic_text(4, 6, 5, cfg["vocab_size"])
    ids = ids.clamp(max=cfg["vocab_size"] - 1)
    labels = torch.tensor([1, 0, 1, 0])
    batch = to_dev(dict(visual_inputs=vis, text_input_ids=ids, text_input_mask=mask, labels=labels, n_examples_list=[2, 2]), hw.dev)
    # oracle composition of the same loop
    per_clip = []
    v = vis.view(2, 2, 2, *vis.shape[2:])
    for c in range(2):
        b = dict(visual_inputs=v[:, c], text_input_ids=ids, text_input_mask=mask, labels=labels, n_examples_list=[2, 2])
        with torch.no_grad():
            per_clip.append(O.clipbert_forward(sd, b, cfg, "retrieval")["logits"])
    pooled = O.aggregate_clip_logits(per_clip, pool)
    ref = O.lse_train_loss(pooled, labels).mean() if pool == "lse" else torch.nn.functional.cross_entropy(pooled, labels)
    p0 = next(p for n, p in model.named_parameters() if n.endswith("pooler.dense.weight"))
    w_before = p0.detach().float().cpu().clone()
    loss = tasks.train_step(model, opt, batch, tcfg, global_step=0, n_epoch=3)
    torch.testing.assert_close(loss.cpu(), ref, rtol=2e-3, atol=2e-4)
    lrs = [pg["lr"] for pg in opt.param_groups]
    lr_t = optim.get_lr_sched(1, "linear", 1e-3, 10, 0.1)
    lr_c = 1e-3 * 0.5                                       # epoch 3: one milestone (2) passed
    assert lrs[0] == pytest.approx(2.0 * lr_t) and lrs[2] == pytest.approx(lr_t)
    assert lrs[4] == pytest.approx(3.0 * lr_c) and lrs[6] == pytest.approx(lr_c)
    assert (p0.detach().float().cpu() - w_before).abs().max() > 0          # the step moved the weights


def test_multi_step_schedule():
    assert optim.multi_step_schedule(0, [2, 4]) == 1.0
    assert optim.multi_step_schedule(2, [4, 2]) == 0.5
    assert optim.multi_step_schedule(9, [2, 4]) == 0.5 ** 3          # the reference's gamma**(len+1) after the last milestone


def test_retrieval_metrics_definition():
    rng = np.random.default_rng(0)
    n_txt, n_vid = 40, 25
    sm = rng.random((n_txt, n_vid)).astype(np.float32)
    gt = rng.integers(0, n_vid, n_txt)
    got = tasks.retrieval_metrics_from_scores(sm, gt)
    ranks = np.array([1 + int((sm[i] > sm[i, gt[i]]).sum()) for i in range(n_txt)])      # no ties in random floats
    assert got["r1"] == pytest.approx(100.0 * (ranks <= 1).mean())
    assert got["r5"] == pytest.approx(100.0 * (ranks <= 5).mean())
    assert got["r10"] == pytest.approx(100.0 * (ranks <= 10).mean())
    assert got["medianR"] == pytest.approx(float(np.median(ranks))) and got["meanR"] == pytest.approx(float(ranks.mean()))
    # eval_retrieval: one caption per video, ids as strings, a duplicated row must be ignored
    n = 12
    sm2 = rng.random((n, n)).astype(np.float32)
    rows = [dict(vid_id=f"v{j}", txt_id=f"t{i}", score=float(sm2[i, j])) for i in range(n) for j in range(n)]
    rows.append(dict(vid_id="v0", txt_id="t0", score=123.0))
    res = tasks.eval_retrieval(rows, {f"t{i}": f"v{i}" for i in range(n)})
    assert res["text2video"] == tasks.retrieval_metrics_from_scores(sm2, list(range(n)))
    assert res["video2text"] == tasks.retrieval_metrics_from_scores(sm2.T, list(range(n)))
    assert tasks.qa_accuracy([1, 2, 3, 0], [1, 2, 0, 0]) == 75.0


def test_qa_predict_multiple_choice(hw):
    """TGIF-QA action style: 5 candidate answers per video, 2 clips, mean pooling; answer id = argmax (config 4)."""
    cfg, sd, model = build("multiple_choice", dict(num_labels=5, loss_type="ce"), torch.float32, hw.dev)
    model.eval()
    # the collated batch counts QUESTIONS per video ([1, 1]); forward_step multiplies by num_labels (run_video_qa.py:206-210)
    qcfg = SimpleNamespace(inference_n_clips=2, num_frm=2, score_agg_func="mean", task="action", num_labels=5)
    vis = _frames(2, 4, 21)
    ids, mask = S.synthetic_text(10, 6, 21, cfg["vocab_size"])
    ids = ids.clamp(max=cfg["vocab_size"] - 1)
    batch = dict(visual_inputs=vis, text_input_ids=ids, text_input_mask=mask, n_examples_list=[1, 1])
    pred = tasks.qa_predict(model, to_dev(batch, hw.dev), qcfg)
    v = vis.view(2, 2, 2, *vis.shape[2:])
    per_clip = []
    for c in range(2):
        with torch.no_grad():
            per_clip.append(O.clipbert_forward(sd, dict(visual_inputs=v[:, c], text_input_ids=ids, text_input_mask=mask, n_examples_list=[5, 5]),
                                               cfg, "multiple_choice")["logits"])
    ref = O.aggregate_clip_logits(per_clip, "mean")
    assert tuple(ref.shape) == (2, 5)
    assert pred == ref.max(dim=-1)[1].tolist()                # argmax-exact answer ids
    assert tasks.qa_accuracy(pred, pred) == 100.0
    if hw.name == "emul":                                     # all (clip, video) pairs in one forward: same answers
        assert tasks.qa_predict(model, to_dev(batch, hw.dev), qcfg, fold_clips=True) == pred


def test_gradient_accumulation_sums_micro_batches(hw):
    """two micro-steps with gradient_accumulation_steps=2 == one step whose gradient is the SUM of the two (:426-436)."""
    cfg, sd, model = build("retrieval", RET, torch.float32, hw.dev)
    model.eval()
    base = dict(train_n_clips=1, num_frm=2, score_agg_func="mean", learning_rate=1e-3, cnn_learning_rate=1e-3, decay="constant",
                cnn_lr_decay="constant", num_train_steps=10, warmup_ratio=0.0)
    vis = _frames(2, 2, 31)
    ids, mask = S.synthetic_text(4, 6, 31, cfg["vocab_size"])
    ids = ids.clamp(max=cfg["vocab_size"] - 1)
    labels = torch.tensor([1, 0, 0, 1])
    halves = [to_dev(dict(visual_inputs=vis[i:i + 1], text_input_ids=ids[2 * i:2 * i + 2], text_input_mask=mask[2 * i:2 * i + 2],
                          labels=labels[2 * i:2 * i + 2], n_examples_list=[2]), hw.dev) for i in range(2)]
    opt = optim.FusedAdamW(model.rt.bank, lr=1e-3, betas=(0.9, 0.98), weight_decay=0.0, cnn_lr=1e-3, max_grad_norm=-1.0)
    bank = model.rt.bank
    # reference gradient: sum of the two micro-batch gradients
    bank.zero_grad()
    for h in halves:
        out = model(dict(h))
        out["loss"].mean().backward()
    g_sum = bank.grad.clone()
    w0 = bank.master.clone()
    acfg = SimpleNamespace(gradient_accumulation_steps=2, **base)
    tasks.train_step(model, opt, dict(halves[0]), acfg, global_step=0, micro_step=0)
    torch.testing.assert_close(bank.master, w0, rtol=0, atol=0)                      # no update after the first micro-step
    tasks.train_step(model, opt, dict(halves[1]), acfg, global_step=0, micro_step=1)
    torch.testing.assert_close(bank.grad, g_sum, rtol=1e-5, atol=1e-8)
    assert (bank.master - w0).abs().max() > 0


@pytest.mark.parametrize("pool", [pytest.param("mean", marks=pytest.mark.gpu), "lse"])
def test_folded_clips_equal_clip_loop(hw, pool):
    """tasks.forward_clips_stack(fold=True) -- all clips in ONE CNN batch and ONE encoder batch -- gives the logits, the
    loss and the parameter gradients of the reference's clip loop (run_video_retrieval.py:391-419)."""
    cfg, sd, model = build("retrieval", RET, torch.float32, hw.dev)
    model.eval()
    vis = _frames(2, 4, 41)                                   # 2 videos x (2 clips x 2 frames)
    ids, mask = S.synthetic_text(3, 6, 41, cfg["vocab_size"])
    ids = ids.clamp(max=cfg["vocab_size"] - 1)
    labels = torch.tensor([1, 0, 1])
    batch = to_dev(dict(visual_inputs=vis, text_input_ids=ids, text_input_mask=mask, labels=labels, n_examples_list=[2, 1]), hw.dev)
    bank = model.rt.bank
    res = {}
    for fold in (False, True):
        bank.zero_grad()
        stack = tasks.forward_clips_stack(model, dict(batch), 2, 2, fold=fold)
        assert tuple(stack.shape) == (2, 3, 2)
        loss = tasks.training_loss(model, stack, batch["labels"], batch["n_examples_list"], pool)
        loss.backward()
        res[fold] = (stack.detach().float().cpu(), float(loss), bank.grad.clone().cpu())
    torch.testing.assert_close(res[True][0], res[False][0], rtol=1e-4, atol=1e-5)
    assert res[True][1] == pytest.approx(res[False][1], rel=1e-5, abs=1e-6)
    g0, g1 = res[False][2], res[True][2]
    assert float((g1 - g0).norm() / g0.norm()) < 2e-4
    # a text batch that does not match n_examples_list is refused (it would index the grid out of range)
    bad = dict(batch, n_examples_list=[2, 2])
    with pytest.raises(ValueError):
        tasks.forward_clips_stack(model, bad, 2, 2, fold=True)


def test_each_forward_draws_its_own_dropout_masks(hw):
    """nn.Dropout semantics: two training forwards of the same input differ (the per-forward counter is part of every
    seed), while the backward of each forward regenerates exactly its own masks (gradient check by finite differences is
    out of scope here: the mask agreement of the kernels is tested in test_kernels_misc)."""
    cfg, sd, model = build("retrieval", RET, torch.float32, hw.dev)
    model.train()
    vis = _frames(1, 2, 43)
    ids, mask = S.synthetic_text(2, 6, 43, cfg["vocab_size"])
    ids = ids.clamp(max=cfg["vocab_size"] - 1)
    b = to_dev(dict(visual_inputs=vis, text_input_ids=ids, text_input_mask=mask, labels=torch.tensor([1, 0]), n_examples_list=[2]), hw.dev)
    with torch.no_grad():
        l0 = model(dict(b))["logits"].float().cpu()
        l1 = model(dict(b))["logits"].float().cpu()
    assert (l0 - l1).abs().max() > 0
    model.rt.forward_count = 0                                 # same counter -> same masks
    with torch.no_grad():
        l2 = model(dict(b))["logits"].float().cpu()
    torch.testing.assert_close(l2, l0, rtol=0, atol=0)


def test_lazy_zero_grad_first_writer_stores(hw):
    """zero_grad(lazy=True) skips the memset of the encoder weight gradients; the batched weight-gradient launches then STORE
    instead of accumulating.  Same gradients as the fully zeroed path, also when garbage sits in the span beforehand, for a
    second (accumulating) backward, and the optimizer refuses to step if no backward followed."""
    cfg, sd, model = build("retrieval", RET, torch.float32, hw.dev)
    model.eval()
    bank = model.rt.bank
    assert bank.lazy_span is not None and bank.lazy_span[1] - bank.lazy_span[0] > 0.5 * bank.n_train * 0   # the encoder weights tile one range
    opt = optim.FusedAdamW(bank, lr=1e-3, betas=(0.9, 0.98), weight_decay=0.0, cnn_lr=1e-3, max_grad_norm=-1.0)
    vis = _frames(2, 2, 61)
    ids, mask = S.synthetic_text(4, 6, 61, cfg["vocab_size"])
    b = to_dev(dict(visual_inputs=vis, text_input_ids=ids.clamp(max=cfg["vocab_size"] - 1), text_input_mask=mask,
                    labels=torch.tensor([1, 0, 0, 1]), n_examples_list=[2, 2]), hw.dev)

    def backward():
        model(dict(b))["loss"].mean().backward()

    opt.zero_grad()
    backward()
    ref1 = bank.grad.clone()
    backward()
    ref2 = bank.grad.clone()                                   # two accumulated backwards
    a, e = bank.lazy_span
    bank.grad[a:e].fill_(123.0)                                # stale values in the span
    opt.zero_grad(lazy=True)
    assert float(bank.grad[a:e].min()) == 123.0                # not touched by the lazy zero
    backward()
    torch.testing.assert_close(bank.grad[a:e], ref1[a:e], rtol=0, atol=0)           # stored, not accumulated onto the stale values
    torch.testing.assert_close(bank.grad, ref1, rtol=1e-4, atol=1e-7)               # (atomically accumulated gradients elsewhere: order noise)
    backward()                                                 # the second backward accumulates
    torch.testing.assert_close(bank.grad, ref2, rtol=1e-4, atol=1e-7)
    opt.zero_grad(lazy=True)
    with pytest.raises(RuntimeError):
        opt.step()                                             # no backward wrote the span
    opt.zero_grad()


def test_pipelined_optimizer_update_equals_plain_update(hw):
    """FusedAdamW.launch(groups=..., prev=True, reuse_norm=True): the CNN groups at the end of a step and the transformer groups
    deferred to the start of the next one (bench.py's software-pipelined plan) give bit-identical parameters / moments to one plain
    launch per step; a deferred launch with no step behind it is a no-op."""
    from clipbert_amd import optim
    from test_model_small import build
    res = []
    for pipelined in (False, True):
        cfg, sd, model = build("retrieval", dict(num_labels=2, loss_type="ce", margin=0.1), torch.float32, hw.dev)
        bank = model.rt.bank
        opt = optim.FusedAdamW(bank, lr=1e-2, betas=(0.9, 0.98), weight_decay=1e-3, cnn_lr=5e-3, max_grad_norm=0.5)
        g = torch.Generator().manual_seed(3)
        t_end = bank.group_range[3][1]
        for step in range(3):
            grads = torch.randn(bank.grad.numel(), generator=g).to(hw.dev)
            if pipelined:
                opt.prepare_step()
                opt.launch(groups=(0, 1, 2, 3), prev=True, reuse_norm=True)      # the previous step's transformer half (step 0: skipped)
                bank.grad.copy_(grads)
                opt.launch(groups=(4, 5, 6, 7))                                  # norm over ALL gradients, CNN half
                opt.deferred_pending = True
            else:
                bank.grad.copy_(grads)
                opt.step()
        if pipelined:
            opt.launch(groups=(0, 1, 2, 3), reuse_norm=True)                     # flush
            opt.deferred_pending = False
        res.append((bank.master.clone().cpu(), bank.exp_avg.clone().cpu(), bank.exp_avg_sq.clone().cpu()))
        assert t_end > 0 and bank.n_train > t_end
    for a, b in zip(*res):
        assert torch.equal(a, b)
