"""Round 6: the squared gradient norm of a single-process step comes from the shares the weight-gradient launches leave behind
(cb_gemm_desc.sq_slots -> ParamBank.fold_* -> cb_sq_sum_fold) instead of a second pass over the gradients, and the ResNet's convolution
weight gradients are STORED by their first writer (accumulate = 2) instead of accumulated into a zero-filled range.  Same step, both
ways, on the host emulator and the GPU: the same norm up to the order of fp32 additions, the same update; a second backward of the step
(gradient accumulation) voids the shares and the full pass runs; a gradient exchange blocks them."""
from types import SimpleNamespace

import pytest
import torch

from clipbert_amd import optim, tasks
from clipbert_amd import synthetic as S
from oracle import clipbert_oracle as O
from test_model_small import build, to_dev

RET = dict(num_labels=2, loss_type="ce", margin=0.1)
TCFG = SimpleNamespace(train_n_clips=2, num_frm=2, score_agg_func="lse", learning_rate=1e-3, cnn_learning_rate=1e-3, decay="linear",
                       cnn_lr_decay="linear", num_train_steps=10, warmup_ratio=0.1, gradient_accumulation_steps=1)


def _batch(dev, cfg, seed=5):
    f = S.synthetic_frames(2, 4, 64, seed)
    vis = O.image_norm(f, S.PIXEL_MEAN, S.PIXEL_STD)
    ids, mask = S.synthetic_text(4, 6, seed, cfg["vocab_size"])
    return to_dev(dict(visual_inputs=vis, text_input_ids=ids.clamp(max=cfg["vocab_size"] - 1), text_input_mask=mask, labels=torch.tensor([1, 0, 1, 0]),
                       n_examples_list=[2, 2]), dev)


def _one_step(hw, fold, acc_steps=1, block=False):
    cfg, sd, model = build("retrieval", RET, torch.bfloat16, hw.dev)
    model.eval()                                                   # (dropout off: both arms see the same forward)
    bank = model.rt.bank
    opt = optim.FusedAdamW(bank, lr=1e-3, betas=(0.9, 0.98), weight_decay=1e-3, cnn_lr=1e-3, max_grad_norm=0.05, fold_norm=fold)
    if block:
        bank.norm_fold_blocked = True
    tcfg = SimpleNamespace(**dict(vars(TCFG), gradient_accumulation_steps=acc_steps))
    used = []
    real = bank.fold_result
    bank.fold_result = lambda: used.append(real()) or used[-1]
    for micro in range(acc_steps):
        tasks.train_step(model, opt, dict(_batch(hw.dev, cfg)), tcfg, global_step=0, micro_step=micro)
    norm2 = float(opt._sq.float().cpu())
    return norm2, bank.grad[:bank.n_train].clone().cpu(), bank.master[:bank.n_train].clone().cpu(), used, bank


def test_folded_norm_and_first_writer_stores_equal_the_full_pass(hw):
    n_fold, g_fold, p_fold, used, bank = _one_step(hw, True)
    n_full, g_full, p_full, used0, _ = _one_step(hw, False)
    assert used and used[-1] is not None and not used0             # the shares were used / never consulted
    segs, slots = used[-1]
    covered = bank.n_train - sum(hi - lo for lo, hi in segs)
    assert covered > 0.5 * bank.n_train                            # encoder + ResNet weights: most of the gradient never re-read
    assert covered == (bank.lazy_span[1] - bank.lazy_span[0]) + (bank.fresh_span[1] - bank.fresh_span[0])
    assert torch.isfinite(g_fold).all()
    for a, b in (bank.lazy_span, bank.fresh_span):               # first-writer stores: the same weight gradients bit for bit (ordered slab sums)
        torch.testing.assert_close(g_fold[a:b], g_full[a:b], rtol=0, atol=0)
    torch.testing.assert_close(g_fold, g_full, rtol=1e-5, atol=1e-9)      # (embedding scatters add through atomics: order of arrival)
    assert abs(n_fold - n_full) <= 2e-6 * n_full, (n_fold, n_full)
    assert n_full > 0.05 ** 2                                      # (the clip is active: the norm matters)
    torch.testing.assert_close(p_fold, p_full, rtol=1e-5, atol=1e-7)
    want = float((g_full.double() ** 2).sum())
    assert abs(n_fold - want) <= 1e-5 * want


def test_gradient_accumulation_and_exchanges_fall_back_to_the_full_pass(hw):
    n2, g2, _p, used, _ = _one_step(hw, True, acc_steps=2)
    assert used and used[-1] is None                               # a second backward accumulated: the first one's shares are void
    want = float((g2.double() ** 2).sum())
    assert abs(n2 - want) <= 1e-5 * want
    n1, g1, _p, used, _ = _one_step(hw, True, block=True)          # (what GradSync sets when there is something to exchange)
    assert used and used[-1] is None
    want = float((g1.double() ** 2).sum())
    assert abs(n1 - want) <= 1e-5 * want
