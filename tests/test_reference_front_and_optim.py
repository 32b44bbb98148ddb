"""Live pins of the oracle / the product's host logic against reference code EXECUTED FROM ITS SOURCE (only where
/root/reference exists).  What the goldens do not cover:

  a21 (update half)  the reference's AdamW class, LR schedules and 8-group parameter split  vs  oracle.adamw_step / warmup_linear_lr /
                     param_groups  and  clipbert_amd.optim.get_lr_sched / params.group_of / ParamBank's group ranges
  a5                 the grid encoder Sequential as GridFeatBackbone.__init__ builds it (grid_feat.py:16-21,43-48)  vs  oracle.grid_encoder
  a6                 repeat_tensor_rows (data_utils.py:344-357)  vs  oracle.repeat_rows
  a1                 ImageNorm.__call__ (data_utils.py:256-276, .cuda() stripped by an AST edit)  vs  oracle.image_norm
"""
import math
import warnings

import pytest
import torch

from clipbert_amd import optim as P_optim
from clipbert_amd import params as P_params
from oracle import clipbert_oracle as O
from oracle import ref_functions as RF
from oracle import ref_shim

pytestmark = pytest.mark.skipif(not ref_shim.available(), reason="reference tree not present")


# ---- a21: AdamW --------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("weight_decay", [0.0, 1e-3, 0.1])
def test_adamw_three_steps_equal_the_reference_class(weight_decay):
    AdamW = RF.adamw_class()
    g = torch.Generator().manual_seed(11)
    shapes = [(7, 5), (33,), (4, 3, 3, 3)]
    lr, betas, eps = 3e-3, (0.9, 0.98), 1e-6
    params = [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in shapes]
    opt = AdamW(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
    mine = [(p.detach().clone(), torch.zeros_like(p), torch.zeros_like(p)) for p in params]
    for step in range(1, 4):
        grads = [torch.randn(*s, generator=g) * (10.0 if step == 2 else 1.0) for s in shapes]
        for p, gr in zip(params, grads):
            p.grad = gr.clone()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")              # deprecated add_(Number, Tensor) overloads of the reference
            opt.step()
        mine = [O.adamw_step(p, gr, m, v, step, lr, betas[0], betas[1], eps, weight_decay) for (p, m, v), gr in zip(mine, grads)]
        for p, (q, m, v) in zip(params, mine):
            torch.testing.assert_close(q, p.detach(), rtol=1e-6, atol=1e-7)
            torch.testing.assert_close(m, opt.state[p]["exp_avg"], rtol=1e-6, atol=1e-8)
            torch.testing.assert_close(v, opt.state[p]["exp_avg_sq"], rtol=1e-6, atol=1e-10)


# ---- a21: LR schedules -------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("total,ratio", [(1000, 0.1), (37, 0.1), (100000, 0.05)])
def test_lr_schedules_equal_the_reference_functions(total, ratio):
    ref = RF.sched_module()
    ws = int(ratio * total)
    steps = sorted({0, 1, 2, max(ws - 1, 0), ws, ws + 1, total // 2, total - 1, total, total + 5} | set(range(0, total + 1, max(1, total // 23))))
    base = 5e-5
    for s in steps:
        if ws > 0:
            assert P_optim.get_lr_sched(s, "linear", base, total, ratio) == ref.get_lr_sched(s, "linear", base, total, ratio), s
            assert O.warmup_linear_lr(s, base, total, ratio) == ref.get_lr_sched(s, "linear", base, total, ratio), s
            assert P_optim.warmup_linear(s, ws, total) == ref.warmup_linear(s, ws, total), s
            if s > 0:                                     # (the reference's noam_schedule divides by step ** 0.5: step 0 is 0 / warm-up only)
                assert P_optim.get_lr_sched(s, "invsqrt", base, total, ratio) == ref.get_lr_sched(s, "invsqrt", base, total, ratio), s
            else:
                assert P_optim.get_lr_sched(0, "invsqrt", base, total, ratio) == ref.get_lr_sched(0, "invsqrt", base, total, ratio)
        assert P_optim.get_lr_sched(s, "constant", base, total, ratio) == ref.get_lr_sched(s, "constant", base, total, ratio)
    for epoch in range(0, 12):
        for ms in ([], [3], [2, 5, 9], [9, 2, 5]):
            assert P_optim.multi_step_schedule(epoch, ms) == ref.multi_step_schedule(epoch, ms)
            assert (P_optim.get_lr_sched(0, "multi_step", base, total, ratio, decay_epochs=ms, multi_step_epoch=epoch)
                    == ref.get_lr_sched(0, "multi_step", base, total, ratio, decay_epochs=ms, multi_step_epoch=epoch))


# ---- a21: the 8 parameter groups on the REAL parameter names ------------------------------------------------------------------
def _real_named_parameters():
    from clipbert_amd import modeling as M
    cfg = dict(O.BASE_CONFIG, num_hidden_layers=2, num_labels=2, loss_type="ce", margin=0.1, vocab_size=400, max_position_embeddings=64)
    model = M.ClipBert(cfg, detectron2_model_cfg="R-50-grid.yaml", transformer_cls=M.ClipBertForVideoTextRetrieval)
    return model, [(n, p) for n, p in model.named_parameters() if p.requires_grad]


@pytest.mark.parametrize("t_prefix,c_prefix", [("", "grid_encoder"), ("classifier", "grid_encoder"), ("", ""), ("bert.encoder.layer.1", "res5")])
def test_parameter_groups_equal_the_reference_builder(t_prefix, c_prefix):
    build = RF.group_builder()
    model, named = _real_named_parameters()
    t_named = [(n, p) for n, p in named if "transformer" in n]            # setup_e2e_optimizer, utils.py:99-104
    c_named = [(n, p) for n, p in named if "cnn" in n]
    assert len(t_named) + len(c_named) == len(named)
    groups = build(t_named, 5e-5, 1e-3, lr_mul=10, lr_mul_prefix=t_prefix) + build(c_named, 1e-4, 1e-2, lr_mul=3, lr_mul_prefix=c_prefix)
    assert len(groups) == 8                                                 # run_video_retrieval.py:455 asserts 8 groups
    ref_group = {}
    for gi, grp in enumerate(groups):
        for p in grp["params"]:
            ref_group[id(p)] = gi
    mine = O.param_groups([n for n, _ in named], t_prefix, c_prefix)
    for n, p in named:
        assert mine[n] == ref_group[id(p)], (n, mine[n], ref_group[id(p)])
        assert P_params.group_of(n, t_prefix, c_prefix) == ref_group[id(p)], n
    # weight decay / lr multipliers sit where the product's optimizer puts them: groups 0, 2, 4, 6 decay; 0, 1, 4, 5 carry the multiplier
    assert [g["weight_decay"] > 0 for g in groups] == [True, False, True, False, True, False, True, False]
    assert ["lr" in g for g in groups] == [True, True, False, False, True, True, False, False]
    # ParamBank lays the flat buffers out in exactly that group order
    bank = P_params.ParamBank(model, "cpu", torch.float32, t_prefix, c_prefix)
    for n, p in named:
        lo, hi = bank.group_range[ref_group[id(p)]]
        assert lo <= bank.offset[id(p)] and bank.offset[id(p)] + p.numel() <= hi, n
    for gi, grp in enumerate(groups):
        lo, hi = bank.group_range[gi]
        assert sum(p.numel() for p in grp["params"]) <= hi - lo


# ---- a5: grid encoder ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("hw", [7, 14, 24, (8, 12)])
def test_grid_encoder_equals_the_reference_sequential(hw):
    h, w = (hw, hw) if isinstance(hw, int) else hw
    cin, hidden = 64, 48                                     # (the arithmetic does not depend on the channel counts; 2048 -> 768 below)
    enc = RF.grid_encoder(cin, hidden).eval()
    assert [type(m).__name__ for m in enc] == ["Conv2d", "MaxPool2d", "ReLU"]
    assert enc[0].bias is None and enc[0].padding == (1, 1) and enc[0].stride == (1, 1)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, cin, h, w, generator=g)
    sd = {"cnn.grid_encoder.0.weight": enc[0].weight.detach().clone()}
    with torch.no_grad():
        ref = enc(x.clone())
        mine = O.grid_encoder(sd, x, "cnn.")
    assert mine.shape == ref.shape == (3, hidden, h // 2, w // 2)
    torch.testing.assert_close(mine, ref, rtol=0, atol=0)


def test_grid_encoder_full_width_and_gradients():
    enc = RF.grid_encoder(2048, 768)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 2048, 7, 7, generator=g, requires_grad=True)
    x2 = x.detach().clone().requires_grad_(True)
    w2 = enc[0].weight.detach().clone().requires_grad_(True)
    ref = enc(x)
    mine = O.grid_encoder({"cnn.grid_encoder.0.weight": w2}, x2, "cnn.")
    torch.testing.assert_close(mine, ref, rtol=0, atol=0)
    go = torch.randn(ref.shape, generator=g)
    ref.backward(go)
    mine.backward(go)
    torch.testing.assert_close(x2.grad, x.grad, rtol=0, atol=0)
    torch.testing.assert_close(w2.grad, enc[0].weight.grad, rtol=0, atol=0)


# ---- a6: repeat_tensor_rows ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("counts", [[1, 1, 1], [2, 2, 2], [3, 1, 2], [5], [1, 4, 1, 2]])
def test_repeat_rows_equals_the_reference_function(counts):
    fn = RF.repeat_tensor_rows()
    x = torch.arange(len(counts) * 6, dtype=torch.float32).view(len(counts), 2, 3)
    ref = fn(x, counts)
    mine = O.repeat_rows(x, counts)
    assert torch.equal(mine, ref)


# ---- a1: ImageNorm -----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mean,std", [((123.675, 116.28, 103.53), (1.0, 1.0, 1.0)), ((0.485, 0.456, 0.406), (0.229, 0.224, 0.225)),
                                      ((123.675, 116.28, 103.53), (58.395, 57.12, 57.375))])
def test_image_norm_equals_the_reference_class(mean, std):
    ImageNorm = RF.image_norm_class()
    norm = ImageNorm(mean=mean, std=std)
    g = torch.Generator().manual_seed(9)
    u8 = torch.randint(0, 256, (2, 3, 3, 10, 12), generator=g, dtype=torch.uint8)
    ref = norm(u8.float())                                   # (dataloader.py:104 hands float frames; the call works in place)
    mine = O.image_norm(u8, mean, std)
    torch.testing.assert_close(mine, ref, rtol=1e-6, atol=1e-6)
    # the /255 branch only fires for a [0, 1] mean (data_utils.py:274-275)
    m, s = torch.tensor(mean).view(1, 1, 3, 1, 1), torch.tensor(std).view(1, 1, 3, 1, 1)
    plain = (u8.float() / (255.0 if max(mean) <= 1 else 1.0) - m) / s
    torch.testing.assert_close(ref, plain, rtol=1e-6, atol=1e-6)
