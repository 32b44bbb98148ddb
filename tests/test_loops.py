"""Rows N2-N4 of SURVEY 8f around the hot path: the training loop with validation / checkpointing (start_training), the
checkpoint module (ModelSaver, E2E_TrainingRestorer, optimizer state, detectron2 / torchvision backbone import, compute-copy
refresh after load_state_dict) and the double-buffered input loader.  Host logic over the emulator build of the kernels."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from clipbert_amd import checkpoint as C
from clipbert_amd import data as D
from clipbert_amd import modeling as M
from clipbert_amd import optim, tasks
from clipbert_amd import synthetic as S
from oracle import clipbert_oracle as O
from test_model_small import HEAD_CLS, SMALL, build

RET = dict(num_labels=2, loss_type="ce", margin=0.1)
CPU = torch.device("cpu")


def _batches(cfg, n, seed0=50, n_clips=2):
    out = []
    for i in range(n):
        frames = S.synthetic_frames(2, 2 * n_clips, 64, seed0 + i).contiguous()   # uint8, 64 x 64 px: one visual token
        ids, mask = S.synthetic_text(4, 6, seed0 + i, cfg["vocab_size"])
        out.append(dict(visual_inputs=frames, text_input_ids=ids.clamp(max=cfg["vocab_size"] - 1), text_input_mask=mask,
                        labels=torch.tensor([1, 0, 1, 0]), n_examples_list=[2, 2], caption_ids=[0, 1, 2, 3]))
    return out


def test_start_training_validates_saves_and_resumes(hw, tmp_path):
    """row N2 + N3 on both backends (host emulator / MI355X): loop, validation, model_step_N.pt, restore.pt, resume"""
    CPU = hw.dev                                             # (name kept: every tensor below lives on the backend's device)
    cfg, sd, model = build("retrieval", RET, torch.float32, CPU)
    opt = optim.FusedAdamW(model.rt.bank, lr=1e-3, betas=(0.9, 0.98), weight_decay=1e-3, cnn_lr=1e-3, max_grad_norm=5.0)
    tcfg = SimpleNamespace(train_n_clips=2, num_frm=2, score_agg_func="lse", gradient_accumulation_steps=2, learning_rate=1e-3,
                           cnn_learning_rate=1e-3, decay="linear", cnn_lr_decay="linear", num_train_steps=3, warmup_ratio=0.0, valid_steps=2,
                           train_batch_size=2, max_n_example_per_group=1, output_dir=str(tmp_path), save_steps_ratio=0.34)
    loader = D.PrefetchLoader(_batches(cfg, 3), device=CPU)          # uint8 frames straight into the model (ImageNorm in the stem pack)
    saver = C.ModelSaver(os.path.join(str(tmp_path), "ckpt"))
    restorer = C.E2E_TrainingRestorer(tcfg, model, opt)
    seen = []

    def validate_fn(m, step):
        assert not m.training or True
        seen.append(step)
        return {"valid/dummy": float(step)}

    w0 = model.rt.bank.master.clone()
    end = tasks.start_training(model, opt, loader, tcfg, validate_fn=validate_fn, model_saver=saver, restorer=restorer, total_n_examples=6)
    assert end == 3 and seen == [2, 3]                               # every valid_steps and once at the end (:499-516)
    assert opt.step_count == 3 and (model.rt.bank.master - w0).abs().max() > 0
    assert os.path.exists(os.path.join(str(tmp_path), "ckpt", "model_step_2.pt")) and os.path.exists(os.path.join(str(tmp_path), "ckpt", "model_step_3.pt"))
    assert os.path.exists(os.path.join(str(tmp_path), "restore.pt"))
    # model_step_N.pt carries the reference's key layout in logical shapes and loads into a fresh model
    saved = torch.load(os.path.join(str(tmp_path), "ckpt", "model_step_3.pt"))
    assert set(saved) == set(sd) and saved["cnn.feature.backbone.res5.0.conv2.weight"].is_contiguous()
    # resume: a new model + optimizer pick up step, weights and AdamW moments from restore.pt
    cfg2, _sd2, model2 = build("retrieval", RET, torch.float32, CPU, seed=6)
    opt2 = optim.FusedAdamW(model2.rt.bank, lr=1e-3, betas=(0.9, 0.98), weight_decay=1e-3, cnn_lr=1e-3, max_grad_norm=5.0)
    r2 = C.E2E_TrainingRestorer(tcfg, model2, opt2)
    assert r2.global_step == 3 and opt2.step_count == 3
    torch.testing.assert_close(model2.rt.bank.master, model.rt.bank.master, rtol=0, atol=0)
    torch.testing.assert_close(model2.rt.bank.exp_avg_sq, model.rt.bank.exp_avg_sq, rtol=0, atol=0)
    # the dropout counters travel with restore.pt: the resumed run continues the mask sequence
    assert model2.rt.forward_count == model.rt.forward_count > 0
    # the checkpoint the loop wrote, loaded into the ORACLE, gives the logits the trained model computes (eval mode)
    model.eval()
    b = _batches(cfg, 1, seed0=90, n_clips=1)[0]
    vis = b["visual_inputs"]
    ob = dict(visual_inputs=O.image_norm(vis, S.PIXEL_MEAN, S.PIXEL_STD), text_input_ids=b["text_input_ids"], text_input_mask=b["text_input_mask"],
              n_examples_list=[2, 2])
    with torch.no_grad():
        ref = O.clipbert_forward({k: v.float() if v.is_floating_point() else v for k, v in saved.items()}, ob, cfg, "retrieval")["logits"]
        got = model(dict(visual_inputs=vis.to(CPU), text_input_ids=b["text_input_ids"].to(CPU), text_input_mask=b["text_input_mask"].to(CPU),
                         n_examples_list=[2, 2]))["logits"].float().cpu()
    torch.testing.assert_close(got, ref, rtol=2e-3, atol=2e-3)


def test_load_state_dict_refreshes_compute_copies(hw):
    """loading weights into a PREPARED bf16 model must refresh the bf16 copies the kernels read."""
    CPU = hw.dev
    cfg, sd, model = build("retrieval", RET, torch.bfloat16, CPU)
    frames = S.synthetic_frames(1, 2, 64, 3)[..., :64, :].repeat(1, 1, 1, 1, 2).contiguous()
    ids, mask = S.synthetic_text(2, 6, 3, cfg["vocab_size"])
    b = dict(visual_inputs=frames.to(CPU), text_input_ids=ids.clamp(max=cfg["vocab_size"] - 1).to(CPU), text_input_mask=mask.to(CPU), n_examples_list=[2])
    with torch.no_grad():
        before = model(dict(b))["logits"].float().clone()
    sd2 = S.full_state_dict(cfg, "retrieval", 9)
    assert M.load_state_dict_with_mismatch(model, sd2) == len(sd2)                  # whole-model load
    with torch.no_grad():
        after = model(dict(b))["logits"].float().clone()
    assert (after - before).abs().max() > 1e-3
    fresh = M.ClipBert(cfg, detectron2_model_cfg="R-50-grid.yaml", transformer_cls=HEAD_CLS["retrieval"])
    fresh.load_state_dict(sd2, strict=True)
    fresh.eval().prepare(dtype=torch.bfloat16, device=CPU)
    with torch.no_grad():
        ref = fresh(dict(b))["logits"].float()
    torch.testing.assert_close(after, ref, rtol=0, atol=0)
    # sub-module load (load_separate_ckpt's path) refreshes too
    tsd = {k[len("transformer."):]: v for k, v in sd.items() if k.startswith("transformer.")}
    assert M.load_state_dict_with_mismatch(model.transformer, tsd) == len(tsd)
    bank = model.rt.bank
    torch.testing.assert_close(bank.w16[:bank.n_train].float(), bank.master[:bank.n_train].bfloat16().float())


def test_backbone_import_from_torchvision_and_detectron2_layouts(hw, tmp_path):
    CPU = hw.dev
    cfg, sd, model = build("retrieval", RET, torch.float32, CPU)
    own = model.cnn.feature.state_dict()
    # a torchvision-style ResNet-50 state dict carrying recognisable values
    inv = [(b, a) for a, b in C.TORCHVISION_TO_DETECTRON2]
    tv = {}
    for k, v in own.items():
        name = k[len("backbone."):]
        if name.startswith("stem."):
            name = name[len("stem."):]
        name = name.replace("shortcut.norm", "downsample.1").replace("shortcut", "downsample.0")
        for d2, t in (("conv1.norm", "bn1"), ("conv2.norm", "bn2"), ("conv3.norm", "bn3"), ("res2", "layer1"), ("res3", "layer2"),
                      ("res4", "layer3"), ("res5", "layer4")):
            name = name.replace(d2, t)
        tv[name] = torch.full_like(v, 0.25) if v.is_floating_point() else v
    tv["fc.weight"] = torch.zeros(1000, 2048)
    tv["bn1.num_batches_tracked"] = torch.tensor(0)
    conv = C.convert_torchvision_to_detectron2(tv)
    assert "stem.conv1.weight" in conv and "res2.0.shortcut.norm.running_var" in conv and "res5.2.conv3.norm.weight" in conv
    assert not any(k.startswith("fc.") for k in conv)
    path = os.path.join(str(tmp_path), "resnet50.pth")
    torch.save(tv, path)
    n = C.load_detectron2_backbone(model.cnn, path)
    assert n == len(own)
    assert all(float(v.float().mean()) == 0.25 for v in model.cnn.feature.state_dict().values())
    # detectron2 checkpoint layout: {"model": {"backbone.*", dead "proposal_generator.*" / "roi_heads.*"}}, numpy arrays (.pkl style)
    d2 = {"model": {**{k: np.full(tuple(v.shape), 0.5, dtype=np.float32) for k, v in own.items()},
                    "proposal_generator.rpn_head.conv.weight": np.zeros((4, 4, 3, 3), np.float32), "pixel_mean": np.zeros(3, np.float32)}}
    assert C.load_detectron2_backbone(model.cnn, d2) == len(own)
    assert all(float(v.float().mean()) == 0.5 for v in model.cnn.feature.state_dict().values())
    # a backbone with recognisable RANDOM weights, imported after prepare(): the kernels must compute with it (oracle on the same state)
    sd9 = S.full_state_dict(cfg, "retrieval", 9)
    d2r = {"model": {k[len("cnn.feature."):]: v.numpy() for k, v in sd9.items() if k.startswith("cnn.feature.backbone.")}}
    assert C.load_detectron2_backbone(model.cnn, d2r) == len(own)
    frames = S.synthetic_frames(1, 2, 64, 3)
    ids, mask = S.synthetic_text(2, 6, 3, cfg["vocab_size"])
    ids = ids.clamp(max=cfg["vocab_size"] - 1)
    now = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    with torch.no_grad():
        ref = O.clipbert_forward(now, dict(visual_inputs=O.image_norm(frames, S.PIXEL_MEAN, S.PIXEL_STD), text_input_ids=ids, text_input_mask=mask,
                                           n_examples_list=[2]), cfg, "retrieval")["logits"]
        got = model(dict(visual_inputs=frames.to(CPU), text_input_ids=ids.to(CPU), text_input_mask=mask.to(CPU), n_examples_list=[2]))["logits"].float().cpu()
    torch.testing.assert_close(got, ref, rtol=2e-3, atol=2e-3)
    assert (now["cnn.feature.backbone.res4.0.conv2.weight"] - sd9["cnn.feature.backbone.res4.0.conv2.weight"]).abs().max() == 0
    # load_separate_ckpt raises when nothing matches
    torch.save({"unrelated.weight": torch.zeros(3)}, os.path.join(str(tmp_path), "bad.pth"))
    with pytest.raises(RuntimeError):
        model.load_separate_ckpt(cnn_weights_path=os.path.join(str(tmp_path), "bad.pth"))
    # freezing the backbone after an optimizer was built on the bank is refused (it would orphan the optimizer's views)
    optim.FusedAdamW(model.rt.bank, lr=1e-3)
    with pytest.raises(RuntimeError):
        model.freeze_cnn_backbone()


def test_prefetch_loader_and_infinite_iterator():
    batches = [dict(visual_inputs=torch.full((1, 2, 3, 4, 4), i, dtype=torch.uint8), ids=torch.arange(3) + i, meta=[i, "x"]) for i in range(5)]
    got = list(D.PrefetchLoader(batches, device=CPU))
    assert len(got) == 5 and all(int(g["visual_inputs"][0, 0, 0, 0, 0]) == i and g["meta"] == [i, "x"] for i, g in enumerate(got))
    assert got[2]["visual_inputs"].dtype == torch.uint8                  # no .float(): ImageNorm happens inside the stem pack
    it = iter(D.InfiniteIterator(batches))
    assert [int(next(it)["ids"][0]) for _ in range(12)] == [0, 1, 2, 3, 4, 0, 1, 2, 3, 4, 0, 1]


@pytest.mark.gpu
def test_prefetch_loader_pinned_double_buffer_gpu():
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(0)
    batches = [dict(visual_inputs=torch.randint(0, 256, (2, 4, 3, 64, 64), dtype=torch.uint8, generator=gen),
                    text_input_ids=torch.randint(0, 1000, (4, 8), generator=gen)) for _ in range(7)]
    loader = D.PrefetchLoader(batches, device=dev)
    acc = []
    for b in loader:
        assert b["visual_inputs"].is_cuda and b["visual_inputs"].dtype == torch.uint8
        acc.append((b["visual_inputs"].long().sum() + b["text_input_ids"].sum()).clone())      # consume on the compute stream
    torch.cuda.synchronize()
    ref = [int(b["visual_inputs"].long().sum() + b["text_input_ids"].sum()) for b in batches]
    assert [int(a) for a in acc] == ref
    assert len({k for k in loader._pinned}) == 4                        # two tensor keys x two pinned slots, reused across batches


def test_backbone_load_after_prepare_refreshes_compute_copies(hw):
    """load_detectron2_backbone goes through cnn.feature.load_state_dict -- the bf16 compute copies, folded FrozenBN
    vectors and the packed stem filter of a PREPARED model must follow the new masters."""
    CPU = hw.dev
    cfg, sd, model = build("retrieval", RET, torch.bfloat16, CPU)
    bank = model.rt.bank
    own = model.cnn.feature.state_dict()
    d2 = {"model": {k: np.full(tuple(v.shape), 0.5, dtype=np.float32) for k, v in own.items()}}
    assert C.load_detectron2_backbone(model.cnn, d2) == len(own)
    p = model.cnn.feature.backbone.res4[0].conv2.weight
    assert float(p.float().mean()) == 0.5
    assert bank.is_trainable(p)
    off = bank.offset[id(p)]
    torch.testing.assert_close(bank.w16[off:off + p.numel()].float().cpu(), torch.full((p.numel(),), 0.5))      # compute copy == master
    assert model.rt.stem_w is None and all(m._ss is None for m in model.modules() if isinstance(m, M.Conv2d))


def test_freeze_after_prepare_keeps_the_group_layout(emul):
    """re-preparing inside freeze_cnn_backbone() reuses the arguments of the first prepare()."""
    cfg = dict(SMALL, **RET)
    model = M.ClipBert(cfg, detectron2_model_cfg="R-50-grid.yaml", transformer_cls=HEAD_CLS["retrieval"])
    model.prepare(dtype=torch.float32, device=CPU, transformer_lr_mul_prefix="classifier", cnn_lr_mul_prefix="grid_encoder")
    g0 = model.rt.bank.group_range[0]
    assert g0[1] > g0[0]                                  # the 'new transformer' group holds the classifier
    model.freeze_cnn_backbone()
    assert model.rt.prepare_args["transformer_lr_mul_prefix"] == "classifier"
    g0b = model.rt.bank.group_range[0]
    assert g0b[1] - g0b[0] == g0[1] - g0[0]
