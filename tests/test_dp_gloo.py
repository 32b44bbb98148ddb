"""Data-parallel path on CPU: 2 processes, gloo backend, the product DP code (clipbert_amd.dist.GradSync +
FusedAdamW) over the host-emulator build of the kernels.  DP=2 on two half batches must produce the same
updated weights as DP=1 on the full batch (gradient averaging == global-batch mean loss)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup_emul():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "emul"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes
    import build_emul
    from clipbert_amd import _lib, ops
    _lib._LIB = _lib.bind(ctypes.CDLL(build_emul.build()))
    ops._ALLOW_HOST_POINTERS = True


def _train_one_step(rank, world, out_path, compress=None):
    import test_model_small as T
    from clipbert_amd.dist import GradSync
    from clipbert_amd.optim import FusedAdamW
    cfg, sd, model = T.build("retrieval", dict(num_labels=2, loss_type="ce", margin=0.1), torch.float32, torch.device("cpu"))
    full = T.make_batch(cfg, "retrieval", 2, 2, 6)
    full["labels"] = torch.tensor([1, 0, 0, 1])
    if world == 1:
        batch = full
    else:                                   # DistributedSampler-style shard: one video (+ its 2 texts) per rank
        batch = dict(visual_inputs=full["visual_inputs"][rank:rank + 1].contiguous(),
                     text_input_ids=full["text_input_ids"][2 * rank:2 * rank + 2].contiguous(),
                     text_input_mask=full["text_input_mask"][2 * rank:2 * rank + 2].contiguous(),
                     n_examples_list=[2], labels=full["labels"][2 * rank:2 * rank + 2])
    bank = model.rt.bank
    shard = compress in ("shard", "shard16")
    sync = GradSync(bank, compress="bf16" if compress in ("bf16direct", "shard16") else (None if shard else compress), shard=shard,
                    bucket_bytes=(1 << 16) if shard else (64 << 20))
    sync.broadcast_parameters(0)
    calls, calls5 = [], []
    # attach(): transformer buckets from the end of the encoder backward; the two ends of the CNN range (grid_encoder, res5) from
    # inside the ResNet backward, the middle after it.  (wrapped here only to count the calls)
    sync.attach(model)
    assert len(sync.c_early) == 2 and sync._cnn_late()
    h_enc, h_r5 = model.rt.after_encoder_backward, model.rt.after_res5_backward
    model.rt.after_encoder_backward = lambda: (calls.append(1), h_enc())
    model.rt.after_res5_backward = lambda: (calls5.append(1), h_r5())
    opt = FusedAdamW(bank, lr=1e-3, betas=(0.9, 0.98), weight_decay=1e-3, max_grad_norm=5.0)
    opt.zero_grad()
    out = model(batch)
    out["loss"].mean().backward()
    assert calls5 == [1]
    sync.reduce_cnn()
    g16 = sync.wire_gradients() if compress in ("bf16direct", "shard16") else None     # the optimizer reads the reduced bf16 image itself
    sync.wait(cast_back=g16 is None and not shard)
    if shard:
        # owner-only update: each rank holds the reduced gradients of 1/world of every bucket, updates those slices, and the
        # all-gather distributes the new weights (masters too here: rank 0 saves them)
        pieces = sync.owned_pieces()
        assert sum(hi - lo for lo, hi in pieces) * world == bank.n_train and len(pieces) > 2
        before = bank.master.clone()
        opt.step(grad_scale=sync.grad_scale, grad16=g16, pieces=pieces, norm_reduce=sync.norm_all_reduce)
        if world > 1:
            mine = torch.zeros(bank.n_train, dtype=torch.bool)
            for lo, hi in pieces:
                mine[lo:hi] = True
            assert torch.equal(bank.master[:bank.n_train][~mine], before[:bank.n_train][~mine])      # other ranks' slices untouched
        sync.gather_updated()               # what a training step does: the compute weights (+ the masters kernels read in fp32)
        try:
            opt.state_dict()
            raise AssertionError("state_dict() of a sharded state must refuse")
        except RuntimeError as e:
            assert "gather_state" in str(e)
        sync.gather_state(opt)              # before a checkpoint: masters and moments from their owners
        assert len(opt.state_dict()["state"]) > 0
    else:
        opt.step(grad_scale=sync.grad_scale, grad16=g16)
    assert calls == [1]                     # transformer bucket was launched from inside the backward
    if rank == 0:
        torch.save(dict(master=bank.master.clone(), norm=opt.grad_norm(), exp_avg=bank.exp_avg.clone(), exp_avg_sq=bank.exp_avg_sq.clone()), out_path)


def _train_multi_clip_accumulated(rank, world, out_path, compress=None):
    """tasks.train_step with an UN-FOLDED clip loop (two encoder backwards per micro-step) and gradient accumulation over
    two micro-steps, the overlap hook armed as INTEGRATION.md shows: the transformer buckets must be exchanged exactly once,
    after the last encoder backward of the last micro-step (they used to go out after the first clip)."""
    from types import SimpleNamespace
    import test_model_small as T
    from clipbert_amd import tasks
    from clipbert_amd.dist import GradSync
    from clipbert_amd.optim import FusedAdamW
    cfg, sd, model = T.build("retrieval", dict(num_labels=2, loss_type="ce", margin=0.1), torch.float32, torch.device("cpu"))
    from clipbert_amd import synthetic as S
    from oracle import clipbert_oracle as O
    frames = S.synthetic_frames(4, 4, 64, 7).contiguous()                                       # 4 videos x (2 clips x 2 frames), 64 x 64 px: one visual token
    ids, mask = S.synthetic_text(4, 6, 7, cfg["vocab_size"])                                     # 1 text each
    full = dict(visual_inputs=O.image_norm(frames, S.PIXEL_MEAN, S.PIXEL_STD), text_input_ids=ids.clamp(max=cfg["vocab_size"] - 1),
                text_input_mask=mask, labels=torch.tensor([1, 0, 0, 1]))
    bank = model.rt.bank
    sync = GradSync(bank, compress=compress, bucket_bytes=1 << 16)       # several buckets per range
    sync.broadcast_parameters(0)
    calls, calls5 = [], []
    model.rt.after_encoder_backward = lambda: (calls.append(1), sync.reduce_transformer())
    from clipbert_amd import modeling as M
    sync.set_cnn_split(M.cnn_early_split(model))
    model.rt.after_res5_backward = lambda: (calls5.append(1), sync.reduce_cnn_early())
    opt = FusedAdamW(bank, lr=1e-3, betas=(0.9, 0.98), weight_decay=1e-3, max_grad_norm=5.0)
    tcfg = SimpleNamespace(train_n_clips=2, num_frm=2, score_agg_func="lse", gradient_accumulation_steps=2, learning_rate=1e-3,
                           cnn_learning_rate=1e-3, decay="constant", cnn_lr_decay="constant", num_train_steps=10, warmup_ratio=0.0)
    per = 2 // world                                         # videos per rank per micro-step
    for micro in range(2):
        idx = [micro * 2 + rank * per + j for j in range(per)]
        batch = dict(visual_inputs=full["visual_inputs"][idx].contiguous(), text_input_ids=full["text_input_ids"][idx].contiguous(),
                     text_input_mask=full["text_input_mask"][idx].contiguous(), n_examples_list=[1] * per, labels=full["labels"][idx])
        tasks.train_step(model, opt, batch, tcfg, global_step=0, sync=sync, micro_step=micro, fold_clips=False)
    assert calls == [1], calls               # one exchange: last encoder backward of the last micro-step
    assert calls5 == [1], calls5             # ... and the early CNN part once, from the last ResNet backward of the last micro-step
    if rank == 0:
        torch.save(dict(master=bank.master.clone(), norm=opt.grad_norm()), out_path)


def _sharded_inference(rank, world, out_path, compress=None):
    """configs[4] path: videos are independent units sharded over the ranks (tasks.shard_for_rank), every rank scores its
    videos against all captions, ONE all-gather of the (vid_id, txt_id, score) rows (run_video_retrieval.py:696-724)."""
    from types import SimpleNamespace
    import test_model_small as T
    from clipbert_amd import synthetic as S
    from clipbert_amd import tasks
    from oracle import clipbert_oracle as O
    cfg, sd, model = T.build("retrieval", dict(num_labels=2, loss_type="ce", margin=0.1), torch.float32, torch.device("cpu"))
    icfg = SimpleNamespace(inference_n_clips=2, num_frm=2, score_agg_func="lse", inference_batch_size=2)
    n_vid = 3
    ids, mask = S.synthetic_text(n_vid, 6, 9, cfg["vocab_size"])
    ids = ids.clamp(max=cfg["vocab_size"] - 1)
    videos = []
    for v in range(n_vid):
        fr = S.synthetic_frames(1, 4, 64, 100 + v).contiguous()                             # 64 x 64 px: one visual token
        videos.append(dict(vid_id=f"video{v}", visual_inputs=O.image_norm(fr, S.PIXEL_MEAN, S.PIXEL_STD), text_input_ids=ids,
                           text_input_mask=mask, caption_ids=[f"cap{j}" for j in range(n_vid)]))
    mine = [videos[i] for i in tasks.shard_for_rank(n_vid, rank, world)]
    gt = {f"cap{j}": f"video{j}" for j in range(n_vid)}
    rows, metrics = tasks.inference_retrieval(model, mine, icfg, gt_txt_id2vid_id=gt)
    assert len(rows) == n_vid * n_vid                    # every rank holds all rows after the gather
    if rank == 0:
        torch.save(dict(rows=sorted((r["vid_id"], r["txt_id"], r["score"]) for r in rows), metrics=metrics), out_path)


def _training_loop(rank, world, out_path, compress=None):
    """tasks.start_training on 2 ranks: 2 optimizer steps, validation + model_step_N.pt, restore.pt -- with the
    all-reduce exchange or (compress == "shard") the owner-only update, whose checkpoints need the state back from its owners."""
    from types import SimpleNamespace
    import test_model_small as T
    from clipbert_amd import checkpoint as C
    from clipbert_amd import data as D
    from clipbert_amd import synthetic as S
    from clipbert_amd import tasks
    from clipbert_amd.dist import GradSync
    from clipbert_amd.optim import FusedAdamW
    cpu = torch.device("cpu")
    cfg, sd, model = T.build("retrieval", dict(num_labels=2, loss_type="ce", margin=0.1), torch.float32, cpu)
    bank = model.rt.bank
    shard = compress == "shard"
    sync = GradSync(bank, shard=shard, bucket_bytes=(1 << 16) if shard else (64 << 20))
    sync.broadcast_parameters(0)
    opt = FusedAdamW(bank, lr=1e-3, betas=(0.9, 0.98), weight_decay=1e-3, cnn_lr=1e-3, max_grad_norm=5.0)
    out_dir = os.path.dirname(out_path) + f"/loop_{compress}"
    tcfg = SimpleNamespace(train_n_clips=2, num_frm=2, score_agg_func="lse", gradient_accumulation_steps=1, learning_rate=1e-3,
                           cnn_learning_rate=1e-3, decay="linear", cnn_lr_decay="linear", num_train_steps=2, warmup_ratio=0.0, valid_steps=2,
                           train_batch_size=1, max_n_example_per_group=1, output_dir=out_dir, save_steps_ratio=0.5)
    batches = []
    for i in range(2):                                       # this rank's share: one video (2 clips x 2 frames) + 2 texts per micro-step
        frames = S.synthetic_frames(1, 4, 64, 50 + 2 * i + rank).contiguous()
        ids, mask = S.synthetic_text(2, 6, 50 + 2 * i + rank, cfg["vocab_size"])
        batches.append(dict(visual_inputs=frames, text_input_ids=ids.clamp(max=cfg["vocab_size"] - 1), text_input_mask=mask,
                            labels=torch.tensor([1, 0]), n_examples_list=[2]))
    saver = C.ModelSaver(os.path.join(out_dir, "ckpt"))
    restorer = C.E2E_TrainingRestorer(tcfg, model, opt)
    end = tasks.start_training(model, opt, D.PrefetchLoader(batches, device=cpu), tcfg, sync=sync, model_saver=saver, restorer=restorer,
                               total_n_examples=12)
    assert end == 2
    if shard:
        sync.gather_state(opt)
    if rank == 0:
        ck = torch.load(os.path.join(out_dir, "restore.pt"))
        torch.save(dict(master=bank.master.clone(), exp_avg=bank.exp_avg.clone(), step=ck["global_step"],
                        ckpt_model={k: v for k, v in torch.load(os.path.join(out_dir, "ckpt", "model_step_2.pt")).items()},
                        restore_optim=ck["optim_state_dict"]["state"]), out_path)


def _worker(rank, world, port, out_path, compress=None, fn="one_step"):
    torch.set_num_threads(2)
    os.environ["EMUL_THREADS"] = "4"
    _setup_emul()
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        dict(one_step=_train_one_step, multi=_train_multi_clip_accumulated, infer=_sharded_inference, loop=_training_loop)[fn](rank, world, out_path, compress)
    finally:
        dist.destroy_process_group()


def _spawn_all(jobs):
    """start every (world, out_path, compress, fn) job at once (independent rendezvous ports) and join them all: the runs are
    dominated by process start-up and model construction, so they overlap well on the test host's cores"""
    ctxs = [mp.spawn(_worker, args=(world, _free_port(), out, compress, fn), nprocs=world, join=False) for world, out, compress, fn in jobs]
    for c in ctxs:
        while not c.join():
            pass


def test_dp2_multi_clip_loop_with_accumulation(tmp_path):
    p2, p1 = str(tmp_path / "dp2.pt"), str(tmp_path / "dp1.pt")
    _spawn_all([(2, p2, None, "multi"), (1, p1, None, "multi")])
    a, b = torch.load(p2), torch.load(p1)
    # accumulated gradients are SUMS over micro-steps of per-rank mean losses: DP=2 averages two half-size means
    assert abs(a["norm"] - b["norm"]) / b["norm"] < 1e-3
    torch.testing.assert_close(a["master"], b["master"], rtol=1e-4, atol=2e-6)


@pytest.fixture(scope="module")
def one_step_runs(tmp_path_factory):
    """one optimizer step of the same global batch under every exchange variant, all spawned at once (11 processes) and shared by the
    tests below: DP = 1; DP = 2 with fp32 / bf16 wire / bf16 wire consumed directly; DP = 2 owner-only update with fp32 / bf16 wire"""
    d = tmp_path_factory.mktemp("one_step")
    jobs = {name: (world, str(d / f"{name}.pt"), compress, "one_step")
            for name, world, compress in (("dp1", 1, None), ("dp2", 2, None), ("dp2c", 2, "bf16"), ("dp2d", 2, "bf16direct"),
                                          ("shard", 2, "shard"), ("shard16", 2, "shard16"))}
    _spawn_all(list(jobs.values()))
    return {k: torch.load(v[1]) for k, v in jobs.items()}


def test_dp2_equals_dp1_on_the_global_batch(one_step_runs):
    # dp2c: bf16 gradients on the wire; dp2d: ... consumed by AdamW without the cast back
    a, c, d, b = (one_step_runs[k] for k in ("dp2", "dp2c", "dp2d", "dp1"))
    torch.testing.assert_close(d["master"], c["master"], rtol=0, atol=0)            # same values as cast-back + fp32 AdamW, bit for bit
    assert d["norm"] == c["norm"]
    assert abs(a["norm"] - b["norm"]) / b["norm"] < 1e-3
    torch.testing.assert_close(a["master"], b["master"], rtol=1e-4, atol=2e-6)
    assert abs(c["norm"] - b["norm"]) / b["norm"] < 1e-2
    # the first AdamW step moves a weight by ~lr * sign(g) (1e-3): bf16 wire precision can flip the sign of a gradient that
    # is ~0, so bound the worst element by 2.5 lr and the mean deviation tightly
    diff = (c["master"] - b["master"]).abs()
    assert diff.max() < 2.5e-3 and diff.mean() < 2e-6, (diff.max(), diff.mean())


def test_dp2_owner_only_update_equals_dp1(one_step_runs):
    """GradSync(shard=True): reduce-scatter -> AdamW on the owned 1/world of every bucket -> all-gather of the new weights (the
    direct exchange of SURVEY 8e with the optimizer in between) gives the weights of the all-reduce path and of DP = 1."""
    s, s16, ar, ar16, one = (one_step_runs[k] for k in ("shard", "shard16", "dp2", "dp2d", "dp1"))
    # same reduced gradients as the all-reduce path; the norm partials are summed in another order (two ranks' halves)
    assert abs(s["norm"] - ar["norm"]) / ar["norm"] < 1e-5 and abs(s16["norm"] - ar16["norm"]) / ar16["norm"] < 1e-5
    torch.testing.assert_close(s["master"], ar["master"], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(s16["master"], ar16["master"], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(s["master"], one["master"], rtol=1e-4, atol=2e-6)
    for k in ("exp_avg", "exp_avg_sq"):                  # gather_state(): the moments of every piece from its owner
        torch.testing.assert_close(s[k], ar[k], rtol=1e-5, atol=1e-7)     # (atomic weight-gradient sums differ in the last bits run to run)


def test_dp2_training_loop_owner_only_checkpoints_equal_all_reduce(tmp_path):
    """start_training on two ranks, all-reduce vs owner-only update: same final weights, and the files rank 0 wrote (model_step_2.pt,
    restore.pt with the AdamW moments) hold the WHOLE state -- gathered from the owners before each save."""
    ps, pa = str(tmp_path / "loop_s.pt"), str(tmp_path / "loop_a.pt")
    _spawn_all([(2, ps, "shard", "loop"), (2, pa, None, "loop")])
    s, a = torch.load(ps), torch.load(pa)
    assert s["step"] == a["step"] == 2
    torch.testing.assert_close(s["master"], a["master"], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(s["exp_avg"], a["exp_avg"], rtol=1e-5, atol=1e-7)
    assert set(s["ckpt_model"]) == set(a["ckpt_model"])
    for k, v in a["ckpt_model"].items():
        if torch.is_tensor(v) and v.is_floating_point():
            torch.testing.assert_close(s["ckpt_model"][k], v, rtol=1e-5, atol=1e-7, msg=k)
    for name, st in a["restore_optim"].items():
        torch.testing.assert_close(s["restore_optim"][name]["exp_avg_sq"], st["exp_avg_sq"], rtol=1e-4, atol=1e-10, msg=name)


def test_sharded_retrieval_inference_gathers_all_rows(tmp_path):
    p2, p1 = str(tmp_path / "inf2.pt"), str(tmp_path / "inf1.pt")
    _spawn_all([(2, p2, None, "infer"), (1, p1, None, "infer")])
    a, b = torch.load(p2), torch.load(p1)
    assert a["rows"] == b["rows"]                        # same (vid, txt, score) rows, scores rounded to 4 places
    assert a["metrics"] == b["metrics"] and set(a["metrics"]) == {"text2video", "video2text"}
