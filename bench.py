#!/usr/bin/env python
"""MI355X benchmark of the ClipBERT hot path (metric of BASELINE.json: clips/sec/node at N_clip x N_frame = 2 x 2,
224 px, L_txt = 32).

Default (`--mode train`): one "step" = one full data-parallel TRAINING step of MSRVTT retrieval
(src/configs/msrvtt_ret_base_resnet50.json: 16 videos per GPU, 1 positive + 1 negative text per video, score_agg_func
"lse", AdamW + grad-norm clipping) at the METRIC's shape: every video contributes N_clip = 2 clips of N_frame = 2 frames,
224 x 224, L_txt = 32 -- 32 clips = 64 frames and 64 (text, clip) pairs per GPU per step.  Inside the timed step:
uint8 frames (resident in HBM) -> ImageNorm + BGR + NHWC pack -> ResNet-50 grid backbone -> cross-modal BERT over all
clips at once (the reference's clip loop folded into the batch) -> retrieval head -> LSE pooling over the clips + loss ->
backward -> gradient all-reduce over RCCL (N > 1) -> global-norm clipping + fused AdamW; dropout on, bf16 compute, fp32
master weights.  The device work of a step is a hipGraph; the optimizer's hyper-parameter upload and (N > 1) the RCCL
all-reduces are issued eagerly around / between the graphs.

Other workloads (rows of BASELINE.json `configs`; never the default line):
  --mode tgif     configs[3]: TGIF-QA action training step (ClipBertForMultipleChoice, 5 options per question, L_txt = 25,
                  N_clip = 2, mean pooling)
  --mode infer16  configs[4]: retrieval inference, one video x 16 clips against a mini-batch of 64 captions per step (grid
                  features computed once, encoder passes of 4 clips x 64 captions = 256 pairs); N > 1 shards the videos over the ranks and
                  gathers the (vid, txt, score) rows once at the end (inside the timed region)

Prints ONE JSON line (rank 0).  `roofline` describes the dominant kernel family of the step (bf16 MFMA GEMM /
implicit-GEMM conv), timed live with HIP events on the launch stream; `cpu_baseline` is the CPU oracle
(oracle/clipbert_oracle.py, a port of the reference arithmetic in stock PyTorch fp32 ops) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BF16_MFMA_PEAK_TFLOPS = 2500.0        # MI355X_MICROARCH.md: dense bf16 MFMA peak (2:1 sparsity excluded)
PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r06z_pmc_traffic.json")

BASE_CONFIG = dict(
    max_temporal_position_embeddings=100, backbone_channel_in_size=2048, max_grid_row_position_embeddings=100,
    max_grid_col_position_embeddings=100, attention_probs_dropout_prob=0.1, hidden_act="gelu", hidden_dropout_prob=0.1,
    hidden_size=768, initializer_range=0.02, intermediate_size=3072, layer_norm_eps=1e-12, max_position_embeddings=512,
    model_type="bert", num_attention_heads=12, num_hidden_layers=12, pad_token_id=0, type_vocab_size=2, vocab_size=30522,
    num_labels=2, loss_type="ce", margin=0.1)

MODES = {
    # mode: (head, defaults)
    "train": dict(head="retrieval", n_clips=2, frames=2, size=224, txt_len=32, repeat=2, pool="lse", videos=16),
    "tgif": dict(head="multiple_choice", n_clips=2, frames=2, size=224, txt_len=25, repeat=5, pool="mean", videos=16),
    "infer16": dict(head="retrieval", n_clips=16, frames=2, size=224, txt_len=32, repeat=64, pool="lse", videos=1),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", choices=sorted(MODES), default="train")
    ap.add_argument("--videos", type=int, default=None, help="videos per GPU per step (train_batch_size of the JSON config: 16)")
    ap.add_argument("--n-clips", type=int, default=None)
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--size", type=int, default=None)
    ap.add_argument("--txt-len", type=int, default=None)
    ap.add_argument("--repeat", type=int, default=None, help="text rows per video (train: pos + neg; tgif: options; infer16: captions per mini-batch)")
    ap.add_argument("--pool", choices=["mean", "max", "lse"], default=None)
    ap.add_argument("--no-fold", action="store_true", help="diagnostic: the reference's clip LOOP instead of one folded forward")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--forward-only", action="store_true", help="diagnostic: forward of the training batch only (not the reported metric)")
    a = ap.parse_args()
    for k, v in MODES[a.mode].items():
        if k != "head" and getattr(a, k, None) is None:
            setattr(a, k, v)
    a.head = MODES[a.mode]["head"]
    return a


_T0 = time.perf_counter()


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.perf_counter() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


# ---- launching N > 1: clipbert_amd/bench/launch.py (supervisor, fallback ladder, stub workers of the CPU tests) ---------------------------------
from clipbert_amd.bench.launch import ATTEMPTS, ATTEMPT_TIMEOUT_S, _marker, _run_attempts, _stub_worker, _touch, supervise  # noqa: E402,F401


def main():
    args = parse()
    rc = supervise(args)
    if rc is not None:
        sys.exit(rc)
    if os.environ.get("CB_BENCH_TEST_STUB"):
        return _stub_worker(args)
    import faulthandler
    faulthandler.enable()
    faulthandler.dump_traceback_later(240, repeat=True, file=sys.stderr)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    # test hooks (never set by the driver): run the N > 1 control flow on a 1-GPU box -- all ranks on GPU 0, gloo collectives
    if os.environ.get("CB_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    backend = os.environ.get("CB_BENCH_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl" and rank == 0 and "NCCL_DEBUG" not in os.environ:
            # rank 0 keeps RCCL's own account of the communicator it built (INIT lines only, in a file): rccl_summary() reads it
            os.environ["CB_BENCH_RCCL_LOG"] = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"cb_bench_rccl_{os.getpid()}_%p.log")
            os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,TUNING", NCCL_DEBUG_FILE=os.environ["CB_BENCH_RCCL_LOG"])
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launch with `python bench.py --gpus N`, or under "
                         f"torch.distributed.run with --nproc-per-node equal to --gpus)")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from clipbert_amd import modeling as M
    from clipbert_amd import ops
    from clipbert_amd import synthetic as S
    from clipbert_amd import tasks
    from clipbert_amd.dist import GradSync
    from clipbert_amd.optim import FusedAdamW

    log("imports done")
    train = args.mode in ("train", "tgif") and not args.forward_only
    cfg = dict(BASE_CONFIG)
    if args.head == "multiple_choice":
        cfg.update(num_labels=args.repeat, loss_type="ce")
        cls = M.ClipBertForMultipleChoice
    else:
        cls = M.ClipBertForVideoTextRetrieval
    model = M.ClipBert(cfg, detectron2_model_cfg="R-50-grid.yaml", transformer_cls=cls)
    model.load_state_dict(S.full_state_dict(cfg, args.head, 42), strict=True)
    model.to(dev)
    model.train(train)
    model.prepare(dtype=torch.bfloat16, device=dev, overlap_wgrad=int(os.environ.get("CB_OVERLAP_WGRAD", "0")))     # Runtime.overlap bits (diagnostic A/B)
    log("model prepared")
    bank = model.rt.bank

    bv, nclip, T, rep = args.videos, args.n_clips, args.frames, args.repeat
    # uint8 RGB frames, resident in HBM: ImageNorm (a1) runs inside the step, fused into the stem's input pack
    frames = S.synthetic_frames(bv, nclip * T, args.size, 42 + rank).to(dev)
    ids, mask = S.synthetic_text(bv * rep, args.txt_len, 42 + rank)
    ids, mask = ids.to(dev), mask.to(dev)
    tcfg = SimpleNamespace(train_n_clips=nclip, inference_n_clips=nclip, num_frm=T, score_agg_func=args.pool, task="action" if args.mode == "tgif" else None,
                           num_labels=cfg["num_labels"], inference_batch_size=rep, gradient_accumulation_steps=1, learning_rate=5e-5,
                           cnn_learning_rate=5e-5, decay="linear", cnn_lr_decay="linear", num_train_steps=100000, warmup_ratio=0.1)
    if args.mode == "tgif":
        labels = S.synthetic_labels(bv, rep, 42 + rank).to(dev)          # answer id per question
        counts = [1] * bv                                                # questions per video; x num_labels inside (run_video_qa.py:206)
    else:
        labels = torch.tensor(([1] + [0] * (rep - 1)) * bv, dtype=torch.long, device=dev)      # 1 positive + negatives per video
        counts = [rep] * bv
    batch = dict(visual_inputs=frames, text_input_ids=ids, text_input_mask=mask, labels=labels, n_examples_list=counts)
    fold = not args.no_fold

    sync = opt = None
    if train:
        # CB_COMM=native: buckets through the library's own RCCL entry point (cb_allreduce_bucket) instead of torch.distributed
        # CB_BENCH_DRY_DP=N (single process): the N-rank replay plan with every collective skipped -- times what data parallelism
        # costs a rank besides link time (three graphs instead of one, wire casts, bf16-direct AdamW).  A diagnostic, not the metric.
        dry_dp = int(os.environ.get("CB_BENCH_DRY_DP", "0")) if world == 1 else 0
        # CB_BENCH_LOOPBACK=1 (single process): the N-rank plan with REAL cb_allreduce_bucket calls on a world-size-1 communicator,
        # captured into the step's hipGraph -- shows that the whole exchange is capturable; a diagnostic, not the metric.
        loopback = world == 1 and not dry_dp and os.environ.get("CB_BENCH_LOOPBACK") == "1"
        sync = GradSync(bank, compress=None if os.environ.get("CB_BENCH_FP32_WIRE") == "1" else "bf16",
                        comm=os.environ.get("CB_COMM", "auto"), pretend_world=dry_dp, loopback=loopback,   # auto: the library's own RCCL entry points on "nccl"
                        shard=os.environ.get("CB_BENCH_SHARD") == "1")      # opt-in: reduce-scatter -> owner-only AdamW -> all-gather
        sync.broadcast_parameters(0)
        opt = FusedAdamW(bank, lr=5e-5, betas=(0.9, 0.98), weight_decay=1e-3, max_grad_norm=5.0, fold_norm=os.environ.get("CB_BENCH_NO_FOLD") is None)
    # ---- the pieces of a step: clipbert_amd.bench.step.make_step builds them (tests/test_bench_step.py checks exactly these closures: eager
    # == captured replay, gradients against the oracle's autograd) ------------------------------------------------------------------
    from clipbert_amd.bench import step as bench_step
    if train:
        _fns = bench_step.make_step(model, batch, tcfg, opt, sync, labels, counts, nclip, T, args.pool, fold=fold)
        state, one = _fns.state, _fns.one
        forward_loss, host_prepare, device_step_single = _fns.forward_loss, _fns.host_prepare, _fns.device_step
    else:
        state = {"global_step": 0}
        one = torch.ones((), dtype=torch.float32, device=dev)

        def forward_loss():
            stack = tasks.forward_clips_stack(model, batch, nclip, T, fold=fold, cfg=tcfg)       # (n_clips, pairs, C) logits
            return tasks.training_loss(model, stack, labels, counts, args.pool)                  # clip pooling (a20) + loss
        host_prepare = device_step_single = None

    # diagnostic step variants (CB_BENCH_PIPELINE / CB_BENCH_CHAINS): clipbert_amd/bench/diag.py
    from clipbert_amd.bench import diag as bench_diag
    _diag = bench_diag.build(SimpleNamespace(model=model, bank=bank, opt=opt, args=args, frames=frames, ids=ids, mask=mask, labels=labels, counts=counts,
                                             tcfg=tcfg, one=one, bv=bv, nclip=nclip, T=T, rep=rep, fold=fold, train=train, world=world))
    device_step_pipelined, device_step_chains, chains = _diag.device_step_pipelined, _diag.device_step_chains, _diag.chains
    T_GROUPS = _diag.T_GROUPS

    def train_step_eager():
        if chains > 1:
            host_prepare()
            return device_step_chains()
        host_prepare()
        opt.zero_grad(lazy=True)
        model.rt.pending_encoder_nodes = model.rt.pending_cnn_nodes = 0
        loss = forward_loss()
        loss.backward()                         # (N > 1: the transformer buckets leave from inside the encoder backward,
                                                #  grid_encoder + res5 from inside the ResNet backward)
        if model.rt.after_encoder_backward is None:
            sync.reduce_transformer()
        sync.reduce_cnn()
        g16 = sync.wire_gradients()             # N > 1, bf16 wire: AdamW reads the reduced image directly (no cast back to fp32)
        sync.wait(cast_back=g16 is None)
        ops.counter_add(model.rt.seed_dev)
        dp_update(g16)
        return loss

    def dp_update(g16):
        """the optimizer behind a finished gradient exchange: every rank updates everything (all-reduce), or (CB_BENCH_SHARD=1) only
        the 1/world of every bucket it received from the reduce-scatter, followed by the all-gather of the new weights"""
        if sync.shard:
            opt.launch(grad16=g16, pieces=sync.owned_pieces(), norm_reduce=sync.norm_all_reduce)
            sync.gather_updated()
        else:
            opt.launch(grad16=g16)

    def forward_only_step():
        with torch.no_grad():
            return forward_loss()

    infer_rows = []

    def infer_step():
        sc = tasks.inference_retrieval_video(model, frames[:1], ids[:rep], mask[:rep], tcfg, cache_cnn=True)
        infer_rows.extend(dict(vid_id=f"r{rank}v{len(infer_rows) // rep}", txt_id=j, score=s) for j, s in enumerate(sc))
        return None

    PAIRS_PER_PASS = 256                        # clips x captions per encoder pass (tasks.inference_retrieval_video)

    def infer_device_step():
        """the device work of infer_step without the host-side score list (capturable)"""
        from clipbert_amd import clips
        with torch.no_grad():
            grid = model.grid_features(frames[:1].view(nclip, T, *frames.shape[2:]))
            cpp = max(1, min(nclip, PAIRS_PER_PASS // rep))
            per_clip = []
            for c0 in range(0, nclip, cpp):
                nc = min(cpp, nclip - c0)
                out = model.forward_from_grid(dict(visual_inputs=grid[c0:c0 + nc], text_input_ids=ids[:rep].repeat(nc, 1),
                                                   text_input_mask=mask[:rep].repeat(nc, 1), labels=None, n_examples_list=[rep] * nc))
                per_clip.extend(out["logits"].view(nc, rep, -1).unbind(0))
            pooled = clips.aggregate_clip_logits(per_clip, args.pool)
            if args.pool == "lse":
                pooled = clips.lse_inference_logits(pooled)
            state["scores"] = torch.softmax(pooled.float(), dim=1)[:, 1]
        return None

    if args.mode == "infer16":
        eager_fn = infer_step
    elif not train:
        eager_fn = forward_only_step
    else:
        sync.set_cnn_split(M.cnn_early_split(model))           # grid_encoder + res5 gradients can leave before res4 / res3 are done
        if world > 1 or sync.loopback:
            model.rt.after_encoder_backward = sync.reduce_transformer
            model.rt.after_res5_backward = sync.reduce_cnn_early
        eager_fn = train_step_eager

    log("inputs ready")
    # ---- eager warm-up (also builds pixel tables etc.) ---------------------------------------------------------------------
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            loss = eager_fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    log("eager warm-up done" + (f", loss {float(loss.item()):.4f}" if loss is not None else ""))
    dp_check = None
    if train and world > 1:
        dp_check = dp_self_check(bank, dist, dev, compute_weights=sync.shard)
        log(f"DP self-check: {dp_check}")

    # ---- replay plan -----------------------------------------------------------------------------------------------------
    #   N = 1          : host_prepare (eager) + ONE hipGraph with the whole device step.
    #   N > 1 (default): host_prepare + three hipGraphs with EAGER RCCL all-reduces (bf16 on the wire, 64 MiB buckets) between
    #                    them: [zero + forward + heads/encoder backward] -> transformer buckets (async: they cross xGMI while
    #                    the next graph runs) -> [ResNet backward] -> CNN buckets -> wait -> [clip + AdamW].  Only this
    #                    library's kernels are captured: collectives stay out of the graphs.
    #   CB_BENCH_PLAN=eager / --no-graph: no graphs (the hook issues the transformer buckets from inside the backward).
    def capture(fn):
        g = torch.cuda.CUDAGraph()
        # N > 1: the process group's watchdog thread polls HIP events while this thread captures; "thread_local" keeps calls made
        # by OTHER threads from invalidating the capture (the default "global" mode would)
        with torch.cuda.graph(g, capture_error_mode="thread_local" if world > 1 else "global"):
            out = fn()
        return g, out

    flush_pipeline = None
    recapture_dp = None
    plan_env = os.environ.get("CB_BENCH_PLAN", "")
    run, plan, n_graphs = eager_fn, "eager", 0
    use_graph = not args.no_graph and plan_env != "eager"
    if use_graph and args.mode == "infer16":
        g1, _ = capture(infer_device_step)

        def run_infer():
            g1.replay()
            sc = [round(x, 4) for x in state["scores"].tolist()]           # D2H of the 64 scores: the step's result
            infer_rows.extend(dict(vid_id=f"r{rank}v{len(infer_rows) // rep}", txt_id=j, score=x) for j, x in enumerate(sc))
        run, plan, n_graphs = run_infer, "one hipGraph per (video, caption mini-batch) + score read-back", 1
    elif use_graph and not train:
        g1, loss = capture(forward_only_step)
        run, plan, n_graphs = g1.replay, "one hipGraph", 1
    elif use_graph and train and sync.active and not sync.dry and sync.shard and sync.carrier != "native":
        run, plan, n_graphs = eager_fn, "eager (owner-only update over torch.distributed: not captured)", 0
    elif use_graph and train and sync.active and not sync.dry and sync.carrier == "native" and (plan_env != "split" or sync.loopback or sync.shard):
        # ONE hipGraph for the whole data-parallel step: the bucket all-reduces (cb_allreduce_bucket on GradSync's comm stream,
        # forked / joined by events) are captured with the kernels -- no host between the backward and the collectives.  The DEFAULT
        # for N > 1 (round 4); CB_BENCH_PLAN=split selects the four-graph plan below, which is also what the supervisor (ATTEMPTS)
        # falls back to when this plan fails or hangs on a node.
        def device_step_dp():
            opt.zero_grad(lazy=True)
            model.rt.pending_encoder_nodes = model.rt.pending_cnn_nodes = 0
            loss_ = forward_loss()
            loss_.backward(one)                    # the hooks issue the transformer / grid_encoder + res5 buckets from inside
            if model.rt.after_encoder_backward is None:
                sync.reduce_transformer()
            sync.reduce_cnn()
            g16 = sync.wire_gradients()
            sync.wait(cast_back=g16 is None)
            ops.counter_add(model.rt.seed_dev)
            dp_update(g16)
            return loss_
        g1, loss = capture(device_step_dp)
        dp_graph = {"g": g1}

        def run_dp():
            host_prepare()
            dp_graph["g"].replay()

        def recapture_dp():
            dp_graph["g"] = capture(device_step_dp)[0]
        run, plan, n_graphs = run_dp, "eager hyper-parameter upload + one hipGraph with the bucketed bf16 all-reduces captured inside", 1
    elif use_graph and world == 1 and not (train and sync.dry):
        pipelined = train and chains == 1 and fold and os.environ.get("CB_BENCH_PIPELINE", "0") == "1" and not os.environ.get("CB_BENCH_TUNE")
        g1, loss = capture(device_step_pipelined if pipelined else (device_step_chains if chains > 1 else device_step_single))
        if os.environ.get("CB_BENCH_TUNE") and train and chains == 1:
            # tools/tune_instep.py: launch configurations judged by the whole captured step (a tuning run, not a measurement)
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import tune_instep
            tune_instep.run(lambda: capture(device_step_single)[0], host_prepare, os.environ["CB_BENCH_TUNE"],
                            os.environ.get("CB_BENCH_TUNE_CAND", os.path.join(ROOT, "profiles", "r03f_gemm_tuning_cold.json")),
                            mode=os.environ.get("CB_BENCH_TUNE_MODE", args.mode), max_shapes=int(os.environ.get("CB_BENCH_TUNE_SHAPES", "45")), max_cands=int(os.environ.get("CB_BENCH_TUNE_CANDS", "4")))
            g1, loss = capture(device_step_single)
        if os.environ.get("CB_BENCH_TUNE_WGRAD") and train and chains == 1:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import tune_instep
            tune_instep.run_wgrad_groups(lambda: capture(device_step_single)[0], host_prepare, os.environ["CB_BENCH_TUNE_WGRAD"])
            g1, loss = capture(device_step_single)

        def run_single():
            host_prepare()
            g1.replay()
            if pipelined:
                opt.deferred_pending = True          # this step's transformer update rides in the next replay (or the final flush)

        def flush_pipeline():
            if pipelined and opt.deferred_pending:
                opt.launch(groups=T_GROUPS, reuse_norm=True)                 # the update the last replay left behind
                opt.deferred_pending = False
        run, plan, n_graphs = run_single, "eager hyper-parameter upload + one hipGraph" + (
            " (transformer AdamW of step i software-pipelined beside the ResNet forward of step i+1; flushed inside the timed region)" if pipelined else ""), 1
    elif use_graph:
        # The backward is cut at the grid features: autograd.grad(loss, grid) runs the heads' and the encoder's backward
        # (parameter gradients land in the flat buffer as a side effect), grid.backward(dgrid) runs the CNN trunk's.
        model.rt.after_encoder_backward = None                 # collectives are issued between the graphs, eagerly
        model.rt.after_res5_backward = None
        cut = {}

        def part_a():
            opt.zero_grad(lazy=True)
            model.rt.pending_encoder_nodes = model.rt.pending_cnn_nodes = 0
            vis = frames.view(bv * nclip, T, *frames.shape[2:]) if (fold and nclip > 1) else frames
            assert fold or nclip == 1, "the split replay plan needs the folded clip forward (one encoder node)"
            grid = model.grid_features(vis)
            mini = dict(visual_inputs=grid, text_input_ids=ids, text_input_mask=mask, labels=None,
                        n_examples_list=tasks._pair_counts(tcfg, counts))
            lg = model.forward_from_grid(mini, clip_fold=nclip)["logits"]
            stack = lg.reshape(nclip, lg.shape[0] // nclip, *lg.shape[1:])
            loss_ = tasks.training_loss(model, stack, labels, counts, args.pool)
            (dgrid,) = torch.autograd.grad(loss_, [grid])
            cut["grid"], cut["dgrid"] = grid, dgrid
            sync.cast_transformer()                # the bf16 wire image of the finished range is produced inside the graph
            return loss_

        def part_b():
            cut["grid"].backward(cut["dgrid"])
            sync.cast_cnn()

        # The ResNet backward in two graphs: [grid_encoder + res5] -> their buckets leave -> [res4, res3] -> the rest.  The
        # generator (modeling.cnn_backward_steps) is driven by hand with the saved activations of the grid-feature autograd node.
        def part_b1():
            node = cut["grid"].grad_fn
            cut["cnn_steps"] = M.cnn_backward_steps(node.bb, node.pack, cut["dgrid"].contiguous())
            if next(cut["cnn_steps"], None) is None:
                cut["cnn_steps"] = None                            # nothing after res5: the generator already ran to its end
            sync.cast_cnn_early()

        def part_b2():
            if cut["cnn_steps"] is not None:
                for _ in cut["cnn_steps"]:
                    pass
            cut["cnn_steps"] = None
            sync.cast_cnn(late_only=True)

        def part_c():
            ops.counter_add(model.rt.seed_dev)
            opt.launch(grad16=sync.wire_gradients())

        wire16 = sync.wire_gradients() is not None
        split_cnn = bool(sync.c_early) and wire16 and os.environ.get("CB_BENCH_CNN_SPLIT", "1") != "0"
        ga, loss = capture(part_a)
        if split_cnn:
            gb1, _ = capture(part_b1)
            gb2, _ = capture(part_b2)
        else:
            gb, _ = capture(part_b)
        gc, _ = capture(part_c)

        def run_split():
            host_prepare()
            ga.replay()
            sync.reduce_transformer(cast=not wire16)       # only the collectives are issued eagerly
            if split_cnn:
                gb1.replay()
                sync.reduce_cnn_early(cast=False)          # grid_encoder + res5 cross xGMI while res4 / res3 run
                gb2.replay()
                sync.reduce_cnn(cast=False)                # the remaining middle of the CNN range
            else:
                gb.replay()
                sync.reduce_cnn(cast=not wire16)
            sync.wait(cast_back=not wire16)
            gc.replay()
        n_graphs = 4 if split_cnn else 3
        run, plan = run_split, (("four" if split_cnn else "three") + " hipGraphs, eager bucketed bf16 all-reduces (transformer buckets overlap the ResNet "
                                "backward" + ("; grid_encoder + res5 buckets leave after res5, overlapping res4 / res3" if split_cnn else "") + ")")
    log(f"replay plan: {plan}")

    # clock / power-state settling (untimed, before the W warm-up steps): right after process start the first replays run
    # ~15 % slow on some boxes; ~0.75 s of steady replays brings the GPU to its sustained clocks
    # (a FIXED number of replays: with N > 1 every rank must issue the same number of collectives)
    settle = (1, 1) if os.environ.get("CB_BENCH_SHARE_GPU") == "1" else (8, 6)      # (the 1-GPU dry run of N > 1 goes through gloo: slow)
    for _ in range(settle[0]):
        for _ in range(settle[1]):
            run()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        run()
    infer_rows.clear()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    if flush_pipeline is not None:
        flush_pipeline()                                   # (inside the timed region: every step's whole update is paid for)
    gathered = None
    if args.mode == "infer16":
        gathered = tasks.gather_retrieval_rows(infer_rows)              # the job's only exchange: (vid, txt, score) rows
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    ms_per_step = elapsed / args.steps * 1e3
    clips_per_step = bv * nclip * world
    value = clips_per_step / (elapsed / args.steps)
    final_loss = float(loss.item()) if loss is not None else None
    log(f"timed region done: {ms_per_step:.3f} ms/step, {value:.1f} clips/s")

    pairs = bv * rep * nclip
    if args.mode == "train" and not args.forward_only:
        metric = "clips/sec/node (2×2 frames, 224px, L_txt=32) at 1/2/4/8 MI355X"
        if (nclip, T, args.size, args.txt_len) != (2, 2, 224, 32):
            metric = f"clips/sec/node ({nclip}×{T} frames, {args.size}px, L_txt={args.txt_len}) -- NOT the BASELINE.json shape (diagnostic)"
        workload = (f"MSRVTT retrieval training step (msrvtt_ret_base_resnet50.json; BASELINE configs[1]/[2] family) at the metric's shape: "
                    f"{bv} videos x N_clip={nclip} x N_frame={T} uint8 frames {args.size}px + {rep} texts/video (pos+neg, r={rep}) L_txt={args.txt_len} "
                    f"per GPU = {bv * nclip} clips, {bv * nclip * T} frames, {pairs} (text, clip) pairs per step; ImageNorm + fwd + "
                    f"{args.pool}-pooling loss + bwd + all-reduce + clip + AdamW in the timed step; clips {'folded into one forward' if fold else 'looped'}")
    elif args.mode == "tgif" and not args.forward_only:
        metric = f"clips/sec/node, TGIF-QA action training ({nclip}×{T} frames, {args.size}px, L_txt={args.txt_len}, {rep} options) -- BASELINE configs[3] row"
        workload = (f"BASELINE configs[3]: TGIF-QA action training step (ClipBertForMultipleChoice), {bv} videos x N_clip={nclip} x N_frame={T} "
                    f"frames {args.size}px, {rep} options/question L_txt={args.txt_len} = {pairs} pairs per GPU per step, {args.pool} pooling")
    elif args.mode == "infer16":
        metric = f"clips/sec/node, retrieval inference ({nclip} clips × {T} frames, {args.size}px, {rep}-caption mini-batches) -- BASELINE configs[4] row"
        workload = (f"BASELINE configs[4]: retrieval inference, per step 1 video x {nclip} clips x {T} frames {args.size}px against {rep} captions "
                    f"L_txt={args.txt_len} per GPU (CNN once per video, encoder passes of <= 256 (clip, caption) pairs, {args.pool} pooling, scores rounded to 4 "
                    f"places and read back); videos sharded over ranks, one gather of the score rows at the end of the timed region")
    else:
        metric = "clips/sec/node forward-only (diagnostic)"
        workload = f"forward of the {args.mode} batch only"
    out = {
        "metric": metric, "value": round(value, 2), "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": workload, "mode": args.mode, "videos_per_gpu": bv, "n_clips": nclip, "n_frames": T, "img_size": args.size,
                   "txt_len": args.txt_len, "texts_per_video": rep, "pairs_per_gpu": pairs, "score_agg_func": args.pool, "clips_folded": fold,
                   "input": "uint8 frames in HBM",
                   "parallelism": f"dp{world}" + (f" (DRY RUN of the dp{sync.world} plan on one GPU: no collectives)" if (train and sync is not None and sync.dry) else "")
                   + (" (LOOPBACK: the dp plan with world-size-1 RCCL collectives)" if (train and sync is not None and sync.loopback) else ""), "hip_graph": use_graph, "replay_plan": plan, "n_graphs": n_graphs,
                   "dropout": bool(train), "final_loss": None if final_loss is None else round(final_loss, 5)},
    }
    if dp_check is not None:
        out["config"]["dp_self_check"] = dp_check
    if world > 1:
        out["config"]["attempt"] = int(os.environ.get("CB_BENCH_ATTEMPT", "0"))
        out["config"]["attempt_env"] = ATTEMPTS[out["config"]["attempt"]] if os.environ.get("CB_BENCH_WORKER") == "1" else "launcher-pinned"
    if train and world > 1:
        out["config"]["rccl"] = rccl_summary()
    if train and sync is not None and sync.active and not sync.dry:
        if sync.shard:
            out["config"]["update"] = "owner-only: reduce-scatter -> AdamW on 1/world of every bucket -> all-gather of the new weights"
        out["config"]["grad_exchange"] = f"{sync.carrier} ({('cb_reduce_scatter_bucket / cb_allgather_bucket' if sync.shard else 'cb_allreduce_bucket') + ': RCCL behind the C ABI' if sync.carrier == 'native' else 'torch.distributed ' + backend}), bf16 wire, 64 MiB buckets"
    if gathered is not None:
        out["config"]["rows_gathered"] = len(gathered)
    # Exposed communication of the plan (N > 1, after the timed region, never part of `value`): the same plan with every collective
    # muted (GradSync.mute: casts, bucket bookkeeping, graphs, bf16-direct AdamW all stay), timed the same way; the difference is the
    # link time the plan failed to hide.  The ranks' parameters diverge from here on -- nothing after this uses them.  The measured
    # line is COMPLETE before this starts: if the diagnostic hangs (it re-captures the step), a watchdog prints the line without it,
    # marks the attempt done and exits -- a diagnostic must never cost the measurement.
    if train and world > 1 and sync is not None and sync.active and not sync.dry and os.environ.get("CB_BENCH_EXPOSED", "1") != "0":
        import threading

        def bail():
            out["config"]["exposed_comm"] = {"error": "the muted-collectives run did not finish in time"}
            if rank == 0:
                print(json.dumps(out), flush=True)
            if os.environ.get("CB_BENCH_JOB"):
                _touch(_marker(os.environ["CB_BENCH_JOB"], int(os.environ.get("CB_BENCH_ATTEMPT", "0")), f"r{rank}.done"))
            os._exit(0)
        timer = threading.Timer(float(os.environ.get("CB_BENCH_EXPOSED_DEADLINE", "120")), bail)
        timer.daemon = True
        timer.start()
        try:
            sync.mute = True
            if recapture_dp is not None:
                recapture_dp()
            for _ in range(3):
                run()
            dist.barrier()
            torch.cuda.synchronize()
            m0 = time.perf_counter()
            for _ in range(args.steps):
                run()
            torch.cuda.synchronize()
            dist.barrier()
            muted = torch.tensor([time.perf_counter() - m0], dtype=torch.float64, device=dev)
            dist.all_reduce(muted, op=dist.ReduceOp.MAX)
            muted_ms = float(muted.item()) / args.steps * 1e3
            out["config"]["exposed_comm"] = {"ms_per_step_collectives_muted": round(muted_ms, 3), "exposed_comm_ms": round(ms_per_step - muted_ms, 3),
                                             "exposed_comm_frac_of_step": round((ms_per_step - muted_ms) / ms_per_step, 4)}
            log(f"same plan with the collectives muted: {muted_ms:.3f} ms/step -> exposed communication {ms_per_step - muted_ms:.3f} ms")
        except Exception as e:                                  # noqa: BLE001
            out["config"]["exposed_comm"] = {"error": repr(e)[:200]}
        timer.cancel()
    if rank == 0 and world == 1 and not args.no_roofline:
        default_shape = args.mode == "train" and (nclip, T, args.size, args.txt_len, bv, rep) == (2, 2, 224, 32, 16, 2) and not args.forward_only
        out["roofline"] = measure_roofline(eager_fn if args.mode != "infer16" else infer_device_step, pmc_ok=default_shape, ms_per_step=ms_per_step)
        log("roofline measured")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, args)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if os.environ.get("CB_BENCH_JOB"):                     # tells this rank's supervisor that everything that matters is done
        _touch(_marker(os.environ["CB_BENCH_JOB"], int(os.environ.get("CB_BENCH_ATTEMPT", "0")), f"r{rank}.done"))
    if world > 1:
        dist.destroy_process_group()
        if os.environ.get("CB_BENCH_WORKER") == "1":
            # a supervised worker has nothing left to do: skip the interpreter's teardown (process-group / runtime threads of N ranks
            # shutting down in arbitrary order must not be able to hold the job up)
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(0)


def rccl_summary():
    """what RCCL said about itself (rank 0's NCCL_DEBUG=INFO log, written to NCCL_DEBUG_FILE by main()): version, channel count,
    and which algorithms / protocols it set up -- the bench line records what carried the buckets, it does not choose it"""
    import glob
    import re
    info = {"NCCL_ALGO": os.environ.get("NCCL_ALGO"), "NCCL_PROTO": os.environ.get("NCCL_PROTO")}
    try:
        path = os.environ.get("CB_BENCH_RCCL_LOG")
        files = sorted(glob.glob(path.replace("%p", "*"))) if path else []
        text = "".join(open(f, errors="replace").read() for f in files[:4])
        ver = re.search(r"(RCCL|NCCL) version[ :]*([0-9][^\s]*)", text)
        info["version"] = ver.group(0) if ver else None
        ch = re.findall(r"(\d+) coll channels", text)
        info["coll_channels"] = int(ch[-1]) if ch else None
        info["rings_connected"] = "Connected all rings" in text
        info["trees_connected"] = "Connected all trees" in text
        algo = re.findall(r"Algo(?:rithm)?[ =:]+(\w+).{0,40}?Proto(?:col)?[ =:]+(\w+)", text)
        info["algo_proto_seen"] = sorted({f"{a}/{p}" for a, p in algo})[:8] or None
    except Exception as e:                                  # noqa: BLE001
        info["error"] = repr(e)[:120]
    return info


def dp_self_check(bank, dist, dev, compute_weights=False):
    """After one eager data-parallel step every rank must hold identical parameters (same all-reduced gradients, same
    AdamW update): compares an order-independent checksum of the fp32 masters across the ranks (owner-only update: of the bf16
    compute weights, which are what the all-gather distributes -- the fp32 masters of the decay groups live on their owners)."""
    w = bank.w16 if (compute_weights and bank.w16 is not None) else bank.master
    local = torch.stack([w.double().sum(), w.double().abs().sum()]).to(dev)
    lo, hi = local.clone(), local.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if bool(torch.equal(lo, hi)):
        return "ok: parameters bit-identical on all ranks after a step"
    rel = float(((hi - lo).abs() / hi.abs().clamp_min(1e-30)).max())
    if rel < 1e-9:
        return f"ok: parameter checksums agree across ranks to {rel:.1e} (not bit-identical)"
    return f"MISMATCH: parameter checksums differ across ranks by {rel:.1e} ({lo.tolist()} vs {hi.tolist()})"


def measure_roofline(step_fn, pmc_ok=True, ms_per_step=None):
    """Every GEMM problem of the step, by family, each against THE ROOF THAT BOUNDS IT (clipbert_amd/gemm_log.py): the encoder linears,
    the 3x3 / 7x7 convolutions, the ResNet 1x1 convolutions with K > 256 and all weight gradients against the dense bf16 MFMA peak with
    their algorithmic flops (2*M*N*K); the ResNet 1x1 convolutions with K <= 256 against the HBM peak with their algorithmic bytes
    (operands + output + the epilogue's M x N operands, each once).

    One eager step is recorded (every cb_gemm / cb_gemm_group call with its live operands); each family's calls are then replayed back
    to back -- captured in a hipGraph, `reps` times -- between ONE pair of HIP events on the launch stream: avg launch duration =
    elapsed / launches, the figure rocprofv3's kernel trace of the same command reports (profiles/).  `kernel` = the family with the
    most time; `step` = all GEMM flops of the step over the measured wall time of the whole step."""
    from clipbert_amd import gemm_log
    fams = gemm_log.family_table(step_fn)
    dom_name, dom = max(fams.items(), key=lambda kv: kv[1]["ms"])
    mf = {k: v for k, v in fams.items() if v["bound"] == "mfma"}
    tot_fl = sum(v["gflop"] for v in fams.values())
    tot_ms = sum(v["ms"] for v in fams.values())
    # HBM bytes per launch of the dominant family from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate
    # runs, gfx950 correction applied: tools/pmc_traffic.py); null when no PMC data is committed for this workload / family
    traffic, over = None, None
    try:
        if not pmc_ok:                       # the committed PMC passes were taken on the default (metric) workload only
            raise KeyError("no PMC data for this workload")
        with open(PMC_TRAFFIC_FILE) as fh:
            fam = json.load(fh)["families"][dom_name]
        traffic, over = fam["hbm_bytes_per_launch"], fam.get("hbm_over_algorithmic")
    except Exception:
        traffic = None
    hbm = dom["bound"] == "hbm"
    out = {"bound": dom["bound"], "kernel": f"cb_gemm<bf16> {dom_name}" if not dom_name.startswith("fused") else dom_name, "launches": dom["launches"], "avg_launch_us": dom["avg_launch_us"],
           "achieved": dom["gbs"] if hbm else dom["tflops"], "peak": gemm_log.HBM_PEAK_GBS if hbm else BF16_MFMA_PEAK_TFLOPS,
           "unit": "GB/s" if hbm else "TFLOP/s", "frac": dom["frac"],
           "traffic": traffic, "traffic_unit": f"HBM bytes per launch (PMC, {os.path.relpath(PMC_TRAFFIC_FILE, ROOT)})",
           "hbm_over_algorithmic": over,
           "algorithmic_flop_per_launch": round(dom["gflop"] * 1e9 / dom["launches"]),
           "algorithmic_bytes_per_launch": round(dom["algorithmic_mbytes"] * 1e6 / dom["launches"]),
           "families": fams,
           "all_gemm_kernels": {"achieved": round(tot_fl / tot_ms, 2), "frac": round(tot_fl / tot_ms / BF16_MFMA_PEAK_TFLOPS, 4),
                                "time_ms": round(tot_ms, 3), "gflop_per_step": round(tot_fl, 1), "launches": sum(v["launches"] for v in fams.values()),
                                "mfma_bound_families_frac": round(sum(v["gflop"] for v in mf.values()) / max(1e-9, sum(v["ms"] for v in mf.values())) / BF16_MFMA_PEAK_TFLOPS, 4)}}
    if ms_per_step:
        out["step"] = {"gflop": round(tot_fl, 1), "ms": round(ms_per_step, 3), "tflops": round(tot_fl / ms_per_step, 1),
                       "frac": round(tot_fl / ms_per_step / BF16_MFMA_PEAK_TFLOPS, 4),
                       "note": "GEMM / conv flops of the step (attention, LayerNorm, AdamW not counted) over the wall time of the WHOLE step, vs the dense bf16 MFMA peak"}
    return out


def cpu_baseline(cfg, args):
    """The CPU oracle (port of the reference arithmetic in stock PyTorch fp32) on a bounded sample of the same workload --
    2 videos of the step's shape, the reference's clip LOOP, pooled loss; forward + backward (training modes) or forward only
    (infer16) -- on this host's cores."""
    from clipbert_amd import synthetic as S
    from oracle import clipbert_oracle as O
    ncores = min(os.cpu_count() or 1, 32)      # more threads than this only adds contention for 4-frame convs
    torch.set_num_threads(ncores)
    train = args.mode != "infer16"
    sd = S.full_state_dict(cfg, args.head, 42)
    sd = {k: v.clone().requires_grad_(train and v.is_floating_point() and ".norm." not in k and "stem" not in k and "res2" not in k)
          for k, v in sd.items()}
    nv = 1 if args.mode == "infer16" else 2
    nclip = min(args.n_clips, 2) if args.mode == "infer16" else args.n_clips
    rep = min(args.repeat, 8) if args.mode == "infer16" else args.repeat
    frames = S.synthetic_frames(nv, nclip * args.frames, args.size, 42)
    ids, mask = S.synthetic_text(nv * rep, args.txt_len, 42)
    vis = O.image_norm(frames, S.PIXEL_MEAN, S.PIXEL_STD).view(nv, nclip, args.frames, 3, args.size, args.size)
    if args.mode == "tgif":
        labels = S.synthetic_labels(nv, rep, 42)
    else:
        labels = torch.tensor(([1] + [0] * (rep - 1)) * nv)

    def one():
        per_clip = []
        for c in range(nclip):                                # the reference's clip loop (run_video_retrieval.py:396-401)
            b = dict(visual_inputs=vis[:, c], text_input_ids=ids, text_input_mask=mask, n_examples_list=[rep] * nv)
            per_clip.append(O.clipbert_forward(sd, b, cfg, args.head)["logits"])
        pooled = O.aggregate_clip_logits(per_clip, args.pool)
        if not train:
            return
        if args.pool == "lse":
            loss = O.lse_train_loss(pooled, labels).mean()
        else:
            loss = torch.nn.functional.cross_entropy(pooled.view(-1, rep) if args.mode == "tgif" else pooled, labels)
        loss.backward()

    ctx = torch.enable_grad() if train else torch.no_grad()
    with ctx:
        one()                                   # warm-up
        times = []
        t_budget = time.perf_counter()
        while len(times) < 5 and time.perf_counter() - t_budget < 25:
            t0 = time.perf_counter()
            one()
            times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    return {"value": round(nv * nclip / med, 3), "unit": "clips/s", "cores": ncores, "kind": "port",
            "sample": f"{nv} videos x {nclip} clips x {args.frames} frames {args.size}px + {nv * rep} texts L_txt={args.txt_len}, "
                      f"{'fwd+bwd (no optimizer)' if train else 'forward only'}, clip loop as in the reference, median of {len(times)} iterations, "
                      f"torch {torch.__version__} fp32, {ncores} threads"}


if __name__ == "__main__":
    main()
