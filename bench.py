#!/usr/bin/env python
"""MI355X benchmark of the ClipBERT hot path (metric of BASELINE.json: clips/sec/node).

One "step" = one full data-parallel TRAINING step of BASELINE configs[1] at the headline shape
(MSRVTT retrieval, N_clip = 1, N_frame = 2, 224x224, L_txt = 32, 16 videos x (pos+neg) text per GPU):
forward (ResNet-50 grid backbone -> cross-modal BERT -> retrieval head -> CE), backward, gradient
all-reduce over RCCL (N > 1), global-norm clipping and fused AdamW -- dropout on, bf16 compute, fp32
master weights.  Inputs are resident in HBM before the timed region.  The whole step is captured in a
hipGraph (torch.cuda.graph) and replayed.

Prints ONE JSON line (rank 0).  `roofline` describes the dominant kernel family of the step (the
bf16 MFMA GEMM / implicit-GEMM conv kernel), timed live with HIP events on the launch stream;
`cpu_baseline` is the CPU oracle (oracle/clipbert_oracle.py, a port of the reference arithmetic in
stock PyTorch fp32 ops) running the same fwd+bwd on a bounded sample on this host.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BF16_MFMA_PEAK_TFLOPS = 2500.0        # MI355X_MICROARCH.md: dense bf16 MFMA peak (2:1 sparsity excluded)

BASE_CONFIG = dict(
    max_temporal_position_embeddings=100, backbone_channel_in_size=2048, max_grid_row_position_embeddings=100,
    max_grid_col_position_embeddings=100, attention_probs_dropout_prob=0.1, hidden_act="gelu", hidden_dropout_prob=0.1,
    hidden_size=768, initializer_range=0.02, intermediate_size=3072, layer_norm_eps=1e-12, max_position_embeddings=512,
    model_type="bert", num_attention_heads=12, num_hidden_layers=12, pad_token_id=0, type_vocab_size=2, vocab_size=30522,
    num_labels=2, loss_type="ce", margin=0.1)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--videos", type=int, default=16, help="videos per GPU per step (train_batch_size of msrvtt_ret_base_resnet50.json)")
    ap.add_argument("--n-clips", type=int, default=1)
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--txt-len", type=int, default=32)
    ap.add_argument("--repeat", type=int, default=2, help="text examples per video (pos + neg)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--forward-only", action="store_true", help="diagnostic: inference throughput (not the reported metric)")
    return ap.parse_args()


_T0 = time.perf_counter()


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.perf_counter() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def main():
    import faulthandler
    faulthandler.enable()
    faulthandler.dump_traceback_later(240, repeat=True, file=sys.stderr)
    args = parse()
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
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from clipbert_amd import clips
    from clipbert_amd import modeling as M
    from clipbert_amd import ops
    from clipbert_amd import synthetic as S
    from clipbert_amd.dist import GradSync
    from clipbert_amd.optim import FusedAdamW

    log("imports done")
    cfg = dict(BASE_CONFIG)
    model = M.ClipBert(cfg, detectron2_model_cfg="R-50-grid.yaml", transformer_cls=M.ClipBertForVideoTextRetrieval)
    model.load_state_dict(S.full_state_dict(cfg, "retrieval", 42), strict=True)
    model.to(dev)
    model.train(not args.forward_only)
    model.prepare(dtype=torch.bfloat16, device=dev, overlap_wgrad=os.environ.get("CB_OVERLAP_WGRAD", "0") != "0")
    log("model prepared")
    bank = model.rt.bank
    sync = GradSync(bank, compress=None if os.environ.get("CB_BENCH_FP32_WIRE") == "1" else "bf16")
    sync.broadcast_parameters(0)
    opt = FusedAdamW(bank, lr=5e-5, betas=(0.9, 0.98), weight_decay=1e-3, max_grad_norm=5.0)
    model.rt.after_encoder_backward = sync.reduce_transformer

    bv, nclip, T = args.videos, args.n_clips, args.frames
    frames = S.synthetic_frames(bv, nclip * T, args.size, 42 + rank)
    vis_all = ops_image_norm_host(frames).to(dev)                 # (Bv, nclip*T, 3, S, S) fp32, mean-subtracted
    ids, mask = S.synthetic_text(bv * args.repeat, args.txt_len, 42 + rank)
    ids, mask = ids.to(dev), mask.to(dev)
    labels = torch.tensor([1, 0] * bv if args.repeat == 2 else [1] * (bv * args.repeat), dtype=torch.long, device=dev)
    counts = [args.repeat] * bv
    vis_clips = vis_all.view(bv, nclip, T, 3, args.size, args.size)

    def forward_loss():
        logits = []
        for c in range(nclip):                                    # clip loop of run_video_retrieval.py:396-401
            batch = dict(visual_inputs=vis_clips[:, c].contiguous() if nclip > 1 else vis_clips[:, 0], text_input_ids=ids,
                         text_input_mask=mask, n_examples_list=list(counts))
            logits.append(model(batch)["logits"])
        lg = logits[0] if nclip == 1 else clips.aggregate_clip_logits(logits, "mean")     # score_agg_func of the JSON config
        _, loss = model.transformer.calc_loss(lg, labels, sample_size=bv)
        return loss.mean()

    def train_step():
        opt.zero_grad()
        loss = forward_loss()
        loss.backward()
        sync.reduce_cnn()
        sync.wait()
        model.rt.seed_dev.add_(1)
        opt.step(grad_scale=sync.grad_scale)
        return loss

    def infer_step():
        with torch.no_grad():
            return forward_loss()

    step_fn = infer_step if args.forward_only else train_step

    log("inputs ready")
    # ---- eager warm-up (also builds pixel tables etc.), then graph capture --------------------------------
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            loss = step_fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    log(f"eager warm-up done, loss {float(loss.item()):.4f}")
    # ---- replay plan -----------------------------------------------------------------------------------------------------
    #   N = 1          : the whole step in one hipGraph.
    #   N > 1 (default): three hipGraphs with EAGER RCCL all-reduces (bf16 on the wire) between them; the transformer bucket
    #                    travels while the ResNet-backward graph runs.  Only this library's kernels are ever captured -- a
    #                    capture that fails on this stack cannot be recovered from inside the process, so collectives stay
    #                    out of it.
    #   CB_BENCH_PLAN=full : N > 1 with the whole step (RCCL calls included) in one hipGraph; the transformer bucket is then
    #                    issued from inside the backward and overlaps the ResNet backward.  Opt-in until it can be tested
    #                    on a multi-GPU box.   CB_BENCH_PLAN=eager / --no-graph: no graphs.
    def capture(fn):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = fn()
        return g, out

    plan_env = os.environ.get("CB_BENCH_PLAN", "")
    run, plan = step_fn, "eager"
    if not args.no_graph and plan_env != "eager":
        if world == 1 or args.forward_only or plan_env == "full":
            g1, loss = capture(step_fn)
            run, plan = g1.replay, "one hipGraph"
        else:
            # [zero + forward + encoder backward] -> transformer bucket (async, crosses xGMI during the next graph) ->
            # [ResNet backward] -> CNN bucket -> wait -> [clip + AdamW].  The backward is cut at the grid features:
            # autograd.grad(loss, grid) runs the heads' and the encoder's backward (parameter gradients land in the flat
            # buffer as a side effect), grid.backward(dgrid) runs the CNN trunk's.
            assert nclip == 1, "the split replay plan handles one clip per step (the headline config)"
            state = {}

            def part_a():
                opt.zero_grad()
                grid = model.grid_features(vis_clips[:, 0])
                out = model.forward_from_grid(dict(visual_inputs=grid, text_input_ids=ids, text_input_mask=mask,
                                                   n_examples_list=list(counts)))
                _, l = model.transformer.calc_loss(out["logits"], labels, sample_size=bv)
                loss_ = l.mean()
                (dgrid,) = torch.autograd.grad(loss_, [grid])
                state["grid"], state["dgrid"] = grid, dgrid
                return loss_

            def part_b():
                state["grid"].backward(state["dgrid"])

            def part_c():
                model.rt.seed_dev.add_(1)
                opt.step(grad_scale=sync.grad_scale)

            model.rt.after_encoder_backward = None                 # collectives are issued between the graphs, eagerly
            ga, loss = capture(part_a)
            gb, _ = capture(part_b)
            gc, _ = capture(part_c)

            def run_split():
                ga.replay()
                sync.reduce_transformer()
                gb.replay()
                sync.reduce_cnn()
                sync.wait()
                gc.replay()
            run, plan = run_split, "three hipGraphs, eager bf16 all-reduces (transformer bucket overlaps the ResNet backward)"
    graph = None if plan == "eager" else True
    log(f"replay plan: {plan}")

    # clock / power-state settling (untimed, before the W warm-up steps): right after process start the first replays run
    # ~15 % slow on some boxes; ~0.75 s of steady replays brings the GPU to its sustained clocks
    # (a FIXED number of replays: with N > 1 every rank must issue the same number of collectives)
    for _ in range(12):
        for _ in range(8):
            run()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        run()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
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
    final_loss = float(loss.item()) if loss is not None else float("nan")
    log(f"timed region done: {ms_per_step:.3f} ms/step, {value:.1f} clips/s")

    out = {
        "metric": "clips/sec/node (2\u00d72 frames, 224px, L_txt=32) at 1/2/4/8 MI355X" if not args.forward_only else "clips/sec/node forward-only (diagnostic)",
        "value": round(value, 2), "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: MSRVTT retrieval training step (fwd+bwd+allreduce+clip+AdamW), "
                               f"{bv} videos x {nclip} clip x {T} frames {args.size}px + {args.repeat} texts L_txt={args.txt_len} per GPU",
                   "videos_per_gpu": bv, "n_clips": nclip, "n_frames": T, "img_size": args.size, "txt_len": args.txt_len,
                   "texts_per_video": args.repeat, "parallelism": f"dp{world}", "hip_graph": graph is not None, "replay_plan": plan,
                   "dropout": not args.forward_only, "final_loss": round(final_loss, 5)},
    }
    if rank == 0 and world == 1 and not args.no_roofline:
        out["roofline"] = measure_roofline(step_fn)
        log("roofline measured")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, args)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def ops_image_norm_host(frames_u8):
    """ImageNorm of the synthetic frames (done once, outside the timed region): uint8 -> fp32 - mean."""
    mean = torch.tensor([123.675, 116.28, 103.53]).view(1, 1, 3, 1, 1)
    return frames_u8.float() - mean


def measure_roofline(step_fn):
    """Durations of the cb_gemm kernels of the step, by kernel family, against the dense bf16 MFMA peak with their
    ALGORITHMIC flops (2*M*N*K per problem).

    HIP events cost several microseconds each on this stack, so bracketing every ~20 us launch individually would
    measure the markers.  Instead one eager step is recorded (every cb_gemm call with its live operands), then each
    family's calls are replayed back to back -- captured in a hipGraph, `reps` times -- between ONE pair of HIP events
    on the launch stream: avg launch duration = elapsed / (reps * launches), which is what rocprofv3's kernel trace of
    the same command reports (profiles/)."""
    from clipbert_amd import ops
    calls = []
    orig = ops.gemm

    def logged(a, b, M, N, K, **kw):
        form = ("wgrad" if kw.get("a_mode", 0) == ops.KROW else ("dgrad" if kw.get("b_mode", 0) in (ops.KROW, ops.KROW_TAPS) else "fwd"))
        conv = kw.get("a_mode", 0) == ops.ROWK_GATHER or kw.get("b_mode", 0) == ops.KROW_GATHER
        key = f"gemm_kernel<bf16> {form}{' (implicit-GEMM conv)' if conv else ''}"
        calls.append((key, 2.0 * M * N * K * kw.get("batch", 1), (a, b, M, N, K), kw))
        return orig(a, b, M, N, K, **kw)

    ops.gemm = logged
    try:
        step_fn()
        torch.cuda.synchronize()
    finally:
        ops.gemm = orig
    fams = {}
    for key, fl, pos, kw in calls:
        fams.setdefault(key, []).append((fl, pos, kw))
    reps, outer = 3, 5
    agg = {}
    for key, lst in fams.items():
        def replay():
            for _ in range(reps):
                for fl, pos, kw in lst:
                    orig(*pos, **kw)
        replay()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            replay()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(outer):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / (reps * outer)             # seconds for one pass over the family's launches
        agg[key] = [sum(x[0] for x in lst), t, len(lst)]
        del g
    tot_fl = sum(a[0] for a in agg.values())
    tot_t = sum(a[1] for a in agg.values())
    dom = max(agg.items(), key=lambda kv: kv[1][1])
    achieved = dom[1][0] / dom[1][1] / 1e12
    # HBM bytes per launch of that kernel family from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
    # separate runs, gfx950 correction applied -- profiles/r01_pmc_traffic.json); null when no PMC data is committed
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as fh:
            traffic = json.load(fh)["families"][dom[0]]["hbm_bytes_per_launch"]
    except Exception:
        traffic = None
    return {"bound": "mfma", "kernel": dom[0], "launches": dom[1][2], "avg_launch_us": round(dom[1][1] / dom[1][2] * 1e6, 2),
            "achieved": round(achieved, 2), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / BF16_MFMA_PEAK_TFLOPS, 4),
            "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC, profiles/r01_pmc_traffic.json)",
            "algorithmic_flop_per_launch": round(dom[1][0] / dom[1][2]),
            "all_gemm_kernels": {"achieved": round(tot_fl / tot_t / 1e12, 2), "time_ms": round(tot_t * 1e3, 3),
                                 "gflop_per_step": round(tot_fl / 1e9, 1)},
            "by_kernel": {k: {"tflops": round(v[0] / v[1] / 1e12, 2), "ms": round(v[1] * 1e3, 3), "launches": v[2]} for k, v in agg.items()}}


def cpu_baseline(cfg, args):
    """The CPU oracle (port of the reference arithmetic in stock PyTorch fp32) on a bounded sample of the
    same workload: 2 videos x 2 frames + 4 texts, forward + backward, this host's cores."""
    from clipbert_amd import synthetic as S
    from oracle import clipbert_oracle as O
    ncores = min(os.cpu_count() or 1, 32)      # more threads than this only adds contention for 4-frame convs
    torch.set_num_threads(ncores)
    sd = S.full_state_dict(cfg, "retrieval", 42)
    sd = {k: v.clone().requires_grad_(v.is_floating_point() and ".norm." not in k and "stem" not in k and "res2" not in k) for k, v in sd.items()}
    nv = 2
    frames = S.synthetic_frames(nv, args.frames, args.size, 42)
    ids, mask = S.synthetic_text(nv * args.repeat, args.txt_len, 42)
    batch = dict(visual_inputs=O.image_norm(frames, S.PIXEL_MEAN, S.PIXEL_STD), text_input_ids=ids, text_input_mask=mask,
                 n_examples_list=[args.repeat] * nv, labels=torch.tensor([1, 0] * nv)[: nv * args.repeat])

    def one():
        out = O.clipbert_forward(sd, batch, cfg, "retrieval")
        out["loss"].mean().backward()

    one()                                   # warm-up
    times = []
    t_budget = time.perf_counter()
    while len(times) < 5 and time.perf_counter() - t_budget < 25:
        t0 = time.perf_counter()
        one()
        times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    return {"value": round(nv / med, 3), "unit": "clips/s", "cores": ncores, "kind": "port",
            "sample": f"{nv} videos x {args.frames} frames {args.size}px + {nv * args.repeat} texts, fwd+bwd (no optimizer), "
                      f"median of {len(times)} iterations, torch {torch.__version__} fp32, {ncores} threads"}


if __name__ == "__main__":
    main()
