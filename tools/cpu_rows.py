#!/usr/bin/env python
"""the CPU oracle (oracle/clipbert_oracle.py -- test infrastructure, a stock-PyTorch fp32
restatement of the reference; the reference itself cannot be imported where detectron2 is absent) timed on THIS host's cores on
bounded samples of each BASELINE config.  Prints a markdown table (commit it under profiles/ next to the MI355X rows).

    python tools/cpu_rows.py [--threads N] [--budget-s 12] > profiles/rNN_cpu_rows_<host>.md

Protocol: torch.set_num_threads(N); seed 42; synthetic inputs as SURVEY 8(d); 1 warm-up + up to 10 timed iterations
within the per-row time budget; median."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from clipbert_amd import synthetic as S          # noqa: E402
from oracle import clipbert_oracle as O          # noqa: E402


def timed(fn, budget_s, grad):
    ctx = torch.enable_grad() if grad else torch.no_grad()
    with ctx:
        fn()
        ts, t_start = [], time.perf_counter()
        while len(ts) < 10 and (time.perf_counter() - t_start < budget_s or not ts):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2], len(ts)


def state(cfg, head, train):
    sd = S.full_state_dict(cfg, head, 42)
    return {k: v.clone().requires_grad_(train and v.is_floating_point() and ".norm." not in k and "stem" not in k and "res2" not in k)
            for k, v in sd.items()}


def clip_loop(sd, cfg, head, vis, ids, mask, rep, pool, labels, train, mc=False):
    nv, nclip = vis.shape[0], vis.shape[1]

    def one():
        per_clip = [O.clipbert_forward(sd, dict(visual_inputs=vis[:, c], text_input_ids=ids, text_input_mask=mask, n_examples_list=[rep] * nv),
                                       cfg, head)["logits"] for c in range(nclip)]
        pooled = O.aggregate_clip_logits(per_clip, pool)
        if not train:
            return
        if pool == "lse":
            loss = O.lse_train_loss(pooled, labels).mean()
        else:
            loss = torch.nn.functional.cross_entropy(pooled.view(-1, rep) if mc else pooled, labels)
        loss.backward()
    return one


def inputs(nv, nclip, T, size, rep, lt):
    frames = S.synthetic_frames(nv, nclip * T, size, 42)
    ids, mask = S.synthetic_text(nv * rep, lt, 42)
    vis = O.image_norm(frames, S.PIXEL_MEAN, S.PIXEL_STD).view(nv, nclip, T, 3, size, size)
    return vis, ids, mask


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=min(os.cpu_count() or 1, 32))
    ap.add_argument("--budget-s", type=float, default=12.0)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    torch.manual_seed(42)
    rows = []
    base = dict(O.BASE_CONFIG)

    # C1: configs[0] -- ClipBertForPreTraining ITM + MLM forward, 2 images 224 px (n_frm = 1), 20-token captions
    cfg = dict(base)
    sd = state(cfg, "pretraining", False)
    vis, ids, mask = inputs(2, 1, 1, 224, 1, 20)
    b = dict(visual_inputs=vis[:, 0], text_input_ids=ids, text_input_mask=mask, n_examples_list=[1, 1])
    t, n = timed(lambda: O.clipbert_forward(sd, dict(b), cfg, "pretraining"), args.budget_s, False)
    rows.append(("C1", "configs[0]: ClipBertForPreTraining ITM+MLM forward, B = 2 images 224 px, L_txt 20", f"{t * 1e3:.0f} ms / forward", f"{2 / t:.1f} images/s", n))

    # C2: the headline clip -- T = 2 frames 224 px, L_txt 32, retrieval head, r = 1: forward and forward + backward
    cfg = dict(base, num_labels=2, loss_type="ce", margin=0.1)
    for train in (False, True):
        sd = state(cfg, "retrieval", train)
        vis, ids, mask = inputs(2, 1, 2, 224, 1, 32)
        fn = clip_loop(sd, cfg, "retrieval", vis, ids, mask, 1, "mean", torch.tensor([1, 0]), train)
        t, n = timed(fn, args.budget_s, train)
        rows.append(("C2", f"headline clip: 2 clips of 2 frames 224 px + 1 text L_txt 32 each, retrieval head, {'forward + backward' if train else 'forward'}",
                     f"{t * 1e3:.0f} ms", f"{2 / t:.2f} clips/s", n))

    # C3: configs[1]/[2] -- MSRVTT retrieval, r = 2 (pos + neg), N_clip = 1 and N_clip = 4 (LSE), forward + backward, 2 videos
    for nclip, pool in ((1, "mean"), (4, "lse")):
        sd = state(cfg, "retrieval", True)
        vis, ids, mask = inputs(2, nclip, 2, 224, 2, 32)
        fn = clip_loop(sd, cfg, "retrieval", vis, ids, mask, 2, pool, torch.tensor([1, 0, 1, 0]), True)
        t, n = timed(fn, args.budget_s, True)
        rows.append(("C3", f"configs[{1 if nclip == 1 else 2}]: MSRVTT retrieval, 2 videos x N_clip {nclip} x 2 frames 224 px, r = 2, {pool} pooling, forward + backward (clip loop)",
                     f"{t * 1e3:.0f} ms", f"{2 * nclip / t:.2f} clips/s", n))

    # C4: configs[3] -- TGIF-QA action, multiple-choice head, 5 options L_txt 25, N_clip 2, forward (answer ids) and forward + backward
    cfg4 = dict(base, num_labels=5, loss_type="ce")
    for train in (False, True):
        sd = state(cfg4, "multiple_choice", train)
        vis, ids, mask = inputs(2, 2, 2, 224, 5, 25)
        fn = clip_loop(sd, cfg4, "multiple_choice", vis, ids, mask, 5, "mean", S.synthetic_labels(2, 5, 42), train, mc=True)
        t, n = timed(fn, args.budget_s, train)
        rows.append(("C4", f"configs[3]: TGIF-QA action, 2 videos x N_clip 2 x 2 frames 224 px, 5 options L_txt 25, {'forward + backward' if train else 'forward'}",
                     f"{t * 1e3:.0f} ms", f"{4 / t:.2f} clips/s", n))

    # C5: configs[4] -- retrieval inference, no grad: 1 video x 4 of its 16 clips against 16 of a 64-caption mini-batch (reference order:
    # the CNN runs again for every caption mini-batch, run_video_retrieval.py:655-666)
    sd = state(cfg, "retrieval", False)
    vis, ids, mask = inputs(1, 4, 2, 224, 16, 32)
    fn = clip_loop(sd, cfg, "retrieval", vis, ids, mask, 16, "lse", None, False)
    t, n = timed(fn, args.budget_s, False)
    rows.append(("C5", "configs[4]: retrieval inference, 1 video x 4 clips x 2 frames 224 px against 16 captions L_txt 32 (sample of 16 clips x 64), no grad",
                 f"{t * 1e3:.0f} ms", f"{4 * 16 / t:.1f} (clip, caption) pairs/s = {4 / t:.2f} clips/s", n))

    cpu = "?"
    try:
        cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    print(f"# BASELINE.md rows C1-C5 on the CPU oracle: {args.threads} threads of {os.cpu_count()} logical cores ({cpu}), torch {torch.__version__} fp32\n")
    print("Bounded samples of each BASELINE config (sizes in the row), reference order of evaluation (clip LOOP, CNN per mini-batch); median of n timed")
    print("iterations after one warm-up.  `kind: port` -- the oracle restates the reference (oracle/clipbert_oracle.py); no optimizer step.\n")
    print("| row | workload (sample) | time / iteration | throughput | n |")
    print("|---|---|---:|---:|---:|")
    for r in rows:
        print("| " + " | ".join(str(x) for x in r) + " |")


if __name__ == "__main__":
    main()
