#!/usr/bin/env python
"""Diagnostic for SURVEY 8f row N1 (BASELINE configs[4]-style retrieval inference): ONE video of 16 clips x 2 frames 224px
scored against N captions in mini-batches of 64 -- reference order of evaluation (full forward per clip per mini-batch,
run_video_retrieval.py:655-666) vs cached grid features + clips folded into the encoder batch.  Prints pairs/s for both
and the largest score difference.  Not the headline metric (bench.py is)."""
import argparse
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from clipbert_amd import modeling as M, synthetic as S, tasks

ap = argparse.ArgumentParser()
ap.add_argument("--captions", type=int, default=256)
ap.add_argument("--clips", type=int, default=16)
args = ap.parse_args()
dev = torch.device("cuda", 0)
cfg = dict(bench.BASE_CONFIG)
model = M.ClipBert(cfg, detectron2_model_cfg="R-50-grid.yaml", transformer_cls=M.ClipBertForVideoTextRetrieval)
model.load_state_dict(S.full_state_dict(cfg, "retrieval", 42), strict=True)
model.to(dev).eval()
model.prepare(dtype=torch.bfloat16, device=dev)
icfg = SimpleNamespace(inference_n_clips=args.clips, num_frm=2, score_agg_func="lse", inference_batch_size=64)
vis = bench.ops_image_norm_host(S.synthetic_frames(1, args.clips * 2, 224, 42)).to(dev)
ids, mask = S.synthetic_text(args.captions, 32, 42)
ids, mask = ids.to(dev), mask.to(dev)
res = {}
for name, cache in (("reference order", False), ("cached + folded (N1)", True)):
    tasks.inference_retrieval_video(model, vis, ids[:64], mask[:64], icfg, cache_cnn=cache)       # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res[name] = tasks.inference_retrieval_video(model, vis, ids, mask, icfg, cache_cnn=cache)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pairs = args.captions * args.clips
    print(f"{name:22s}: {dt * 1e3:8.1f} ms for {args.captions} captions x {args.clips} clips = {pairs / dt:9.0f} (clip, caption) pairs/s", flush=True)
a, b = res["reference order"], res["cached + folded (N1)"]
print("max |score difference| =", max(abs(x - y) for x, y in zip(a, b)))
