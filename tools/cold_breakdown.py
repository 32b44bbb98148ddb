#!/usr/bin/env python
"""What makes a GEMM of the step slower than the same launch with hot caches?  For a few encoder / ResNet shapes one launch is timed
behind (a) nothing (hot: back-to-back launches), (b) a 384 MB flush of L2 + Infinity Cache ("all cold"), and behind the flush followed
by a touch (a read pass) of one operand at a time: (c) weights warm again, (d) A warm again, (e) output lines warm again, (f) A and
weights warm.  The differences say which operand's coldness costs what -- i.e. whether prefetching the NEXT launch's weights from the
tail of the current one could pay.     python tools/cold_breakdown.py [--out gpurun_out/cold_breakdown.json]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from clipbert_amd import ops  # noqa: E402

DEV = torch.device("cuda", 0)


def graph_time(fn, inner=8, outer=3, best_of=3):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(inner):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(best_of):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(outer):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / (inner * outer))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "cold_breakdown.json"))
    args = ap.parse_args()
    flush = torch.empty(384 << 20, dtype=torch.uint8, device=DEV)
    sink = torch.zeros(1, dtype=torch.float32, device=DEV)
    shapes = [("FFN1 fwd", 2624, 3072, 768), ("QKV fwd", 2624, 2304, 768), ("FFN2 fwd", 2624, 768, 3072), ("attn-out fwd", 2624, 768, 768),
              ("res4 conv1 fwd", 12544, 256, 1024), ("res4 conv3 fwd", 12544, 1024, 256), ("res5 conv1 fwd", 3136, 512, 2048)]
    rows = []
    for name, M, N, K in shapes:
        a = torch.randn(M, K, device=DEV).bfloat16()
        w = (torch.randn(N, K, device=DEV) * 0.05).bfloat16()
        bias = torch.zeros(N, device=DEV)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)

        def launch():
            ops.gemm(a, w, M, N, K, out=out, shift=bias)

        def touch(t):                      # a read pass: brings the lines back into the Infinity Cache (and partly L2)
            sink.add_(t.view(torch.int16).sum(dtype=torch.float32) * 0.0)

        def pre(warm):
            flush.zero_()
            for t in warm:
                touch(t)
        res = {"shape": name, "M": M, "N": N, "K": K, "gflop": round(2e-9 * M * N * K, 2)}
        res["hot_us"] = round(graph_time(launch, inner=16), 2)
        for key, warm in (("all_cold", ()), ("weights_warm", (w,)), ("a_warm", (a,)), ("out_warm", (out,)), ("a_and_weights_warm", (a, w)),
                          ("all_warm_after_flush", (a, w, out))):
            base = graph_time(lambda: pre(warm))
            full = graph_time(lambda: (pre(warm), launch()))
            res[key + "_us"] = round(max(0.1, full - base), 2)
        rows.append(res)
        print(res, flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(rows, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
