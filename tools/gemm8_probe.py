#!/usr/bin/env python
"""Probe of the 8-wave cb_gemm kernels (tiles 5 = 256x256, 6 = 128x256, 7 = 256x128) against the 4-wave ones on an MI355X.

  python tools/gemm8_probe.py [--out gpurun_out/gemm8_probe.json] [--quick]

Per problem shape (the encoder GEMMs of the metric step, representative ResNet shapes as plain GEMMs, two large squares) and
per configuration (tile x K-loop schedule x K split): result checked against torch fp32 matmul, then 16 back-to-back launches in
a hipGraph, 4 replays between one pair of HIP events, best of 3 -- all configurations of a shape interleaved in one process
(cdna_hip_programming.md 5.4 rule 24).  Random operands (uniform [-1, 1)), never zeros (rule 25)."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from clipbert_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
TILE_DIM = {5: (256, 256), 6: (128, 256), 7: (256, 128), 1: (128, 128), 2: (64, 64), 3: (128, 64), 4: (128, 128)}
TILE_NAME = {0: "auto", 1: "128x128", 2: "64x64", 3: "128x64", 4: "128x128o2", 5: "8w256x256", 6: "8w128x256", 7: "8w256x128"}


def uni(*shape):
    return (torch.rand(*shape, device=dev) * 2 - 1).to(torch.bfloat16)


def make(form, M, N, K, batch):
    """operands + reference + launcher(tile, sched, split) of one problem"""
    ws = ops.splitk_workspace(dev)
    if form == "fwd":
        a, b = uni(M, K), uni(N, K)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        bias = torch.rand(N, device=dev)
        ref = lambda: a.float() @ b.float().t() + bias
        run = lambda tile, sched, split: ops.gemm(a, b, M, N, K, out=out, shift=bias, tile=tile, schedule=sched, split_k=split, splitk_ws=ws)
        return out, ref, run
    if form == "dgrad":                                   # dX[M,N] = g[M,K] W[K,N]
        a, b = uni(M, K), uni(K, N)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        ref = lambda: a.float() @ b.float()
        run = lambda tile, sched, split: ops.gemm(a, b, M, N, K, out=out, b_mode=ops.KROW, ldb=N, tile=tile, schedule=sched, split_k=split, splitk_ws=ws)
        return out, ref, run
    # wgrad dW[M,N] = g[K,M]^T x[K,N] (fp32 out, overwritten), optionally `batch` layers in one launch
    a, b = uni(batch, K, M), uni(batch, K, N)
    out = torch.zeros(batch, M, N, dtype=torch.float32, device=dev)
    rs = torch.zeros(batch, M, dtype=torch.float32, device=dev)
    ref = lambda: torch.einsum("bkm,bkn->bmn", a.float(), b.float())

    def run(tile, sched, split):
        kw = dict(batch=batch, batch_strides=(K * M, K * N, M * N, M)) if batch > 1 else {}
        # (split > 1 on the 4-wave tiles = fp32 atomics onto C: legal for this form; C accumulates across launches -- timing only)
        ops.gemm(a, b, M, N, K, out=out, a_mode=ops.KROW, b_mode=ops.KROW, lda=M, ldb=N, ldc=N, accumulate=(tile < 5 and split > 1),
                 tile=tile, schedule=sched, split_k=split, splitk_ws=ws, a_rowsum=rs, **kw)
    return out, ref, run


def time_fn(fn, inner=16, outer=4, best_of=3):
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
    del g
    return best


def splits_for(tile, M, N, K, batch):
    bm, bn = TILE_DIM[tile]
    tiles = -(-M // bm) * -(-N // bn) * batch
    kt = -(-K // 64)
    cand = {1}
    for target in (256, 512):
        s = max(1, round(target / tiles))
        if s > 1 and kt // s >= 3:
            cand.add(s)
    return sorted(cand)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "gemm8_probe.json"))
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    shapes = [
        ("fwd", 2624, 3072, 768, 1), ("fwd", 2624, 2304, 768, 1), ("fwd", 2624, 768, 3072, 1), ("fwd", 2624, 768, 768, 1),
        ("dgrad", 2624, 768, 3072, 1), ("dgrad", 2624, 3072, 768, 1), ("dgrad", 2624, 768, 2304, 1),
        ("wgrad", 3072, 768, 2624, 12), ("wgrad", 768, 3072, 2624, 12), ("wgrad", 2304, 768, 2624, 12), ("wgrad", 768, 768, 2624, 12),
        ("fwd", 12544, 256, 2304, 1), ("fwd", 50176, 128, 1152, 1), ("fwd", 3136, 512, 4608, 1), ("fwd", 3136, 768, 18432, 1),
        ("fwd", 12544, 1024, 256, 1), ("fwd", 12544, 256, 1024, 1), ("fwd", 50176, 512, 128, 1), ("fwd", 3136, 2048, 512, 1),
        ("dgrad", 12544, 256, 1024, 1), ("dgrad", 50176, 128, 512, 1), ("wgrad", 768, 18432, 3136, 1), ("wgrad", 1024, 256, 12544, 1),
        ("fwd", 8192, 8192, 1024, 1), ("fwd", 4096, 4096, 4096, 1), ("dgrad", 8192, 8192, 1024, 1), ("wgrad", 4096, 4096, 4096, 1),
    ]
    if args.quick:
        shapes = shapes[:3] + shapes[7:8] + shapes[11:12] + shapes[14:15] + shapes[-4:-2] + shapes[-1:]
    rows = []
    for form, M, N, K, batch in shapes:
        out, ref, run = make(form, M, N, K, batch)
        want = ref()
        scale = float(want.abs().max())
        cfgs = [(t, 0, 1) for t in (0, 1, 2, 3, 4)]
        if form == "wgrad" and batch == 1:
            cfgs += [(t, 0, s) for t in (2, 4) for s in (4, 8)]
        for t in (5, 6, 7):
            for s in splits_for(t, M, N, K, batch):
                cfgs += [(t, sched, s) for sched in (1, 2, 3, 4)]          # 4 = the persistent kernel (name .../m3)
        res, bad = {}, []
        for t, sched, split in cfgs:
            name = TILE_NAME[t] + (f"/m{sched - 1}" if t >= 5 else "") + (f"/s{split}" if split > 1 else "")
            try:
                if t < 5 and split > 1:
                    out.zero_()
                run(t, sched, split)
                torch.cuda.synchronize()
                err = float((out.float().view_as(want) - want).abs().max()) / max(scale, 1e-6)
                if err > 2e-2:
                    bad.append((name, err))
                res[name] = time_fn(lambda: run(t, sched, split))
            except Exception as e:                             # noqa: BLE001
                res[name] = None
                bad.append((name, str(e)[:100]))
        flops = 2.0 * M * N * K * batch
        good = {c: v for c, v in res.items() if v}
        best = min(good, key=good.get)
        best4 = min((c for c in good if not c.startswith("8w")), key=good.get)
        best8 = min((c for c in good if c.startswith("8w")), key=good.get, default=None)
        rows.append(dict(form=form, M=M, N=N, K=K, batch=batch, us=res, best=best, best4=best4, best8=best8, bad=bad,
                         tflops={c: round(flops / v / 1e6, 1) for c, v in good.items()}))
        print(f"[probe] {form:5s} M={M:6d} N={N:5d} K={K:5d} b={batch:2d}: 4-wave best {best4} {good[best4]:.1f} us ({flops / good[best4] / 1e6:.0f} TF) | "
              f"8-wave best {best8} {good.get(best8, 0):.1f} us ({flops / good[best8] / 1e6 if best8 else 0:.0f} TF)" + (f"  BAD {bad}" if bad else ""),
              file=sys.stderr, flush=True)
        del out, ref, run, want
        torch.cuda.empty_cache()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as fh:
        json.dump(dict(device=torch.cuda.get_device_name(0), rows=rows), fh, indent=1)
    # schedule comparison over all shapes: total time of the best split per (tile, schedule)
    for sched in (0, 1, 2, 3):
        tot = 0.0
        for r in rows:
            c = [v for k, v in r["us"].items() if k.startswith("8w") and f"/m{sched}" in k and v]
            tot += min(c) if c else 0.0
        print(f"[probe] schedule {sched}: sum of per-shape best 8-wave times {tot / 1e3:.3f} ms", file=sys.stderr)


if __name__ == "__main__":
    main()
