"""cb_gemm_group on the weight gradients of one ResNet stage at the metric step's shapes (64 frames 224 px): the ungrouped launches
(cb_gemm, tuned table) against the grouped ones under every (tile, K split) and under the library's own choice.  Timing: hipGraph of
REP passes, every pass behind a 512 MB cache flush (the state the launches meet in the step), flush time subtracted.
    python tools/group_probe.py [--frames 64] [--out gpurun_out/group_probe.json]"""
import argparse
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clipbert_amd import ops  # noqa: E402

STAGES = {  # name: (blocks, mid, cout, cin of the stage, output H = W, stride of the first block)
    "res3": (4, 128, 512, 256, 28, 2), "res4": (6, 256, 1024, 512, 14, 2), "res5": (3, 512, 2048, 1024, 7, 2)}


def stage_problems(name, frames, dev):
    nb, mid, cout, cin0, hw, stride = STAGES[name]
    dt = torch.bfloat16
    m = frames * hw * hw
    probs = []

    def t(*shape):
        return torch.randn(*shape, device=dev, dtype=dt) * 0.1

    for b in range(nb):
        cin = cin0 if b == 0 else cout
        g3, y2 = t(m, cout), t(m, mid)
        probs.append(dict(g=g3, x=y2, cout=cout, cin=mid, k=1))                           # conv3
        g2, y1 = t(m, mid), t(frames, hw, hw, mid)
        probs.append(dict(g=g2, x=y1, cout=mid, cin=mid, k=3, H=hw, W=hw, s=1, p=1))       # conv2 (3x3)
        g1 = t(m, mid)
        if b == 0:
            x = t(frames, hw * stride, hw * stride, cin)
            probs.append(dict(g=g1, x=x, cout=mid, cin=cin, k=1, H=hw * stride, W=hw * stride, s=stride, p=0, gather=True))   # conv1, strided
            probs.append(dict(g=t(m, cout), x=x, cout=cout, cin=cin, k=1, H=hw * stride, W=hw * stride, s=stride, p=0, gather=True))  # shortcut
        else:
            probs.append(dict(g=g1, x=t(m, cin), cout=mid, cin=cin, k=1))                 # conv1
    for pr in probs:
        kk = pr["k"] ** 2 * pr["cin"]
        pr["out"] = torch.zeros(pr["cout"], kk, device=dev)
        if pr["k"] == 3 or pr.get("gather"):
            oh = hw
            pr["tab"] = ops.build_pixel_table(frames, oh, oh, pr["s"], pr["p"], pr["H"] * pr["W"] * pr["cin"], pr["W"] * pr["cin"], pr["cin"], dev)
    return probs, m


def desc(pr, m, tile=0, split=1):
    kk = pr["k"] ** 2 * pr["cin"]
    if "tab" in pr:
        return ops.gemm_desc(pr["g"], pr["x"], pr["cout"], kk, m, out=pr["out"], a_mode=ops.KROW, lda=pr["cout"], b_mode=ops.KROW_GATHER,
                             b_tab=pr["tab"], ldb=0, R=pr["k"], S=pr["k"], Cin=pr["cin"], H=pr["H"], W=pr["W"], sH=pr["W"] * pr["cin"],
                             sW=pr["cin"], accumulate=True, split_k=split, tile=tile)
    return ops.gemm_desc(pr["g"], pr["x"].view(m, pr["cin"]), pr["cout"], kk, m, out=pr["out"], a_mode=ops.KROW, lda=pr["cout"], b_mode=ops.KROW,
                         ldb=pr["cin"], accumulate=True, split_k=split, tile=tile)


def timed(fn, flush, rep=6):
    def body(with_work):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(rep):
                flush.add_(1.0)
                if with_work:
                    fn()
        g.replay()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1000.0 / rep)
        return best
    fn(); torch.cuda.synchronize()
    return body(True) - body(False)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--out", default="gpurun_out/group_probe.json")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    flush = torch.zeros(128 << 20, device=dev)                       # 512 MB read-modify-write: nothing of the operands stays cached
    import math
    from clipbert_amd.modeling import _pick_split
    res = {}
    for name in STAGES:
        probs, m = stage_problems(name, args.frames, dev)
        flop = sum(2.0 * m * p["cout"] * p["k"] ** 2 * p["cin"] for p in probs)
        row = {"gflop": flop / 1e9, "problems": len(probs)}

        def single():
            for pr in probs:
                kk = pr["k"] ** 2 * pr["cin"]
                d = desc(pr, m, 0, _pick_split(pr["cout"], kk, m)[0])
                ops.gemm_group([d], pr["out"])
        row["single_us"] = timed(single, flush)
        row["auto_us"] = timed(lambda: ops.gemm_group([desc(pr, m) for pr in probs], probs[0]["out"]), flush)
        kt = (m + 63) // 64
        for tile in (2, 4):
            for s in (1, 2, 3, 4, 6, 8, 12, 16, 24):
                if kt // s < 4:
                    continue
                us = timed(lambda: ops.gemm_group([desc(pr, m, tile, s) for pr in probs], probs[0]["out"]), flush)
                row[f"tile{tile}_s{s}_us"] = us
        best = min((v, k) for k, v in row.items() if k.endswith("_us"))
        row["best"] = best[1]
        row["best_tflops"] = flop / best[0] / 1e6
        row["single_tflops"] = flop / row["single_us"] / 1e6
        row["auto_tflops"] = flop / row["auto_us"] / 1e6
        res[name] = row
        print(name, json.dumps({k: (round(v, 1) if isinstance(v, float) else v) for k, v in row.items()}), flush=True)
        del probs
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
