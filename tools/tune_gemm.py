#!/usr/bin/env python
"""Measure every cb_gemm problem of the benchmark steps under each launch configuration (tile x workgroup order) on the GPU.

  python tools/tune_gemm.py [--modes train,tgif,infer16] [--out gpurun_out/gemm_tuning.json]

One eager step of each bench mode is run with ops.gemm logged (as bench.py's roofline does); unique problems
(form, M, N, K, batch, split, taps, epilogue kind) are then replayed per configuration: 16 back-to-back launches captured in
a hipGraph, replayed 4 times between one pair of HIP events, best of 3 -> microseconds per launch.  The JSON holds all
timings; tools/gen_tuned.py turns the winners into clipbert_amd/csrc/gemm_tuned.h.
"""
import argparse
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

TILES = {1: "128x128", 2: "64x64", 3: "128x64", 4: "128x128o2", 5: "8w256x256", 6: "8w128x256", 7: "8w256x128"}
TILE_DIM = {5: (256, 256), 6: (128, 256), 7: (256, 128)}


def splits_for(tile, M, N, K, batch):
    """K splits worth trying for an 8-wave tile: unsplit, and the splits that bring the grid to ~1x / ~2x the 256 CUs"""
    bm, bn = TILE_DIM[tile]
    tiles = -(-M // bm) * -(-N // bn) * batch
    kt = -(-K // 64)
    cand = {1}
    for target in (256, 512):
        s = max(1, round(target / tiles))
        if s > 1 and kt // s >= 2 and s * batch * M * N * 4 <= (128 << 20):
            cand.add(s)
    return sorted(cand)


def record_calls(mode):
    """bench.py --mode <mode> in-process up to the first eager step, with ops.gemm logged.  ``mode`` may carry extra bench flags
    after a colon, e.g. "train:--size 448 --txt-len 20 --n-clips 4" (the JSON config's native sizes)."""
    import importlib
    bench = importlib.import_module("bench")
    from clipbert_amd import ops
    calls = []
    orig = ops.gemm

    def logged(a, b, M, N, K, **kw):
        calls.append(((a, b, M, N, K), dict(kw)))
        return orig(a, b, M, N, K, **kw)

    class Stop(Exception):
        pass

    ops.gemm = logged
    real_log = bench.log

    def log_hook(msg):
        real_log(msg)
        if msg.startswith("eager warm-up done"):
            raise Stop()

    bench.log = log_hook
    argv = sys.argv
    extra = mode.split(":", 1)[1].split() if ":" in mode else []
    sys.argv = ["bench.py", "--mode", mode.split(":", 1)[0], "--no-cpu-baseline", "--no-roofline"] + extra
    try:
        bench.main()
    except Stop:
        pass
    finally:
        sys.argv = argv
        ops.gemm = orig
        bench.log = real_log
    n = len(calls) // 2                     # the warm-up runs the step twice: keep one step's calls
    return calls[n:]


def key_of(pos, kw):
    from clipbert_amd import ops
    a, b, M, N, K = pos
    form = "wgrad" if kw.get("a_mode", 0) == ops.KROW else ("dgrad" if kw.get("b_mode", 0) in (ops.KROW, ops.KROW_TAPS) else "fwd")
    taps = kw.get("R", 1) * kw.get("S", 1)
    epi = "+".join(k for k in ("scale", "shift", "residual", "mask", "out2", "gelu_grad_pre", "a_rowsum", "c_rowmap") if kw.get(k) is not None)
    if kw.get("act", 0):
        epi += f"+act{kw['act']}"
    if kw.get("dropout_p", 0) > 0:
        epi += "+drop"
    if kw.get("relu_bwd"):
        epi += "+relu_bwd"
    return (form, kw.get("a_mode", 0), kw.get("b_mode", 0), M, N, K, kw.get("batch", 1), kw.get("split_k", 1), taps, epi,
            str(a.dtype).replace("torch.", ""))


_COLD = {}


def cold_state(pos):
    """--cold: what precedes every timed launch -- a 384 MB memset (evicts L2 and the 256 MB Infinity Cache: weights and the output
    lines come from / go to HBM as in the step, where ~3 GB stream between two uses of anything) followed by a fresh write of the
    A operand (in the step the activations were produced by the kernel just before)."""
    if "flush" not in _COLD:
        _COLD["flush"] = torch.empty(384 << 20, dtype=torch.uint8, device="cuda")
    a = pos[0]
    key = (a.data_ptr(), tuple(a.shape), tuple(a.stride()))
    if _COLD.get("key") != key:
        _COLD["key"], _COLD["a_src"] = key, a.detach().clone()
        _COLD.pop("base", None)
    return _COLD["flush"], _COLD["a_src"]


def time_config(pos, kw, tile, xcd, inner=16, outer=4, best_of=3, split=None, sched=0):
    if COLD:
        return time_config_cold(pos, kw, tile, xcd, split=split, sched=sched)
    from clipbert_amd import ops
    kw = dict(kw, tile=tile, xcd_order=xcd, schedule=sched)
    if split is not None:
        kw["split_k"] = split

    def burst():
        for _ in range(inner):
            ops.gemm(*pos, **kw)
    try:
        burst()
        torch.cuda.synchronize()
    except Exception as e:                                 # configuration not available for this problem
        return None, str(e)[:120]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        burst()
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
    return best, None


COLD = False


def _graph_time(fn, inner, outer, best_of):
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


def time_config_cold(pos, kw, tile, xcd, split=None, sched=0, inner=8, outer=2, best_of=3):
    """microseconds of one launch issued with cold caches: ([flush, A rewrite, launch] - [flush, A rewrite]) per iteration"""
    from clipbert_amd import ops
    kw = dict(kw, tile=tile, xcd_order=xcd, schedule=sched)
    if split is not None:
        kw["split_k"] = split
    flush, a_src = cold_state(pos)
    a = pos[0]

    def pre():
        with torch.no_grad():
            flush.zero_()
            a.detach().copy_(a_src)

    def full():
        pre()
        ops.gemm(*pos, **kw)
    try:
        ops.gemm(*pos, **kw)
        torch.cuda.synchronize()
    except Exception as e:                                 # noqa: BLE001
        return None, str(e)[:120]
    if "base" not in _COLD:
        _COLD["base"] = _graph_time(pre, inner, outer, best_of)
    return max(0.1, _graph_time(full, inner, outer, best_of) - _COLD["base"]), None


def main():
    global COLD
    ap = argparse.ArgumentParser()
    ap.add_argument("--cold", action="store_true", help="time every launch behind a cache flush (see cold_state)")
    ap.add_argument("--modes", default="train")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "gemm_tuning.json"))
    ap.add_argument("--tiles", default="1,2,3,4,5,6,7")
    ap.add_argument("--sched", default="1,3", help="K-loop schedules of the 8-wave tiles to sweep (cb_gemm_desc.schedule values)")
    args = ap.parse_args()
    COLD = args.cold
    tiles = [int(t) for t in args.tiles.split(",") if int(t) <= 4]
    tiles8 = [int(t) for t in args.tiles.split(",") if int(t) >= 5]
    scheds = [int(t) for t in args.sched.split(",")]
    os.environ["CB_GEMM_NO_TUNED"] = "1"                  # the recorded calls carry tile = 0: measure the heuristics as "auto"
    problems = {}
    for mode in args.modes.split(";" if ";" in args.modes or ":" in args.modes else ","):
        calls = record_calls(mode)
        print(f"[tune] mode {mode}: {len(calls)} cb_gemm calls per step", file=sys.stderr, flush=True)
        for pos, kw in calls:
            k = key_of(pos, kw)
            ent = problems.setdefault(k, dict(pos=pos, kw=kw, count={}, key=k))
            ent["count"][mode] = ent["count"].get(mode, 0) + 1
    print(f"[tune] {len(problems)} unique problems", file=sys.stderr, flush=True)
    out = []
    for k, ent in problems.items():
        pos, kw = ent["pos"], ent["kw"]
        if k[10] != "bfloat16":
            continue
        fixed_tile = kw.get("tile", 0)
        res = {}
        auto, err = time_config(pos, {kk: v for kk, v in kw.items() if kk not in ("tile", "xcd_order")} | {"tile": fixed_tile}, fixed_tile, 0)
        res["auto"] = auto
        for t in tiles:
            for xcd in (2, 1):
                base = {kk: v for kk, v in kw.items() if kk not in ("tile", "xcd_order")}
                us, err = time_config(pos, base, t, xcd)
                res[f"{TILES[t]}/{'xcd' if xcd == 1 else 'rr'}"] = us if us is not None else None
        # weight-gradient form (fp32 C, accumulate): the K split is free to choose as well
        if k[0] == "wgrad" and k[6] == 1 and kw.get("out") is not None and kw["out"].dtype == torch.float32 and kw.get("accumulate"):
            ktiles = (k[5] + 63) // 64
            for sp in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48):
                if sp == k[7] or sp > max(1, ktiles // 4):
                    continue
                for t in tiles:
                    base = {kk: v for kk, v in kw.items() if kk not in ("tile", "xcd_order", "split_k")}
                    us, err = time_config(pos, base, t, 1, split=sp)
                    res[f"{TILES[t]}/xcd/s{sp}"] = us if us is not None else None
        # 8-wave LDS-DMA tiles: any form; their K split goes through workspace slabs (any epilogue)
        for t in tiles8:
            for sp in splits_for(t, k[3], k[4], k[5], k[6]):
                for sc in scheds:
                    base = {kk: v for kk, v in kw.items() if kk not in ("tile", "xcd_order", "split_k", "schedule")}
                    us, err = time_config(pos, base, t, 1, split=sp, sched=sc)
                    res[f"{TILES[t]}/xcd/s{sp}/m{sc - 1}"] = us if us is not None else None
        flops = 2.0 * k[3] * k[4] * k[5] * k[6]
        good = {c: v for c, v in res.items() if v is not None and c != "auto"}
        best = min(good, key=good.get)
        row = dict(form=k[0], a_mode=k[1], b_mode=k[2], M=k[3], N=k[4], K=k[5], batch=k[6], split_k=k[7], taps=k[8], epilogue=k[9],
                   count=ent["count"], caller_tile=fixed_tile, us=res, best=best, best_us=good[best],
                   best_tflops=round(flops / good[best] / 1e6, 1), auto_tflops=round(flops / res["auto"] / 1e6, 1) if res["auto"] else None)
        out.append(row)
        print(f"[tune] {k[0]:5s} M={k[3]:6d} N={k[4]:5d} K={k[5]:5d} b={k[6]:2d} s={k[7]:2d} taps={k[8]} auto {res['auto']:.1f} us  best {best} "
              f"{good[best]:.1f} us ({row['best_tflops']} TF)", file=sys.stderr, flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as fh:
        json.dump(dict(device=torch.cuda.get_device_name(0), problems=out), fh, indent=1)
    tot_auto = sum(sum(r["count"].values()) * (r["us"]["auto"] or 0) for r in out)
    tot_best = sum(sum(r["count"].values()) * r["best_us"] for r in out)
    print(f"[tune] sum over all recorded launches: auto {tot_auto / 1e3:.3f} ms, best-per-shape {tot_best / 1e3:.3f} ms", file=sys.stderr)


if __name__ == "__main__":
    main()
