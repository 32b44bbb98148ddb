#!/usr/bin/env python
"""profiles/r02_gemm_tuning.json -> profiles/r02_gemm_by_shape.md: every cb_gemm problem of one bench mode by shape, with the fastest
measured configuration (what csrc/gemm_tuned.h selects), sorted by total time per step.
    python tools/gemm_by_shape.py [tuning.json] [mode] > profiles/r02_gemm_by_shape.md"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r02_gemm_tuning.json")
mode = sys.argv[2] if len(sys.argv) > 2 else "train"
rows = []
for e in json.load(open(src))["problems"]:
    n = e["count"].get(mode, 0)
    if not n:
        continue
    flop = 2.0 * e["M"] * e["N"] * e["K"] * e["batch"]
    rows.append((n * e["best_us"], e, n, flop))
rows.sort(key=lambda r: -r[0])
tot_us = sum(r[0] for r in rows)
tot_flop = sum(r[2] * r[3] for r in rows)
print(f"# Round 2: every cb_gemm problem of the `{mode}` bench step by shape (MI355X)\n")
print(f"Source: `{os.path.relpath(src, ROOT)}` (`tools/tune_gemm.py`: 16 back-to-back launches in a hipGraph, best of 3; operands warm in L2 / Infinity Cache).")
print("M, N = output rows / cols, K = reduction (conv: taps x Cin); b = strided-batched problems per launch; s = caller's K split; n = launches per step;")
print("`best` = fastest configuration (tile / workgroup order [/ K split]) -- what `csrc/gemm_tuned.h` selects; TF/s = 2*M*N*K*b / time.\n")
print(f"Sum over the step: {tot_us / 1e3:.3f} ms for {tot_flop / 1e9:.0f} GFLOP = {tot_flop / tot_us / 1e6:.0f} TF/s average (in the step itself, with cold operands: see the kernel trace).\n")
print("| form | taps | M | N | K | b | s | n | us / launch | best | TF/s | ms total | epilogue |")
print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---|---:|---:|---|")
for t, e, n, flop in rows:
    print(f"| {e['form']} | {e['taps']} | {e['M']} | {e['N']} | {e['K']} | {e['batch']} | {e['split_k']} | {n} | {e['best_us']:.1f} | {e['best']} | "
          f"{flop / e['best_us'] / 1e6:.0f} | {t / 1e3:.3f} | {e['epilogue'] or '-'} |")
