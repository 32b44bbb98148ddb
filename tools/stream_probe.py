#!/usr/bin/env python
"""The streaming structure (cb_gemm tile 8) against the library's one-workgroup-per-tile choice, problem by problem, on every GEMM of
the bench step that tile 8 covers: cold-cache time of one launch (tools/tune_gemm.py: 384 MB flush + fresh write of A before each) and
the algorithmic HBM rate (clipbert_amd/gemm_log.py byte count).   python tools/stream_probe.py [--out gpurun_out/stream_probe.json]"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["CB_GEMM_STREAM_MIN_ROWS"] = "2000000000"      # auto never streams in this process: tile 0 = the tuned one-workgroup-per-tile launch
import torch  # noqa: E402

import tune_gemm  # noqa: E402
from clipbert_amd import _lib, gemm_log, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "stream_probe.json"))
    ap.add_argument("--mode", default="train")
    args = ap.parse_args()
    tune_gemm.COLD = True
    calls = tune_gemm.record_calls(args.mode)
    seen, rows = set(), []
    for pos, kw in calls:
        a, b, M, N, K = pos
        d = ops.gemm_desc(a, b, M, N, K, **dict(kw, tile=8))
        plan = (C.c_int32 * 4)()
        if _lib.get().cb_gemm_plan(C.byref(d), 1, plan) != 0 or plan[0] != 8:
            continue
        key = tune_gemm.key_of(pos, kw)
        if key in seen:
            continue
        seen.add(key)
        count = sum(1 for p2, k2 in calls if tune_gemm.key_of(p2, k2) == key)
        prob = gemm_log._problem(ops.gemm_desc(a, b, M, N, K, **kw))
        base = {kk: v for kk, v in kw.items() if kk not in ("tile", "xcd_order")}
        auto, _ = tune_gemm.time_config(pos, base, 0, 0)
        stream, err = tune_gemm.time_config(pos, base, 8, 0)
        row = dict(form=prob["form"], M=M, N=N, K=K, epilogue=key[9], launches_per_step=count, variant=int(plan[2]), auto_us=round(auto, 2),
                   stream_us=round(stream, 2) if stream else None, algorithmic_mbytes=round(prob["bytes"] / 1e6, 1),
                   auto_tbs=round(prob["bytes"] / auto / 1e6, 2), stream_tbs=round(prob["bytes"] / stream / 1e6, 2) if stream else None, err=err)
        rows.append(row)
        print(row, flush=True)
    tot_a = sum(r["auto_us"] * r["launches_per_step"] for r in rows)
    tot_s = sum((r["stream_us"] or r["auto_us"]) * r["launches_per_step"] for r in rows)
    summary = dict(problems=len(rows), launches_per_step=sum(r["launches_per_step"] for r in rows), auto_us_per_step=round(tot_a, 1), stream_us_per_step=round(tot_s, 1))
    print(summary)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(dict(summary=summary, rows=rows), open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
