#!/usr/bin/env python
"""Every logged GEMM call of the step (gemm_calls.json of tools/gemm_breakdown.py) joined with the kernels of ONE hipGraph replay of the
same step in bench.py's kernel trace, by shape: launches, us per launch IN THE STEP, algorithmic GB/s and TF/s.

    python tools/join_calls_trace.py gpurun_out/final/gemm_calls.json gpurun_out/final/trace/bench_kernel_trace.csv.gz > profiles/rNN_gemm_by_shape_instep.txt

A call may launch several kernels (grouped chunks; the slab reduce after an 8-wave K split is folded into its GEMM): `kernels` of the log."""
import collections
import csv
import gzip
import io
import json
import sys

GEMM = ("gemm_kernel", "gemm8_kernel", "gemm_dma_kernel", "gemm_group_kernel", "gemm8p_kernel", "gemm_stream_kernel", "gemm_streamk_kernel",
        "res2_block_kernel", "stem_pool_kernel")


def main():
    calls = json.load(open(sys.argv[1]))
    f = gzip.open(sys.argv[2]) if sys.argv[2].endswith(".gz") else open(sys.argv[2], "rb")
    rows = list(csv.DictReader(io.TextIOWrapper(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    first = [i for i, r in enumerate(rows) if "stem_pack" in r["Kernel_Name"]]          # the step's first kernel
    step = rows[first[-2]:first[-1]]
    g = []
    for r in step:
        n, d = r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if "splitk_reduce_kernel" in n:
            g[-1] += d
        elif any(k in n for k in GEMM):
            g.append(d)
    assert len(g) == sum(c["kernels"] for c in calls), (len(g), sum(c["kernels"] for c in calls))
    pos, agg = 0, collections.OrderedDict()
    for c in calls:
        d = sum(g[pos:pos + c["kernels"]])
        pos += c["kernels"]
        p = c["problems"][0]
        key = (p["family"], p["form"], p["M"], p["N"], p["K"], p["taps"], len(c["problems"]))
        a = agg.setdefault(key, [0, 0.0, 0.0, 0.0])
        a[0] += 1; a[1] += d; a[2] += sum(q["bytes"] for q in c["problems"]); a[3] += sum(q["flop"] for q in c["problems"])
    tot = sum(a[1] for a in agg.values())
    print(f"# {len(calls)} logged calls = {len(g)} kernels of one replayed step, {tot / 1e3:.3f} ms; M x N x K of the FIRST problem of a call, t = taps, g = problems in the call")
    print(f"{'family':42s} {'form':5s} {'M':>7s} {'N':>6s} {'K':>6s} {'t':>2s} {'g':>2s} {'n':>3s} {'us/launch':>9s} {'ms':>6s} {'GB/s':>6s} {'TF/s':>6s}")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k[0][:42]:42s} {k[1]:5s} {k[2]:7d} {k[3]:6d} {k[4]:6d} {k[5]:2d} {k[6]:2d} {a[0]:3d} {a[1] / a[0]:9.1f} {a[1] / 1e3:6.3f} {a[2] / a[1] / 1e3:6.0f} {a[3] / a[1] / 1e6:6.0f}")


if __name__ == "__main__":
    main()
