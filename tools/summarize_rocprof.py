#!/usr/bin/env python
"""Condense a rocprofv3 --kernel-trace --stats output directory into a small markdown table
(profiles/<name>.md): per kernel total time, calls, average, % of GPU time.  Usage:
    python tools/summarize_rocprof.py gpurun_out/prof profiles/r01_train_step.md "title"
"""
import csv
import glob
import os
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else src
    files = glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)
    if not files:
        files = glob.glob(os.path.join(src, "**", "*stats*.csv"), recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append(r)
    if not rows:
        print("no stats csv under", src)
        return 1
    def g(r, *names):
        for n in names:
            if n in r:
                return r[n]
        return ""
    agg = {}
    for r in rows:
        name = g(r, "Name", "KernelName", "Kernel_Name")
        calls = int(float(g(r, "Calls", "Count") or 0))
        tot = float(g(r, "TotalDurationNs", "TotalDuration(ns)", "Total_Duration") or 0)
        a = agg.setdefault(name, [0, 0.0])
        a[0] += calls
        a[1] += tot
    total = sum(a[1] for a in agg.values())
    lines = [f"# {title}", "", f"source: `{src}` (rocprofv3 --kernel-trace --stats); total kernel time {total/1e6:.3f} ms", "",
             "| kernel | calls | total ms | avg us | % |", "|---|---:|---:|---:|---:|"]
    for name, (calls, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        short = name if len(name) < 110 else name[:107] + "..."
        lines.append(f"| `{short}` | {calls} | {tot/1e6:.3f} | {tot/max(calls,1)/1e3:.2f} | {100*tot/total:.1f} |")
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    open(dst, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:30]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
