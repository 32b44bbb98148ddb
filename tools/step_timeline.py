#!/usr/bin/env python
"""Timeline of ONE steady-state training step out of a rocprofv3 --kernel-trace CSV (plain or .gz) of bench.py: every kernel in launch
order with its duration, grid and the gap before it, plus a per-phase summary (ResNet forward / encoder forward / heads + loss /
encoder backward / ResNet backward / optimizer), cut at kernels whose names mark the phase boundaries.
    python tools/step_timeline.py <kernel_trace.csv[.gz]> [--full]"""
import csv
import gzip
import re
import sys

sys.path.insert(0, __import__("os").path.dirname(__file__))
import trace_summary as TS  # noqa: E402


def load(path):
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rt") as fh:
        rows = [r for r in csv.DictReader(fh) if r["Kind"] == "KERNEL_DISPATCH"]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return rows


def main():
    rows = load(sys.argv[1])
    full = "--full" in sys.argv
    # one step = from the kernel after an optimizer group to the end of the next one; like tools/trace_summary.py the step BEFORE the
    # last optimizer group of the trace is taken (when the roofline replays follow the timed region, the last group belongs to their
    # eager, un-captured step)
    ad = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
    groups = []
    for i in ad:
        if not groups or i - groups[-1][-1] > 50:
            groups.append([i])
        else:
            groups[-1].append(i)
    a, b = groups[-3][-1] + 1, groups[-2][-1] + 1
    step = rows[a:b]
    t0 = int(step[0]["Start_Timestamp"])
    phase, phases = "resnet fwd", {}
    order = []
    seen_embed = seen_loss = seen_vbwd = False
    prev_end = t0
    for r in step:
        n = r["Kernel_Name"]
        if "text_embed_fwd" in n or "visual_embed_fwd" in n:
            phase = "encoder fwd"
        elif "lse_loss" in n or "cross_entropy" in n:
            phase = "heads + loss + encoder bwd" if seen_loss or True else phase
            seen_loss = True
        elif "visual_embed_bwd" in n or "text_embed_bwd" in n:
            phase = "embed bwd -> resnet bwd"
        elif "sq_sum" in n or "adamw" in n:
            phase = "optimizer"
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        d = phases.setdefault(phase, [0.0, 0, 0.0])
        if phase not in order:
            order.append(phase)
        d[0] += (e - s) / 1e3; d[1] += 1; d[2] += max(0, s - prev_end) / 1e3
        if full:
            nm = TS.gemm_name(n) or re.sub(r"\(.*", "", n)[:60]
            grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]))
            print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  gap {max(0, s - prev_end) / 1e3:5.1f}  wg {grid:6d}  {phase[:12]:12s} {nm}")
        prev_end = e
    tot = (int(step[-1]["End_Timestamp"]) - t0) / 1e3
    print(f"step: {len(step)} kernels, {tot:.1f} us")
    for p in order:
        d = phases[p]
        print(f"  {p:32s} {d[0]:8.1f} us busy  {d[1]:4d} kernels  gaps {d[2]:6.1f} us")


if __name__ == "__main__":
    main()
