#!/usr/bin/env python
"""One steady-state training step out of a rocprofv3 --kernel-trace CSV of bench.py, as a markdown table.
    python tools/trace_summary.py <kernel_trace.csv> [out.md]
The step is cut between two optimizer launches (adamw_kernel groups); cb_gemm instantiations are given readable names."""
import collections
import csv
import re
import sys


def gemm_name(n):
    m = re.search(r"gemm_dma_kernelILi(\d+)ELi(\d+)E", n)
    if m:
        gather = "RowkDmaILi%sELb1" % m.group(1) in n
        return f"cb_gemm {m.group(1)}x{m.group(2)} LDS-DMA ring: {'conv fwd (pixel gather)' if gather else 'linear / 1x1 fwd'}"
    m = re.search(r"gemm_kernelI(DF16b|f)Li(\d+)ELi(\d+)E(.*)", n)
    if not m:
        m2 = re.search(r"gemm_kernel<([^,]+), (\d+), (\d+), (.*)", n)
        if not m2:
            return None
        ty, bm, bn, rest = m2.group(1), m2.group(2), m2.group(3), m2.group(4)
        a_g = "RowkFast<" in rest and ", true>" in rest.split("RowkFast<")[1][:40]
        kinds = re.findall(r"KrowTr<\d+, (\d)>", rest)
    else:
        ty, bm, bn, rest = m.group(1), m.group(2), m.group(3), m.group(4)
        a_g = re.search(r"RowkFastI\w+Li\d+ELb1E", rest) is not None
        kinds = re.findall(r"KrowTrILi\d+ELi(\d)E", rest)
        if "KrowTr" in rest and "S2_" in rest and len(kinds) == 1:
            kinds = kinds * 2
    dt = "bf16" if ty in ("DF16b", "__bf16") else "fp32"
    tile = f"cb_gemm {bm}x{bn} {dt}"
    if "RowkFast" in rest and "KrowFast" in rest:
        return f"{tile}: dgrad (register-transposed B)"
    if "RowkFast" in rest and kinds:
        return f"{tile}: dgrad {'3x3 conv (pixel gather x flipped taps)' if a_g else 'linear / 1x1'} (tr-read B)"
    if len(kinds) == 2:
        return f"{tile}: wgrad {'conv (pixel gather)' if kinds[1] == '2' else 'linear / 1x1 (+ bias row sums)'} (tr-read A, B)"
    if "RowkFast" in rest:
        return f"{tile}: fwd (register-staged)"
    return f"{tile}: other"


def short(n):
    g = gemm_name(n)
    if g:
        return g
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([A-Za-z0-9_]+?)I", n)
    if m:
        return m.group(1)
    return re.split(r"[<(]", n)[0][:60]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
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
    t1 = max(int(r["End_Timestamp"]) for r in step)
    agg = collections.defaultdict(lambda: [0, 0])
    busy, cur = 0, t0
    for r in step:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        k = short(r["Kernel_Name"])
        agg[k][0] += e - s
        agg[k][1] += 1
        busy += max(0, e - max(s, cur))
        cur = max(cur, e)
    lines = [f"One steady-state step (hipGraph replay): {len(step)} kernels, wall {(t1 - t0) / 1e6:.3f} ms, GPU busy {busy / 1e6:.3f} ms, "
             f"sum of kernel durations {sum(v[0] for v in agg.values()) / 1e6:.3f} ms", "",
             "| kernel | launches | total ms | avg us | % of step |", "|---|---:|---:|---:|---:|"]
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        lines.append(f"| {k} | {v[1]} | {v[0] / 1e6:.3f} | {v[0] / v[1] / 1e3:.1f} | {100.0 * v[0] / (t1 - t0):.1f} |")
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "a").write(text + "\n")


if __name__ == "__main__":
    main()
