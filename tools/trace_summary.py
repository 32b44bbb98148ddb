#!/usr/bin/env python
"""One steady-state training step out of a rocprofv3 --kernel-trace CSV of bench.py, as a markdown table.
    python tools/trace_summary.py <kernel_trace.csv> [out.md]
The step is cut between two optimizer launches (adamw_kernel groups); cb_gemm instantiations are given readable names."""
import collections
import csv
import re
import sys


def gemm_name(n):
    """readable name of a cb_gemm instantiation from its (possibly half-demangled) symbol"""
    if "gemm_group_kernel" in n:                        # cb_gemm_group: same instantiation scheme as gemm_kernel
        g = gemm_name(n.replace("gemm_group_kernel", "gemm_kernel"))
        return g.replace("cb_gemm ", "cb_gemm_group ", 1) if g else "cb_gemm_group"
    if "gemm_stream_kernel" in n:                       # streaming structure: <BM, BN, KT, B mode (-1 ROWK = forward, else data gradient), OCC, EPI>
        m = re.search(r"gemm_stream_kernel<(\d+), (\d+), (\d+), \d+, (\d+)>", n) or re.search(r"gemm_stream_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi\d+ELi(\d+)E", n)
        if not m:
            return "cb_gemm streaming"
        bm, bn, kt, epi = m.groups()
        return f"cb_gemm streaming {bm}x{bn} K<={int(kt) * 64} bf16 (persistent, weights resident): fwd 1x1 conv" + ("" if epi == "0" else " + residual")
    if "gemm_skinny_kernel" in n:                       # few rows (tile 9): <B reduction-major>
        return "cb_gemm few rows 32x64 bf16 (waves split K): " + ("dgrad linear" if re.search(r"gemm_skinny_kernel(<true>|ILb1E)", n) else "fwd linear")
    if "splitk_reduce_kernel" in n:
        return "cb_gemm split-K reduce (slabs -> epilogue)"
    d8 = re.search(r"gemm8_kernel<(\d+), (\d+), \d+, \d+, \d+, (\d), cbgemm::(\w+)<\d+, (\w+)>, cbgemm::(\w+)<\d+, (\w+)>, (\w+)>", n)
    if d8:                                                 # demangled form (rocprofv3 7.x)
        bm, bn, sched, ka, aa, kb, ab, rs = d8.groups()
        if ka == "RowkDma8" and kb == "RowkDma8":
            form = "fwd conv (pixel gather)" if aa == "true" else "fwd linear / 1x1 conv"
        elif ka == "RowkDma8":
            form = "dgrad 3x3 conv (pixel gather x flipped taps)" if aa == "true" else "dgrad linear / 1x1 conv"
        else:
            form = "wgrad conv (pixel gather)" if ab == "2" else ("wgrad linear / 1x1 conv + bias row sums" if rs == "true" else "wgrad linear / 1x1 conv")
        return f"cb_gemm 8-wave {bm}x{bn} bf16 LDS-DMA (schedule {sched}): {form}"
    m8 = re.search(r"gemm8_kernelILi(\d+)ELi(\d+)ELi\d+ELi\d+ELi\d+ELi(\d)E(.*)", n)
    if m8:
        rest = m8.group(4)
        loaders = re.findall(r"(RowkDma8|KrowDma8)ILi\d+EL([bi])(\d)", rest)
        if len(loaders) == 1:                              # second loader abbreviated as a substitution (same template, other arguments)
            loaders = loaders * 2
        (ka, _ta, aa), (kb, _tb, ab) = (loaders + [("?", "", "0")] * 2)[:2]
        if ka == "RowkDma8" and kb == "RowkDma8" and "RowkDma8" in rest and "KrowDma8" not in rest:
            form = "fwd conv (pixel gather)" if aa == "1" else "fwd linear / 1x1 conv"
        elif ka == "RowkDma8":
            form = "dgrad 3x3 conv (pixel gather x flipped taps)" if aa == "1" else "dgrad linear / 1x1 conv"
        else:
            form = "wgrad conv (pixel gather)" if re.search(r"(KrowDma8|NS1_)ILi\d+ELi2E", rest) else "wgrad linear / 1x1 conv (+ bias row sums)"
        return f"cb_gemm 8-wave {m8.group(1)}x{m8.group(2)} bf16 LDS-DMA (schedule {m8.group(3)}): {form}"
    if "gemm_kernel" not in n and "gemm_dma_kernel" not in n:
        return None
    m = re.search(r"gemm_dma_kernelILi(\d+)ELi(\d+)E", n)
    if m:
        gather = "RowkDmaILi%sELb1" % m.group(1) in n
        return f"cb_gemm {m.group(1)}x{m.group(2)} bf16 LDS-DMA ring: {'conv fwd (pixel gather)' if gather else 'linear / 1x1 fwd'}"
    m = re.search(r"gemm_kernelI(DF16b|f)Li(\d+)ELi(\d+)E(.*)", n)
    if m:
        dt, bm, bn, rest = ("bf16" if m.group(1) == "DF16b" else "fp32"), m.group(2), m.group(3), m.group(4)
    else:                                               # rocprofv3 sometimes prints a half-demangled form
        m = re.search(r"ELi(\d+)E(NS_.*)", n)
        if not m:
            return "cb_gemm (unparsed)"
        dt, bm, bn, rest = "bf16", m.group(1), m.group(1), m.group(2)
    # first loader
    a = re.search(r"NS_\d+(RowkFast|KrowTr|KrowFast|RowkLoader|KrowLoader)I(?:DF16b|f)?(?:Li\d+E)?L([bi])(\d)", rest)
    if not a:
        return f"cb_gemm {bm}x{bn} {dt}: other"
    a_kind, a_arg = a.group(1), a.group(3)
    after = rest[a.end():]
    if re.match(r"E*S2_", after) or re.match(r"[^N]*S2_", after[:12]):
        b_kind, b_arg = a_kind, a_arg
    else:
        b = re.search(r"NS_\d+(RowkFast|KrowTr|KrowFast|RowkLoader|KrowLoader)I(?:DF16b|f)?(?:Li\d+E)?L([bi])(\d)", after)
        if b:
            b_kind, b_arg = b.group(1), b.group(3)
        else:
            b2 = re.search(r"NS1_I(?:DF16b|f)?(?:Li\d+E)?L([bi])(\d)", after)
            b_kind, b_arg = a_kind, (b2.group(2) if b2 else "?")
    tile = f"cb_gemm {bm}x{bn} {dt}"
    if a_kind in ("RowkFast", "RowkLoader") and b_kind in ("RowkFast", "RowkLoader"):
        return f"{tile}: fwd {'conv (pixel gather)' if a_arg == '1' else 'linear / 1x1 conv'}"
    if a_kind in ("RowkFast", "RowkLoader"):
        how = {"KrowTr": "tr-read B", "KrowFast": "register-transposed B", "KrowLoader": "guarded loads"}[b_kind]
        return f"{tile}: dgrad {'3x3 conv (pixel gather x flipped taps)' if a_arg == '1' else 'linear / 1x1 conv'} ({how})"
    how = {"KrowTr": "tr-read A, B", "KrowFast": "register-transposed", "KrowLoader": "guarded loads"}[a_kind]
    return f"{tile}: wgrad {'conv (pixel gather)' if b_arg == '2' else 'linear / 1x1 conv (+ bias row sums)'} ({how})"


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
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()
