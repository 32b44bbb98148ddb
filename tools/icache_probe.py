"""Does alternating between different kernels cost time (instruction cache / L2 working set)?  Three cb_gemm problems of the encoder (different
kernel instantiations, each ~50-150 KB of code) are replayed inside a hipGraph grouped (AAAA.. BBBB.. CCCC..) and interleaved (ABCABC..)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clipbert_amd import ops
from clipbert_amd._lib import KROW

dev = torch.device("cuda", 0)
M = 2624


def mk(m, n, k, dgrad=False):
    a = torch.randn(m, k, device=dev).bfloat16()
    w = (torch.randn(k, n, device=dev) if dgrad else torch.randn(n, k, device=dev)).bfloat16()
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    if dgrad:
        return lambda: ops.gemm(a, w, m, n, k, out=out, b_mode=KROW)
    return lambda: ops.gemm(a, w, m, n, k, out=out)


def graph_time(fns, reps=7):
    for f in fns[:3]:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fns:
            f()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best


A = mk(M, 3072, 768)            # 128x128 occ2, forward
B = mk(M, 768, 3072)            # 64x64, forward
C = mk(M, 768, 2304, True)      # 64x64, dgrad (transposing LDS reads of B)
D = mk(M, 768, 768, True)
n = 16
for name, fns in (("A", [A]), ("B", [B]), ("C", [C]), ("D", [D])):
    print(f"{name} alone: {graph_time(fns * n) / n:7.2f} us / launch")
grouped = graph_time([A] * n + [B] * n + [C] * n + [D] * n)
inter = graph_time([A, B, C, D] * n)
print(f"grouped AAAA..BBBB..CCCC..DDDD: {grouped:8.1f} us    interleaved ABCD x {n}: {inter:8.1f} us    ({(inter - grouped) / (4 * n):.2f} us per launch)")
# the same kernel instantiation with different DATA each launch (operands cold in L2, code warm)
Bs = [mk(M, 768, 3072) for _ in range(8)]
print(f"B, 8 different operand sets round-robin: {graph_time(Bs * 2) / 16:7.2f} us / launch")
