import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clipbert_amd import ops
from tools.gemm_bench import timeit, dev
dt = torch.bfloat16
def fwd(M, N, K, tile=2):
    a, b = torch.randn(M, K, device=dev).to(dt), torch.randn(N, K, device=dev).to(dt)
    out = torch.empty(M, N, dtype=dt, device=dev)
    g = torch.cuda.CUDAGraph()
    ops.gemm(a, b, M, N, K, out=out, tile=tile)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(20):
            ops.gemm(a, b, M, N, K, out=out, tile=tile)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 200
    print(f"fwd M={M} N={N} K={K} tile={tile}: {us:8.2f} us {2.0*M*N*K/us/1e6:7.1f} TF", flush=True)
for M in (1312, 5248):
    for K in (64, 256, 768, 1536, 3072, 6144):
        fwd(M, 768, K)
