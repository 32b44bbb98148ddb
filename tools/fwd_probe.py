import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clipbert_amd import ops
from tools.gemm_bench import timeit, dev
dt = torch.bfloat16
def fwd(M, N, K, tile=0):
    a, b = torch.randn(M, K, device=dev).to(dt), torch.randn(N, K, device=dev).to(dt)
    out = torch.empty(M, N, dtype=dt, device=dev)
    us = timeit(lambda: ops.gemm(a, b, M, N, K, out=out, tile=tile))
    print(f"fwd M={M} N={N} K={K} tile={tile}: {us:8.1f} us {2.0*M*N*K/us/1e6:7.1f} TF", flush=True)
def dgrad(M, N, K, tile=0):
    a, b = torch.randn(M, K, device=dev).to(dt), torch.randn(K, N, device=dev).to(dt)
    out = torch.empty(M, N, dtype=dt, device=dev)
    us = timeit(lambda: ops.gemm(a, b, M, N, K, out=out, b_mode=ops.KROW, ldb=N, tile=tile))
    print(f"dgrad M={M} N={N} K={K} tile={tile}: {us:8.1f} us {2.0*M*N*K/us/1e6:7.1f} TF", flush=True)
for (M, N, K) in ((1312, 768, 3072), (1312, 768, 768), (1312, 2304, 768), (1312, 3072, 768), (1568, 512, 2048), (6272, 256, 1024), (1568, 2048, 512)):
    fwd(M, N, K, 2)
for (M, N, K) in ((1312, 768, 3072), (1312, 768, 768), (1312, 768, 2304), (1312, 3072, 768)):
    dgrad(M, N, K, 2)
