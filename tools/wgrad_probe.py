import sys, os
sys.path.insert(0, "/root/repo")
import torch
from clipbert_amd import ops
from tools.gemm_bench import timeit, dev
dt = torch.bfloat16
def wgrad(M, N, K, tile, acc=True, split=1):
    a, b = torch.randn(K, M, device=dev).to(dt), torch.randn(K, N, device=dev).to(dt)
    out = torch.zeros(M, N, dtype=torch.float32, device=dev)
    us = timeit(lambda: ops.gemm(a, b, M, N, K, out=out, a_mode=ops.KROW, b_mode=ops.KROW, lda=M, ldb=N, accumulate=acc, tile=tile, split_k=split))
    print(f"wgrad M={M} N={N} K={K} tile={tile} acc={acc} split={split}: {us:8.1f} us {2.0*M*N*K/us/1e6:7.1f} TF", flush=True)
for K in (1312, 5248, 20992):
    wgrad(3072, 768, K, 2)
wgrad(3072, 768, 1312, 2, acc=False)
wgrad(3072, 768, 1312, 1, acc=False)
wgrad(768, 768, 1312, 2)
wgrad(768, 768, 1312, 2, split=3)
wgrad(2304, 768, 1312, 2)
wgrad(512, 4608, 1568, 2)
wgrad(512, 4608, 1568, 1)
wgrad(768, 18432, 1568, 1)
