#!/usr/bin/env python
"""Micro-benchmark of cb_gemm on the GPU: per-shape time and TFLOP/s (HIP events over back-to-back launches)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clipbert_amd import ops

dev = torch.device("cuda", 0)


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def run(M, N, K, form="fwd", tile=0, epi="none", dt=torch.bfloat16):
    if form == "fwd":
        a, b = torch.randn(M, K, device=dev).to(dt), torch.randn(N, K, device=dev).to(dt)
        out = torch.empty(M, N, dtype=dt, device=dev)
        kw = {}
        if epi == "bias":
            kw = dict(shift=torch.randn(N, device=dev))
        elif epi == "gelu":
            kw = dict(shift=torch.randn(N, device=dev), act=ops.ACT_GELU, out2=torch.empty(M, N, dtype=dt, device=dev))
        elif epi == "res":
            kw = dict(shift=torch.randn(N, device=dev), residual=torch.randn(M, N, device=dev).to(dt))
        fn = lambda: ops.gemm(a, b, M, N, K, out=out, tile=tile, **kw)
    elif form == "dgrad":   # dX[M,N] = g[M,K] W[K,N]
        a, b = torch.randn(M, K, device=dev).to(dt), torch.randn(K, N, device=dev).to(dt)
        out = torch.empty(M, N, dtype=dt, device=dev)
        fn = lambda: ops.gemm(a, b, M, N, K, out=out, b_mode=ops.KROW, tile=tile)
    else:                   # wgrad dW[M,N] = g[K,M]^T x[K,N]
        a, b = torch.randn(K, M, device=dev).to(dt), torch.randn(K, N, device=dev).to(dt)
        out = torch.zeros(M, N, dtype=torch.float32, device=dev)
        fn = lambda: ops.gemm(a, b, M, N, K, out=out, a_mode=ops.KROW, b_mode=ops.KROW, lda=M, ldb=N, accumulate=True, tile=tile)
    us = timeit(fn)
    print(f"{form:6s} M={M:6d} N={N:6d} K={K:6d} tile={tile} epi={epi:5s} {us:9.1f} us  {2.0*M*N*K/us/1e6:8.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    for tile in (1, 2):
        for K in (768, 3072, 12288):
            run(1312, 3072, K, "fwd", tile)
    run(1312, 3072, 768, "fwd", 1, "bias")
    run(1312, 3072, 768, "fwd", 1, "gelu")
    run(1312, 768, 3072, "fwd", 2, "res")
    run(8192, 8192, 1024, "fwd", 1)
    run(8192, 8192, 4096, "fwd", 1)
    run(4096, 4096, 4096, "fwd", 2)
    for tile in (1, 2):
        run(1312, 768, 3072, "dgrad", tile)
        run(8192, 8192, 1024, "dgrad", tile)
        run(3072, 768, 1312, "wgrad", tile)
        run(4096, 4096, 4096, "wgrad", tile)
