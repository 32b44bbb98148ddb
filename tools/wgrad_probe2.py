import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clipbert_amd import ops
from tools.gemm_bench import timeit, dev
dt = torch.bfloat16
def wgrad(M, N, K, tile, split=1):
    a, b = torch.randn(K, M, device=dev).to(dt), torch.randn(K, N, device=dev).to(dt)
    out = torch.zeros(M, N, dtype=torch.float32, device=dev)
    us = timeit(lambda: ops.gemm(a, b, M, N, K, out=out, a_mode=ops.KROW, b_mode=ops.KROW, lda=M, ldb=N, accumulate=(split == 1), tile=tile, split_k=split))
    b64 = ((M + 63) // 64) * ((N + 63) // 64)
    print(f"wgrad M={M} N={N} K={K} tile={tile} split={split} blocks={b64*split}: {us:8.1f} us {2.0*M*N*K/us/1e6:7.1f} TF", flush=True)
shapes = ((256, 2304, 6272), (128, 1152, 25088), (1024, 256, 6272), (256, 1024, 6272), (512, 128, 25088), (128, 512, 25088), (64, 576, 100352), (64, 256, 100352), (256, 64, 100352), (64,64,100352),
          (512, 4608, 1568), (2048, 512, 1568), (512, 2048, 1568), (1024, 2048, 1568), (512,1024,6272), (256,512,25088), (1024,512,6272))
for (M, N, K) in shapes:
    b64 = ((M + 63) // 64) * ((N + 63) // 64); kt = (K + 63) // 64
    cands = sorted(set(max(1, min(kt // 4, s)) for s in (1, 256 // b64, 512 // b64, 1024 // b64, 2048//b64)))
    for split in cands:
        wgrad(M, N, K, 2, split)
