import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clipbert_amd import ops
from tools.tile_probe import graph_time, dev, dt
def run(nb, M, N, K, tile, acc, rowsum):
    a, b = torch.randn(nb, K, M, device=dev).to(dt), torch.randn(nb, K, N, device=dev).to(dt)
    out = torch.zeros(nb, M, N, dtype=torch.float32, device=dev)
    rs = torch.zeros(nb, M, dtype=torch.float32, device=dev) if rowsum else None
    fn = lambda: ops.gemm(a, b, M, N, K, out=out, a_mode=ops.KROW, b_mode=ops.KROW, lda=M, ldb=N, ldc=N, accumulate=acc, tile=tile,
                          a_rowsum=rs, batch=nb, batch_strides=(K * M, K * N, M * N, M))
    us = graph_time(fn, reps=5, outer=5)
    print(f"bwgrad nb={nb} M={M} N={N} K={K} tile={tile} acc={acc} rowsum={rowsum}: {us:8.1f} us {2.0*nb*M*N*K/us/1e6:7.1f} TF", flush=True)
for (M, N) in ((3072, 768), (768, 3072), (2304, 768), (768, 768)):
    for tile in (1, 2):
        for acc in (True, False):
            run(12, M, N, 1312, tile, acc, True)
    run(12, M, N, 1312, 2, True, False)
    run(1, M, N, 1312, 2, True, True)
