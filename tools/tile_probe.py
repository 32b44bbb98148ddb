import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clipbert_amd import ops
dev = torch.device("cuda", 0)
dt = torch.bfloat16
def graph_time(fn, reps=20, outer=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(outer): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * outer)
def run(form, M, N, K, tile):
    if form == "fwd":
        a, b = torch.randn(M, K, device=dev).to(dt), torch.randn(N, K, device=dev).to(dt)
        out = torch.empty(M, N, dtype=dt, device=dev)
        fn = lambda: ops.gemm(a, b, M, N, K, out=out, tile=tile)
    elif form == "dgrad":
        a, b = torch.randn(M, K, device=dev).to(dt), torch.randn(K, N, device=dev).to(dt)
        out = torch.empty(M, N, dtype=dt, device=dev)
        fn = lambda: ops.gemm(a, b, M, N, K, out=out, b_mode=ops.KROW, ldb=N, tile=tile)
    else:
        a, b = torch.randn(K, M, device=dev).to(dt), torch.randn(K, N, device=dev).to(dt)
        out = torch.zeros(M, N, dtype=torch.float32, device=dev)
        fn = lambda: ops.gemm(a, b, M, N, K, out=out, a_mode=ops.KROW, b_mode=ops.KROW, lda=M, ldb=N, accumulate=True, tile=tile)
    us = graph_time(fn)
    print(f"{form:5s} M={M:6d} N={N:5d} K={K:5d} tile={tile}: {us:7.2f} us {2.0*M*N*K/us/1e6:7.1f} TF", flush=True)
if __name__ == "__main__":
    shapes = [("fwd", 1312, 768, 3072), ("fwd", 1312, 768, 768), ("fwd", 1312, 2304, 768), ("fwd", 1312, 3072, 768),
              ("dgrad", 1312, 768, 3072), ("dgrad", 1312, 768, 768), ("dgrad", 1312, 768, 2304), ("dgrad", 1312, 3072, 768),
              ("wgrad", 3072, 768, 1312), ("wgrad", 768, 3072, 1312), ("wgrad", 2304, 768, 1312), ("wgrad", 768, 768, 1312),
              ("fwd", 6272, 1024, 256), ("fwd", 6272, 256, 1024), ("fwd", 25088, 512, 128), ("fwd", 25088, 128, 512), ("fwd", 1568, 2048, 512), ("fwd", 1568, 512, 2048),
              ("dgrad", 6272, 1024, 256), ("dgrad", 6272, 256, 1024), ("dgrad", 25088, 512, 128), ("dgrad", 25088, 128, 512)]
    sel = [s for s in shapes if s[3] <= 768 and s[0] != "wgrad"] + [("fwd", 100352, 256, 64), ("fwd", 100352, 64, 256), ("fwd", 25088, 512, 256), ("fwd", 6272, 1024, 512), ("fwd", 6272, 256, 2304)]
    for s in (sel if os.environ.get("SHORT") else shapes):
        for tile in ((2,) if os.environ.get("SHORT") else (2, 3, 1)):
            run(*s, tile)
