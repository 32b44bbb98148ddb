import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clipbert_amd import ops
dev = torch.device("cuda", 0)
dt = torch.bfloat16
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
tile = int(sys.argv[4]) if len(sys.argv) > 4 else 2
a, b = torch.randn(M, K, device=dev).to(dt), torch.randn(N, K, device=dev).to(dt)
out = torch.empty(M, N, dtype=dt, device=dev)
for _ in range(5):
    ops.gemm(a, b, M, N, K, out=out, tile=tile)
torch.cuda.synchronize()
