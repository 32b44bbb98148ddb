import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clipbert_amd import ops
from tools.gemm_bench import timeit, dev
x = torch.randn(1312, 768, device=dev).bfloat16(); dy = torch.randn_like(x)
g = torch.ones(768, device=dev); b = torch.zeros(768, device=dev)
y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-12, save_stats=True)
dg, db = torch.zeros(768, device=dev), torch.zeros(768, device=dev)
print("ln fwd us", timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-12, save_stats=True)))
print("ln bwd us", timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dg, db)))
gg = torch.randn(1312, 3072, device=dev).bfloat16(); out = torch.zeros(3072, device=dev)
print("colsum 1312x3072 us", timeit(lambda: ops.colsum(gg, out)))
