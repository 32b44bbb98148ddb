import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clipbert_amd import ops
from tools.gemm_bench import timeit, dev
B, L, H = 32, 41, 12
for L in (41, 64, 16):
    qkv = torch.randn(B * L, 3 * H * 64, device=dev).bfloat16()
    mask = torch.ones(B, L, device=dev)
    dctx = torch.randn(B * L, H * 64, device=dev).bfloat16()
    for p in (0.0, 0.1):
        ctx, lse = ops.attention_fwd(qkv, mask, B, L, H, save_lse=True, dropout_p=p, dropout_seed=3)
        f = timeit(lambda: ops.attention_fwd(qkv, mask, B, L, H, save_lse=True, dropout_p=p, dropout_seed=3))
        b = timeit(lambda: ops.attention_bwd(qkv, mask, ctx, dctx, lse, B, L, H, dropout_p=p, dropout_seed=3))
        print(f"L={L} p={p}: fwd {f:6.1f} us  bwd {b:6.1f} us", flush=True)
