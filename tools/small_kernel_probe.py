"""Short kernels of the training step timed inside hipGraphs: N back-to-back launches of the same kernel (instruction cache and
L2 warm) vs the same launches interleaved with a GEMM (as in the step, where ~360 launches of ~40 different kernels alternate)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clipbert_amd import ops

dev = torch.device("cuda", 0)


def graph_time(fn, n=20, reps=5):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


M, D, B, L, H = 2624, 768, 64, 41, 12
x = torch.randn(M, D, device=dev).bfloat16(); dy = torch.randn_like(x)
gam = torch.ones(D, device=dev); bet = torch.zeros(D, device=dev)
y, mean, rstd = ops.layernorm_fwd(x, gam, bet, 1e-12, save_stats=True)
dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
dx, dx2 = torch.empty_like(x), torch.empty_like(x)
qkv = torch.randn(B * L, 3 * H * 64, device=dev).bfloat16()
mask = torch.ones(B, L, device=dev)
dctx = torch.randn(B * L, H * 64, device=dev).bfloat16()
ctx, lse = ops.attention_fwd(qkv, mask, B, L, H, save_lse=True, dropout_p=0.1, dropout_seed=3)
a = torch.randn(M, 768, device=dev).bfloat16(); w = torch.randn(3072, 768, device=dev).bfloat16(); out = torch.empty(M, 3072, device=dev).bfloat16()

kernels = {
    "layernorm_fwd": lambda: ops.layernorm_fwd(x, gam, bet, 1e-12, save_stats=True, out=y),
    "layernorm_bwd (+dropped copy)": lambda: ops.layernorm_bwd(dy, x, gam, mean, rstd, dg, db, dropout_p=0.1, dropout_seed=3, dx=dx, dx2=dx2),
    "layernorm_bwd (no dropout)": lambda: ops.layernorm_bwd(dy, x, gam, mean, rstd, dg, db, dx=dx),
    "attention_fwd p=0.1": lambda: ops.attention_fwd(qkv, mask, B, L, H, save_lse=True, dropout_p=0.1, dropout_seed=3, out=ctx),
    "attention_bwd p=0.1": lambda: ops.attention_bwd(qkv, mask, ctx, dctx, lse, B, L, H, dropout_p=0.1, dropout_seed=3),
    "attention_fwd p=0": lambda: ops.attention_fwd(qkv, mask, B, L, H, save_lse=True, out=ctx),
    "attention_bwd p=0": lambda: ops.attention_bwd(qkv, mask, ctx, dctx, lse, B, L, H),
}
gemm = lambda: ops.gemm(a, w, M, 3072, 768, out=out)
tg = graph_time(gemm)
print(f"gemm 2624x3072x768 alone: {tg:.1f} us")
for name, fn in kernels.items():
    t_same = graph_time(fn)
    t_mix = graph_time(lambda: (gemm(), fn())) - tg
    print(f"{name:32s} back-to-back {t_same:6.1f} us   interleaved with the GEMM {t_mix:6.1f} us", flush=True)
