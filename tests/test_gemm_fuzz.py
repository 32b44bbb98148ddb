"""Seeded random shapes / forms / epilogues of cb_gemm against plain torch on the host lane-level emulator: ragged M, N, K,
aligned and unaligned leading dimensions (fast buffer-load path vs guarded generic path), all tile sizes, split-K.
(Emulator only for now; the fixed-shape kernel tests in test_kernels_gemm.py are the ones that also run on the GPU.)"""
import random

import pytest
import torch
import torch.nn.functional as F

from clipbert_amd import ops

CASES = list(range(60))


def _rnd(*shape, gen, scale=1.0):
    return torch.randn(*shape, generator=gen) * scale


@pytest.mark.parametrize("case", CASES)
def test_random_gemm(emul, case):
    class hw:                                     # CPU tensors, host pointers (see the `emul` fixture)
        dev = torch.device("cpu")

        def __new__(cls, t):
            return t

    rng = random.Random(1000 + case)
    gen = torch.Generator().manual_seed(2000 + case)
    dt = rng.choice([torch.bfloat16, torch.bfloat16, torch.float32])
    form = rng.choice(["fwd", "dgrad", "wgrad"])
    aligned = rng.random() < 0.6
    mult = 8 if aligned else 1
    M = rng.randint(1, 40) * (mult if form == "wgrad" else 1) + (0 if aligned else rng.randint(0, 3))
    N = rng.randint(1, 30) * mult + (0 if aligned else rng.randint(0, 5))
    K = rng.randint(1, 25) * mult + (0 if aligned else rng.randint(0, 5))
    # tiles 5-7 = the 8-wave LDS-DMA kernels: taken when the shape qualifies (aligned, N % 8 == 0), silently replaced by the
    # 4-wave kernels otherwise -- either way the answer must be right
    tile = rng.choice([0, 1, 2, 3, 4, 5, 6, 7]) if dt == torch.bfloat16 else 0
    sched = rng.choice([0, 1, 2, 3]) if tile >= 5 else 0
    wsbuf = torch.empty(1 << 20, dtype=torch.float32) if tile >= 5 else None
    xcd = rng.choice([0, 1, 2])
    tol = dict(rtol=3e-2, atol=3e-2) if dt == torch.bfloat16 else dict(rtol=1e-4, atol=1e-4)
    if form == "fwd":
        x, w = hw(_rnd(M, K, gen=gen).to(dt)), hw(_rnd(N, K, gen=gen, scale=0.3).to(dt))
        bias = hw(_rnd(N, gen=gen)) if rng.random() < 0.7 else None
        res = hw(_rnd(M, N, gen=gen).to(dt)) if rng.random() < 0.5 else None
        act = rng.choice([ops.ACT_NONE, ops.ACT_RELU, ops.ACT_GELU, ops.ACT_TANH])
        out = torch.empty(M, N, dtype=dt, device=hw.dev)
        ops.gemm(x, w, M, N, K, out=out, shift=bias, act=act, residual=res, tile=tile, xcd_order=xcd, schedule=sched,
                 split_k=rng.choice([1, 1, 2, 3]) if tile >= 5 else 1, splitk_ws=wsbuf)
        ref = x.float() @ w.float().t()
        if bias is not None:
            ref = ref + bias.float()
        ref = {ops.ACT_NONE: lambda t: t, ops.ACT_RELU: F.relu, ops.ACT_GELU: F.gelu, ops.ACT_TANH: torch.tanh}[act](ref)
        if res is not None:
            ref = ref + res.float()
        torch.testing.assert_close(out.float(), ref, **tol)
    elif form == "dgrad":
        g, w = hw(_rnd(M, K, gen=gen).to(dt)), hw(_rnd(K, N, gen=gen, scale=0.3).to(dt))       # dX[M,N] = g[M,K] W[K,N]
        out = torch.empty(M, N, dtype=dt, device=hw.dev)
        pre = hw(_rnd(M, N, gen=gen).to(dt)) if rng.random() < 0.4 else None
        ops.gemm(g, w, M, N, K, out=out, b_mode=ops.KROW, ldb=N, tile=tile, gelu_grad_pre=pre, xcd_order=xcd, schedule=sched,
                 split_k=rng.choice([1, 2]) if tile >= 5 else 1, splitk_ws=wsbuf)
        ref = g.float() @ w.float()
        if pre is not None:
            p = pre.float().requires_grad_(True)
            F.gelu(p).backward(torch.ones_like(p))
            ref = ref * p.grad
        torch.testing.assert_close(out.float(), ref, **tol)
    else:
        g, x = hw(_rnd(K, M, gen=gen).to(dt)), hw(_rnd(K, N, gen=gen).to(dt))                  # dW[M,N] = g^T x, reduction K
        split = rng.choice([1, 1, 2, 3])
        out = torch.ones(M, N, dtype=torch.float32, device=hw.dev)
        rs = torch.zeros(M, dtype=torch.float32, device=hw.dev) if rng.random() < 0.5 else None
        ops.gemm(g, x, M, N, K, out=out, a_mode=ops.KROW, b_mode=ops.KROW, lda=M, ldb=N, accumulate=True, split_k=split, tile=tile,
                 a_rowsum=rs, xcd_order=xcd, schedule=sched, splitk_ws=wsbuf)
        tolw = dict(rtol=3e-2, atol=6e-2) if dt == torch.bfloat16 else dict(rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(out, 1.0 + g.float().t() @ x.float(), **tolw)
        if rs is not None:
            torch.testing.assert_close(rs, g.float().sum(0), **tolw)
