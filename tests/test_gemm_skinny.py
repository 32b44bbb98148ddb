"""cb_gemm tile 9 (few rows: csrc/gemm_skinny.hip -- the four waves of a 32x64 workgroup split the reduction) against plain PyTorch fp32
references of the same op, and against the tiled kernels on the same inputs.  Emulator build on CPU, the real library on `gpu`.
The shapes are the heads' (pooler 64x768x768 with a strided A, classifier MLP 64x1536x768 / 64x2x1536, their data gradients incl.
the 2-deep reduction) plus ragged ones."""
import pytest
import torch

from clipbert_amd import ops

BF = torch.bfloat16


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def close(a, b, k):
    tol = 2e-2 * max(1.0, (k / 256) ** 0.5)
    torch.testing.assert_close(a.float(), b, rtol=2e-2, atol=tol)


@pytest.mark.parametrize("M,N,K,act", [(64, 768, 768, ops.ACT_TANH), (64, 1536, 768, ops.ACT_RELU), (64, 2, 1536, ops.ACT_NONE),
                                       (5, 2, 8, ops.ACT_NONE), (33, 70, 200, ops.ACT_GELU), (17, 136, 36, ops.ACT_NONE)])
def test_forward_bias_act(hw, M, N, K, act):
    x, w, b = hw(rnd(M, K, seed=1).to(BF)), hw(rnd(N, K, seed=2, scale=0.1).to(BF)), hw(rnd(N, seed=3))
    out = torch.empty(M, N, dtype=BF, device=hw.dev)
    ops.gemm(x, w, M, N, K, out=out, shift=b, act=act, tile=9)
    ref = x.float() @ w.float().t() + b
    ref = {ops.ACT_TANH: torch.tanh, ops.ACT_RELU: torch.relu, ops.ACT_GELU: torch.nn.functional.gelu, ops.ACT_NONE: lambda t: t}[act](ref)
    close(out, ref, K)
    # the library takes this structure by itself for M <= 64: same bits as asking for it
    auto = torch.empty(M, N, dtype=BF, device=hw.dev)
    ops.gemm(x, w, M, N, K, out=auto, shift=b, act=act)
    assert ops.gemm_plan(x, w, M, N, K, out=auto, shift=b, act=act)[0] == 9
    assert torch.equal(auto, out)
    # and the tiled kernel gives the same result up to the order of the additions
    tiled = torch.empty(M, N, dtype=BF, device=hw.dev)
    ops.gemm(x, w, M, N, K, out=tiled, shift=b, act=act, tile=2)
    close(out, tiled.float(), K)


def test_pooler_form_strided_rows_f32_out_and_residual(hw):
    """A = the first token of every sequence (lda = L * d), fp32 output with a 2-column pitch, residual + dropout-free epilogue"""
    B, L, d = 24, 5, 64
    seq = hw(rnd(B, L, d, seed=1).to(BF))
    w, b = hw(rnd(48, d, seed=2, scale=0.2).to(BF)), hw(rnd(48, seed=3))
    res = hw(rnd(B, 48, seed=4).to(BF))
    out = torch.empty(B, 48, dtype=BF, device=hw.dev)
    ops.gemm(seq, w, B, 48, d, out=out, lda=L * d, shift=b, act=ops.ACT_TANH, residual=res, tile=9)
    close(out, torch.tanh(seq[:, 0].float() @ w.float().t() + b) + res.float(), d)
    w2 = hw(rnd(2, d, seed=5).to(BF))
    o32 = torch.empty(B, 2, dtype=torch.float32, device=hw.dev)
    ops.gemm(seq, w2, B, 2, d, out=o32, lda=L * d, tile=9)
    close(o32, seq[:, 0].float() @ w2.float().t(), d)


@pytest.mark.parametrize("M,N,K", [(64, 768, 1536), (64, 1536, 2), (64, 768, 768), (40, 72, 100), (7, 8, 3)])
def test_data_gradient_reduction_major_weights(hw, M, N, K):
    """dX[M, N] = g[M, K] W[K, N]  (B reduction-major: staged lines + transpose read)"""
    g, w = hw(rnd(M, K, seed=1).to(BF)), hw(rnd(K, N, seed=2, scale=0.1).to(BF))
    dx = torch.empty(M, N, dtype=BF, device=hw.dev)
    ops.gemm(g, w, M, N, K, out=dx, b_mode=ops.KROW, tile=9)
    close(dx, g.float() @ w.float(), K)
    assert ops.gemm_plan(g, w, M, N, K, out=dx, b_mode=ops.KROW)[0] == 9


def test_accumulate_into_strided_rows(hw):
    """the pooler's data gradient: added onto row 0 of every sequence of a (B, L, d) gradient (ldc = L * d, accumulate)"""
    B, L, d = 20, 3, 64
    g, w = hw(rnd(B, d, seed=1).to(BF)), hw(rnd(d, d, seed=2, scale=0.2).to(BF))
    dx = hw(rnd(B, L, d, seed=3).to(BF))
    before = dx.float().clone()
    ops.gemm(g, w, B, d, d, out=dx, b_mode=ops.KROW, ldc=L * d, accumulate=True, tile=9)
    ref = before.clone()
    ref[:, 0] += g.float() @ w.float()
    close(dx, ref, d)
    assert torch.equal(dx[:, 1:].float(), before[:, 1:])


def test_what_it_does_not_cover_fails_loudly(hw):
    g, x = hw(rnd(64, 16, seed=1).to(BF)), hw(rnd(64, 24, seed=2).to(BF))
    dw = torch.zeros(16, 24, dtype=torch.float32, device=hw.dev)
    with pytest.raises(RuntimeError, match="tile 9"):
        ops.gemm(g, x, 16, 24, 64, out=dw, a_mode=ops.KROW, b_mode=ops.KROW, accumulate=True, tile=9)      # weight-gradient form
    w = hw(rnd(24, 16, seed=3).to(torch.float32))
    with pytest.raises(RuntimeError, match="tile 9"):
        ops.gemm(hw(rnd(8, 16, seed=4)), w, 8, 24, 16, out=torch.empty(8, 24, device=hw.dev), tile=9)      # fp32 parity mode
