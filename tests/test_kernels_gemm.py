"""GEMM / implicit-GEMM convolution kernels (clipbert_amd/csrc/gemm.hip) against plain PyTorch fp32
references of the same op (forward, dgrad, wgrad).  Every case runs twice: on the host lane-level emulator
build (CPU suite) and, marked `gpu`, through the real libclipbert_hip.so on an MI355X."""
import pytest
import torch
import torch.nn.functional as F

from clipbert_amd import ops

DT = [torch.float32, torch.bfloat16]


def tol(dt):
    return dict(rtol=2e-2, atol=2e-2) if dt == torch.bfloat16 else dict(rtol=1e-4, atol=1e-4)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,N,K,tile", [(100, 72, 96, 2), (130, 136, 64, 1), (5, 2, 8, 0), (33, 37, 40, 2), (300, 136, 128, 4), (270, 40, 192, 3)])
def test_linear_forward_epilogues(hw, dt, M, N, K, tile):
    if dt == torch.float32 and tile in (1, 3, 4):
        tile = 0
    x, w, b = hw(rnd(M, K, seed=1).to(dt)), hw(rnd(N, K, seed=2, scale=0.2).to(dt)), hw(rnd(N, seed=3))
    res = hw(rnd(M, N, seed=4).to(dt))
    out = torch.empty(M, N, dtype=dt, device=hw.dev)
    pre = torch.empty(M, N, dtype=dt, device=hw.dev)
    ops.gemm(x, w, M, N, K, out=out, shift=b, act=ops.ACT_GELU, residual=res, out2=pre, tile=tile)
    ref_pre = x.float() @ w.float().t() + b
    ref = F.gelu(ref_pre) + res.float()
    torch.testing.assert_close(pre.float(), ref_pre, **tol(dt))
    torch.testing.assert_close(out.float(), ref, **tol(dt))


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("tile", [2, 1])
def test_linear_dgrad_wgrad(hw, dt, tile):
    if dt == torch.float32 and tile == 1:
        pytest.skip("fp32 parity mode has one tile size")
    M, N, K = 70, 48, 64
    x, w, g = hw(rnd(M, K, seed=1).to(dt)), hw(rnd(N, K, seed=2, scale=0.2).to(dt)), hw(rnd(M, N, seed=3).to(dt))
    dx = torch.empty(M, K, dtype=dt, device=hw.dev)
    ops.gemm(g, w, M, K, N, out=dx, b_mode=ops.KROW, tile=tile)           # dX = g W
    torch.testing.assert_close(dx.float(), g.float() @ w.float(), **tol(dt))
    dw = torch.zeros(N, K, dtype=torch.float32, device=hw.dev)
    ops.gemm(g, x, N, K, M, out=dw, a_mode=ops.KROW, b_mode=ops.KROW, accumulate=True, split_k=3, tile=tile)   # dW = g^T x
    torch.testing.assert_close(dw, g.float().t() @ x.float(), **tol(dt))
    # tiny / unaligned head shapes (num_labels = 2): scalar guarded loaders
    g2, w2 = hw(rnd(M, 2, seed=5).to(dt)), hw(rnd(2, K, seed=6).to(dt))
    dx2 = torch.empty(M, K, dtype=dt, device=hw.dev)
    ops.gemm(g2, w2, M, K, 2, out=dx2, b_mode=ops.KROW)
    torch.testing.assert_close(dx2.float(), g2.float() @ w2.float(), **tol(dt))
    dw2 = torch.empty(2, K, dtype=torch.float32, device=hw.dev)
    ops.gemm(g2, x, 2, K, M, out=dw2, a_mode=ops.KROW, b_mode=ops.KROW)
    torch.testing.assert_close(dw2, g2.float().t() @ x.float(), **tol(dt))


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("tile", [2, 1])
def test_fused_bias_grad_and_gelu_grad(hw, dt, tile):
    """wgrad with the bias gradient on the matrix core (a_rowsum) and dgrad with GELU' in the epilogue."""
    if dt == torch.float32 and tile == 1:
        pytest.skip("fp32 parity mode has one tile size")
    M, N, K = 150, 72, 136                               # tokens, out features, in features
    x, g = hw(rnd(M, K, seed=1).to(dt)), hw(rnd(M, N, seed=3).to(dt))
    for split in (1, 2):
        dw = torch.zeros(N, K, dtype=torch.float32, device=hw.dev)
        db = torch.ones(N, dtype=torch.float32, device=hw.dev)
        ops.gemm(g, x, N, K, M, out=dw, a_mode=ops.KROW, b_mode=ops.KROW, accumulate=True, split_k=split, tile=tile, a_rowsum=db)
        torch.testing.assert_close(dw, g.float().t() @ x.float(), **tol(dt))
        torch.testing.assert_close(db, 1.0 + g.float().sum(0), **tol(dt))
    # unaligned shapes take the generic loaders: same contract
    g2 = hw(rnd(M, 3, seed=5).to(dt))
    dw2 = torch.zeros(3, K, dtype=torch.float32, device=hw.dev)
    db2 = torch.zeros(3, dtype=torch.float32, device=hw.dev)
    ops.gemm(g2, x, 3, K, M, out=dw2, a_mode=ops.KROW, b_mode=ops.KROW, accumulate=True, a_rowsum=db2)
    torch.testing.assert_close(dw2, g2.float().t() @ x.float(), **tol(dt))
    torch.testing.assert_close(db2, g2.float().sum(0), **tol(dt))
    # dX = (g W) * gelu'(pre)
    w = hw(rnd(N, K, seed=2, scale=0.2).to(dt))
    pre = hw(rnd(M, K, seed=7).to(dt))
    dx = torch.empty(M, K, dtype=dt, device=hw.dev)
    ops.gemm(g, w, M, K, N, out=dx, b_mode=ops.KROW, tile=tile, gelu_grad_pre=pre)
    pr = pre.float().requires_grad_(True)
    F.gelu(pr).backward(g.float() @ w.float())
    torch.testing.assert_close(dx.float(), pr.grad, **tol(dt))
    # round 5: the forward stores gelu'(pre) as its second output (CB_ACT_GELU_SAVE_GRAD), the data gradient multiplies by it as stored
    # (CB_ACT_SAVED_GRAD) -- the pair gives what gelu_grad_pre on the pre-activation gives, up to the storage rounding of the derivative
    wf, bf = hw(rnd(K, N, seed=9, scale=0.2).to(dt)), hw(rnd(K, seed=10))
    xin = hw(rnd(M, N, seed=11).to(dt))
    y, dsave = torch.empty(M, K, dtype=dt, device=hw.dev), torch.empty(M, K, dtype=dt, device=hw.dev)
    y0, pre0 = torch.empty(M, K, dtype=dt, device=hw.dev), torch.empty(M, K, dtype=dt, device=hw.dev)
    ops.gemm(xin, wf, M, K, N, out=y, shift=bf, act=ops.ACT_GELU_SAVE_GRAD, out2=dsave, tile=tile)
    ops.gemm(xin, wf, M, K, N, out=y0, shift=bf, act=ops.ACT_GELU, out2=pre0, tile=tile)
    assert torch.equal(y, y0) or (y.float() - y0.float()).abs().max() <= 1e-6 * max(1.0, float(y0.float().abs().max()))
    z = (xin.float() @ wf.float().t() + bf).requires_grad_(True)
    F.gelu(z).sum().backward()
    torch.testing.assert_close(dsave.float(), z.grad, **tol(dt))
    dx2 = torch.empty(M, K, dtype=dt, device=hw.dev)
    ops.gemm(g, w, M, K, N, out=dx2, b_mode=ops.KROW, tile=tile, gelu_grad_pre=dsave, act=ops.ACT_SAVED_GRAD)
    torch.testing.assert_close(dx2.float(), (g.float() @ w.float()) * dsave.float(), **tol(dt))
    y1 = torch.empty(M, K, dtype=dt, device=hw.dev)
    ops.gemm(xin, wf, M, K, N, out=y1, shift=bf, act=ops.ACT_GELU_SAVE_GRAD, tile=tile)        # no second output: a plain GELU
    assert torch.equal(y1, y0)


@pytest.mark.parametrize("dt", DT)
def test_strided_batched_wgrad(hw, dt):
    """weight gradients (+ bias row sums) of several layers in one launch: grid z = layer * split_k + split."""
    nb, M, N, K = 3, 90, 72, 40
    x, g = hw(rnd(nb, M, K, seed=1).to(dt)), hw(rnd(nb, M, N, seed=3).to(dt))
    for split in (1, 2):
        dw = torch.zeros(nb, N * K + 16, dtype=torch.float32, device=hw.dev)     # layer-strided gradient buffer with a gap
        db = torch.zeros(nb, N + 8, dtype=torch.float32, device=hw.dev)
        ops.gemm(g, x, N, K, M, out=dw, ldc=K, lda=N, ldb=K, a_mode=ops.KROW, b_mode=ops.KROW, accumulate=True, split_k=split,
                 a_rowsum=db, batch=nb, batch_strides=(M * N, M * K, dw.stride(0), db.stride(0)))
        for b in range(nb):
            torch.testing.assert_close(dw[b, :N * K].view(N, K), g[b].float().t() @ x[b].float(), **tol(dt))
            torch.testing.assert_close(db[b, :N], g[b].float().sum(0), **tol(dt))
        assert dw[:, N * K:].abs().max() == 0 and db[:, N:].abs().max() == 0
    # batched forward form
    w = hw(rnd(nb, N, K, seed=5, scale=0.2).to(dt))
    y = torch.empty(nb, M, N, dtype=dt, device=hw.dev)
    ops.gemm(x, w, M, N, K, out=y, lda=K, ldb=K, ldc=N, batch=nb, batch_strides=(M * K, N * K, M * N, 0))
    torch.testing.assert_close(y.float(), torch.einsum("bmk,bnk->bmn", x.float(), w.float()), **tol(dt))


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("tile", [2, 1])
def test_relu_bwd_epilogue(hw, dt, tile):
    """dgrad launch that also does the consumer block's ReLU x FrozenBN-scale backward: C = t*s_a, C2 = t*s_b,
    t = (acc [+ C] [+ residual]) where y > 0."""
    if dt == torch.float32 and tile == 1:
        pytest.skip("fp32 parity mode has one tile size")
    M, N, K = 150, 72, 40
    g, w = hw(rnd(M, K, seed=1).to(dt)), hw(rnd(K, N, seed=2, scale=0.2).to(dt))          # dX = g W  (W as [K][N], KROW)
    y, res = hw(rnd(M, N, seed=3).to(dt)), hw(rnd(M, N, seed=4).to(dt))
    sa, sb = hw(rnd(N, seed=5).abs() + 0.5), hw(rnd(N, seed=6).abs() + 0.5)
    base = g.float() @ w.float()
    for accumulate, residual, s2 in ((False, res, sb), (True, None, None)):
        c0 = hw(rnd(M, N, seed=7).to(dt))
        c = c0.clone()
        c2 = torch.empty(M, N, dtype=dt, device=hw.dev)
        ops.gemm(g, w, M, N, K, out=c, b_mode=ops.KROW, ldb=N, tile=tile, accumulate=accumulate, residual=residual, mask=y,
                 relu_bwd=True, post_scale=sa, post_scale2=s2, out2=c2)
        t = base + (c0.float() if accumulate else 0) + (residual.float() if residual is not None else 0)
        t = torch.where(y.float() > 0, t, torch.zeros_like(t))
        torch.testing.assert_close(c.float(), t * sa, **tol(dt))
        torch.testing.assert_close(c2.float(), t * s2 if s2 is not None else t, **tol(dt))


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("tile,k,stride,pad,H,W,Cin,Cout", [
    (2, 3, 1, 1, 7, 9, 32, 64), (2, 1, 2, 0, 8, 6, 64, 40), (2, 1, 1, 0, 5, 5, 32, 64),
    (1, 3, 1, 1, 7, 9, 32, 64), (1, 1, 2, 0, 8, 6, 64, 40), (1, 1, 1, 0, 5, 5, 32, 64),
    (4, 3, 1, 1, 7, 9, 64, 136), (4, 1, 2, 0, 8, 6, 64, 40), (3, 3, 1, 1, 7, 9, 32, 64)])
def test_conv_forward_backward(hw, dt, tile, k, stride, pad, H, W, Cin, Cout):
    if dt == torch.float32 and tile != 2:
        pytest.skip("fp32 parity mode has one tile size")
    n = 2
    x = hw(rnd(n, Cin, H, W, seed=1).to(dt))
    w = hw(rnd(Cout, Cin, k, k, seed=2, scale=0.1).to(dt))
    scale, shift = hw(rnd(Cout, seed=3).abs() + 0.5), hw(rnd(Cout, seed=4))
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    xh = _nhwc(x)                                        # (n, H, W, Cin)
    wk = w.permute(0, 2, 3, 1).contiguous()              # KRSC
    M = n * OH * OW
    tab = ops.build_pixel_table(n, OH, OW, stride, pad, H * W * Cin, W * Cin, Cin, x.device)
    res = hw(rnd(M, Cout, seed=5).to(dt))
    y = torch.empty(M, Cout, dtype=dt, device=hw.dev)
    ops.gemm(xh, wk, M, Cout, k * k * Cin, out=y, a_mode=ops.ROWK_GATHER, a_tab=tab, lda=0, ldb=k * k * Cin,
             R=k, S=k, Cin=Cin, H=H, W=W, sH=W * Cin, sW=Cin, scale=scale, shift=shift, residual=res, relu_after=True, tile=tile)
    xr = x.float().requires_grad_(True)
    wr = w.float().requires_grad_(True)
    conv = F.conv2d(xr, wr, None, stride, pad)
    ref = F.relu(conv * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + res.float().view(n, OH, OW, Cout).permute(0, 3, 1, 2))
    torch.testing.assert_close(y.float().view(n, OH, OW, Cout), ref.permute(0, 2, 3, 1), **tol(dt))

    # backward of the bare convolution for an upstream gradient g (already scaled/masked)
    g = hw(rnd(n, Cout, OH, OW, seed=6).to(dt))
    conv.backward(g.float())
    gh = _nhwc(g).view(M, Cout)
    # dgrad: transposed conv = gather over g with flipped taps (stride 1), or scatter (1x1 strided)
    dx = torch.zeros(n * H * W, Cin, dtype=dt, device=hw.dev)
    if stride == 1:
        tab_in = ops.build_pixel_table(n, H, W, 1, k - 1 - pad, OH * OW * Cout, OW * Cout, Cout, x.device)
        ops.gemm(gh, wk, n * H * W, Cin, k * k * Cout, out=dx, a_mode=ops.ROWK_GATHER, a_tab=tab_in, lda=0,
                 b_mode=ops.KROW_TAPS, ldb=k * k * Cin, R=k, S=k, Cin=Cout, H=OH, W=OW, sH=OW * Cout, sW=Cout,
                 flip_taps=True, tile=tile)
    else:
        rowmap = (torch.arange(n).view(n, 1, 1) * H * W + (torch.arange(OH) * stride).view(1, OH, 1) * W
                  + (torch.arange(OW) * stride).view(1, 1, OW)).reshape(-1).int().to(hw.dev)
        ops.gemm(gh, wk, M, Cin, Cout, out=dx, b_mode=ops.KROW_TAPS, ldb=Cin, R=1, S=1, Cin=Cout, c_rowmap=rowmap, tile=tile)
        if stride == 2 and H % 2 == 0 and W % 2 == 0:
            # zero_fill_pitch: the launch also writes the zeros of the three other pixels of every 2x2 patch (no pre-zeroed output)
            dz = torch.full((n * H * W, Cin), float("nan"), dtype=dt, device=hw.dev)
            d2 = torch.full((n * H * W, Cin), float("nan"), dtype=dt, device=hw.dev)
            ops.gemm(gh, wk, M, Cin, Cout, out=dz, b_mode=ops.KROW_TAPS, ldb=Cin, R=1, S=1, Cin=Cout, c_rowmap=rowmap, tile=tile,
                     zero_fill_pitch=W)
            assert torch.equal(dz, dx)
            # ... also for the two outputs of the fused ReLU x FrozenBN backward, accumulating onto a first strided gradient
            ymask = hw(rnd(n * H * W, Cin, seed=9).to(dt))
            ps = hw(rnd(Cin, seed=10).abs() + 0.5)
            acc = dz.clone()
            ops.gemm(gh, wk, M, Cin, Cout, out=acc, b_mode=ops.KROW_TAPS, ldb=Cin, R=1, S=1, Cin=Cout, c_rowmap=rowmap, tile=tile,
                     zero_fill_pitch=W, accumulate=True, mask=ymask, relu_bwd=True, post_scale=ps, out2=d2)
            t = torch.where(ymask.float() > 0, 2 * dx.float(), torch.zeros_like(dx.float()))
            torch.testing.assert_close(d2.float(), t, **tol(dt))
            torch.testing.assert_close(acc.float(), t * ps, **tol(dt))
    torch.testing.assert_close(dx.float().view(n, H, W, Cin), xr.grad.permute(0, 2, 3, 1), **tol(dt))
    # wgrad: dW[co][(r,s,c)] = sum_m g[m,co] X[pix(m,r,s), c], split over the pixel reduction
    dw = torch.zeros(Cout, k * k * Cin, dtype=torch.float32, device=hw.dev)
    ops.gemm(gh, xh, Cout, k * k * Cin, M, out=dw, a_mode=ops.KROW, lda=Cout, b_mode=ops.KROW_GATHER, b_tab=tab,
             ldb=0, R=k, S=k, Cin=Cin, H=H, W=W, sH=W * Cin, sW=Cin, accumulate=True, split_k=2, tile=tile)
    torch.testing.assert_close(dw.view(Cout, k, k, Cin), wr.grad.permute(0, 2, 3, 1), **tol(dt))


def test_dropout_epilogue_is_consistent_and_unbiased(hw):
    M, N, K = 64, 64, 32
    x, w = torch.ones(M, K, device=hw.dev), torch.ones(N, K, device=hw.dev) / K
    a = torch.empty(M, N, device=hw.dev)
    b = torch.empty(M, N, device=hw.dev)
    ops.gemm(x, w, M, N, K, out=a, dropout_p=0.25, dropout_seed=7)
    ops.gemm(x, w, M, N, K, out=b, dropout_p=0.25, dropout_seed=7)
    assert torch.equal(a, b)
    kept = (a > 0).float().mean().item()
    assert abs(kept - 0.75) < 0.03
    torch.testing.assert_close(a[a > 0], torch.full_like(a[a > 0], 1 / 0.75))
