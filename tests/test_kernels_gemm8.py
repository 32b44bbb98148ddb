"""The 8-wave LDS-DMA kernels of cb_gemm (clipbert_amd/csrc/gemm8_impl.h: tiles 5 = 256x256, 6 = 128x256, 7 = 256x128; the three
K-loop schedules; K split through fp32 slabs + the reduce kernel) against plain PyTorch fp32 references of the same op --
forward, data gradient, weight gradient, convolutions, every epilogue the 4-wave kernels have.  Each case runs on the host
lane-level emulator (CPU suite) and, marked `gpu`, through the real library on an MI355X.

The emulator runs every case twice where the LDS ring matters: with LDS-DMA bytes landing at issue (default) and with
EMUL_DMA_LAZY=1 -- landing only when the issuing lane's counted s_waitcnt vmcnt(N) retires them -- so a fragment read that is
not covered by the wait + barrier of its tile fails deterministically (second run in a child process: the switch is read once)."""
import os
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F

from clipbert_amd import ops

BF = torch.bfloat16
TOL = dict(rtol=2e-2, atol=3e-2)
TILES = [5, 6, 7]
SCHEDULES = [1, 2, 3]            # cb_gemm_desc.schedule: forces schedule 0 / 1 / 2 of gemm8_impl.h


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def ws(hw, nbytes=8 << 20):
    return torch.empty(nbytes // 4, dtype=torch.float32, device=hw.dev)


@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("sched", SCHEDULES)
def test_linear_forward_epilogues(hw, tile, sched):
    # ragged M, N edges; 7 K tiles + a K tail (ring wraps more than twice for every NST)
    M, N, K = 300, 264, 64 * 7 + 40
    x, w, b = hw(rnd(M, K, seed=1).to(BF)), hw(rnd(N, K, seed=2, scale=0.1).to(BF)), hw(rnd(N, seed=3))
    res = hw(rnd(M, N, seed=4).to(BF))
    out = torch.empty(M, N, dtype=BF, device=hw.dev)
    pre = torch.empty(M, N, dtype=BF, device=hw.dev)
    ops.gemm(x, w, M, N, K, out=out, shift=b, act=ops.ACT_GELU, residual=res, out2=pre, tile=tile, schedule=sched)
    ref_pre = x.float() @ w.float().t() + b
    torch.testing.assert_close(pre.float(), ref_pre, **TOL)
    torch.testing.assert_close(out.float(), F.gelu(ref_pre) + res.float(), **TOL)
    # short reductions: fewer K tiles than ring stages
    for k2 in (64, 128, 8):
        o2 = torch.empty(M, N, dtype=BF, device=hw.dev)
        ops.gemm(x[:, :k2], w[:, :k2], M, N, k2, out=o2, lda=K, ldb=K, tile=tile, schedule=sched)
        torch.testing.assert_close(o2.float(), x[:, :k2].float() @ w[:, :k2].float().t(), **TOL)


@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("sched", [1, 3])
def test_split_k_slabs_any_epilogue(hw, tile, sched):
    """K split of the 8-wave tiles: partial products in workspace slabs, summed in index order by the reduce kernel, which also
    applies the whole epilogue (here: bias + dropout + residual into a bf16 output -- impossible with the atomics path)."""
    M, N, K = 200, 136, 64 * 5
    x, w, b = hw(rnd(M, K, seed=1).to(BF)), hw(rnd(N, K, seed=2, scale=0.1).to(BF)), hw(rnd(N, seed=3))
    res = hw(rnd(M, N, seed=4).to(BF))
    buf = ws(hw)
    ref = x.float() @ w.float().t() + b
    outs = []
    for split in (1, 2, 3, 5, 9):                        # 9 > K tiles: clamped
        out = torch.empty(M, N, dtype=BF, device=hw.dev)
        ops.gemm(x, w, M, N, K, out=out, shift=b, residual=res, tile=tile, schedule=sched, split_k=split, splitk_ws=buf)
        torch.testing.assert_close(out.float(), ref + res.float(), **TOL)
        outs.append(out)
    # deterministic: same split, same bits
    again = torch.empty(M, N, dtype=BF, device=hw.dev)
    ops.gemm(x, w, M, N, K, out=again, shift=b, residual=res, tile=tile, schedule=sched, split_k=3, splitk_ws=buf)
    assert torch.equal(again, outs[2])
    # dropout mask of the split path == the unsplit kernel's (same seed, same (row, column) addressing)
    a = torch.empty(M, N, dtype=BF, device=hw.dev)
    c = torch.empty(M, N, dtype=BF, device=hw.dev)
    ops.gemm(x, w, M, N, K, out=a, dropout_p=0.3, dropout_seed=11, tile=tile, schedule=sched)
    ops.gemm(x, w, M, N, K, out=c, dropout_p=0.3, dropout_seed=11, tile=tile, schedule=sched, split_k=2, splitk_ws=buf)
    assert torch.equal(a == 0, c == 0)
    # no workspace: the explicit split falls back to the 4-wave atomics kernel (fp32 accumulate form only)
    dwa = torch.zeros(M, N, dtype=torch.float32, device=hw.dev)
    ops.gemm(x, w, M, N, K, out=dwa, accumulate=True, tile=tile, split_k=2, splitk_ws=torch.empty(4, dtype=torch.float32, device=hw.dev))
    torch.testing.assert_close(dwa, x.float() @ w.float().t(), **TOL)


@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("sched", SCHEDULES)
def test_dgrad_wgrad_rowsum_batched(hw, tile, sched):
    M, N, K = 264, 200, 64 * 4 + 24                      # tokens, out features, in features
    x, w, g = hw(rnd(M, K, seed=1).to(BF)), hw(rnd(N, K, seed=2, scale=0.1).to(BF)), hw(rnd(M, N, seed=3).to(BF))
    # dX = g W (W as [N][K]: reduction index outermost -> transpose-read image), GELU' fused
    pre = hw(rnd(M, K, seed=7).to(BF))
    dx = torch.empty(M, K, dtype=BF, device=hw.dev)
    ops.gemm(g, w, M, K, N, out=dx, b_mode=ops.KROW, tile=tile, schedule=sched, gelu_grad_pre=pre)
    pr = pre.float().requires_grad_(True)
    F.gelu(pr).backward(g.float() @ w.float())
    torch.testing.assert_close(dx.float(), pr.grad, **TOL)
    # round 5: forward with C2 = gelu'(pre) (CB_ACT_GELU_SAVE_GRAD), data gradient x the stored derivative (CB_ACT_SAVED_GRAD)
    wf, bf = hw(rnd(K, N, seed=9, scale=0.1).to(BF)), hw(rnd(K, seed=10))
    y, dsave, y0 = (torch.empty(M, K, dtype=BF, device=hw.dev) for _ in range(3))
    ops.gemm(g, wf, M, K, N, out=y, shift=bf, act=ops.ACT_GELU_SAVE_GRAD, out2=dsave, tile=tile, schedule=sched)
    ops.gemm(g, wf, M, K, N, out=y0, shift=bf, act=ops.ACT_GELU, tile=tile, schedule=sched)
    assert torch.equal(y, y0) or (y.float() - y0.float()).abs().max() <= 1e-6 * max(1.0, float(y0.float().abs().max()))
    z = (g.float() @ wf.float().t() + bf).requires_grad_(True)
    F.gelu(z).sum().backward()
    torch.testing.assert_close(dsave.float(), z.grad, **TOL)
    dx2 = torch.empty(M, K, dtype=BF, device=hw.dev)
    ops.gemm(g, w, M, K, N, out=dx2, b_mode=ops.KROW, tile=tile, schedule=sched, gelu_grad_pre=dsave, act=ops.ACT_SAVED_GRAD)
    torch.testing.assert_close(dx2.float(), (g.float() @ w.float()) * dsave.float(), **TOL)
    # dW = g^T x (+ bias gradient as row sums on the matrix core), unsplit and split through slabs
    buf = ws(hw)
    tolw = dict(rtol=2e-2, atol=8e-2)
    for split in (1, 3):
        dw = torch.ones(N, K, dtype=torch.float32, device=hw.dev)
        db = torch.ones(N, dtype=torch.float32, device=hw.dev)
        ops.gemm(g, x, N, K, M, out=dw, a_mode=ops.KROW, b_mode=ops.KROW, accumulate=True, split_k=split, tile=tile, schedule=sched,
                 a_rowsum=db, splitk_ws=buf)
        torch.testing.assert_close(dw, 1.0 + g.float().t() @ x.float(), **tolw)
        torch.testing.assert_close(db, 1.0 + g.float().sum(0), **tolw)
    # strided-batched weight gradients (all encoder layers in one launch), split
    nb = 3
    xb, gb = hw(rnd(nb, M, K, seed=11).to(BF)), hw(rnd(nb, M, N, seed=13).to(BF))
    for split in (1, 2):
        dwb = torch.zeros(nb, N * K + 16, dtype=torch.float32, device=hw.dev)
        dbb = torch.zeros(nb, N + 8, dtype=torch.float32, device=hw.dev)
        ops.gemm(gb, xb, N, K, M, out=dwb, ldc=K, lda=N, ldb=K, a_mode=ops.KROW, b_mode=ops.KROW, accumulate=True, split_k=split,
                 a_rowsum=dbb, batch=nb, batch_strides=(M * N, M * K, dwb.stride(0), dbb.stride(0)), tile=tile, schedule=sched, splitk_ws=buf)
        for b in range(nb):
            torch.testing.assert_close(dwb[b, :N * K].view(N, K), gb[b].float().t() @ xb[b].float(), **tolw)
            torch.testing.assert_close(dbb[b, :N], gb[b].float().sum(0), **tolw)
        assert dwb[:, N * K:].abs().max() == 0 and dbb[:, N:].abs().max() == 0


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("sched", [1, 3])
@pytest.mark.parametrize("k,stride,pad,H,W,Cin,Cout", [(3, 1, 1, 9, 11, 64, 136), (3, 1, 1, 6, 7, 64, 128), (1, 2, 0, 8, 6, 64, 72), (1, 1, 0, 7, 5, 128, 64)])
def test_conv_forward_backward(hw, tile, sched, k, stride, pad, H, W, Cin, Cout):
    n = 3
    x = hw(rnd(n, Cin, H, W, seed=1).to(BF))
    w = hw(rnd(Cout, Cin, k, k, seed=2, scale=0.05).to(BF))
    scale, shift = hw(rnd(Cout, seed=3).abs() + 0.5), hw(rnd(Cout, seed=4))
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    xh, wk = _nhwc(x), w.permute(0, 2, 3, 1).contiguous()
    M = n * OH * OW
    tab = ops.build_pixel_table(n, OH, OW, stride, pad, H * W * Cin, W * Cin, Cin, x.device)
    res = hw(rnd(M, Cout, seed=5).to(BF))
    buf = ws(hw)
    y = torch.empty(M, Cout, dtype=BF, device=hw.dev)
    ops.gemm(xh, wk, M, Cout, k * k * Cin, out=y, a_mode=ops.ROWK_GATHER, a_tab=tab, lda=0, ldb=k * k * Cin,
             R=k, S=k, Cin=Cin, H=H, W=W, sH=W * Cin, sW=Cin, scale=scale, shift=shift, residual=res, relu_after=True, tile=tile, schedule=sched)
    xr, wr = x.float().requires_grad_(True), w.float().requires_grad_(True)
    conv = F.conv2d(xr, wr, None, stride, pad)
    ref = F.relu(conv * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + res.float().view(n, OH, OW, Cout).permute(0, 3, 1, 2))
    torch.testing.assert_close(y.float().view(n, OH, OW, Cout), ref.permute(0, 2, 3, 1), **TOL)
    if k == 3:                                           # the same convolution with its reduction split in 3 (slabs)
        y3 = torch.empty(M, Cout, dtype=BF, device=hw.dev)
        ops.gemm(xh, wk, M, Cout, k * k * Cin, out=y3, a_mode=ops.ROWK_GATHER, a_tab=tab, lda=0, ldb=k * k * Cin, R=k, S=k, Cin=Cin, H=H, W=W,
                 sH=W * Cin, sW=Cin, scale=scale, shift=shift, residual=res, relu_after=True, tile=tile, schedule=sched, split_k=3, splitk_ws=buf)
        torch.testing.assert_close(y3.float(), y.float(), **TOL)

    g = hw(rnd(n, Cout, OH, OW, seed=6).to(BF))
    conv.backward(g.float())
    gh = _nhwc(g).view(M, Cout)
    dx = torch.zeros(n * H * W, Cin, dtype=BF, device=hw.dev)
    if stride == 1:
        tab_in = ops.build_pixel_table(n, H, W, 1, k - 1 - pad, OH * OW * Cout, OW * Cout, Cout, x.device)
        ops.gemm(gh, wk, n * H * W, Cin, k * k * Cout, out=dx, a_mode=ops.ROWK_GATHER, a_tab=tab_in, lda=0,
                 b_mode=ops.KROW_TAPS, ldb=k * k * Cin, R=k, S=k, Cin=Cout, H=OH, W=OW, sH=OW * Cout, sW=Cout,
                 flip_taps=True, tile=tile, schedule=sched)
    else:
        rowmap = (torch.arange(n).view(n, 1, 1) * H * W + (torch.arange(OH) * stride).view(1, OH, 1) * W
                  + (torch.arange(OW) * stride).view(1, 1, OW)).reshape(-1).int().to(hw.dev)
        ymask = hw(rnd(n * H * W, Cin, seed=9).to(BF))
        ps = hw(rnd(Cin, seed=10).abs() + 0.5)
        # strided 1x1 data gradient with the fused ReLU x FrozenBN backward, two outputs, own zeros (zero_fill_pitch)
        d1 = torch.full((n * H * W, Cin), float("nan"), dtype=BF, device=hw.dev)
        d2 = torch.full((n * H * W, Cin), float("nan"), dtype=BF, device=hw.dev)
        ops.gemm(gh, wk, M, Cin, Cout, out=d1, b_mode=ops.KROW_TAPS, ldb=Cin, R=1, S=1, Cin=Cout, c_rowmap=rowmap, tile=tile, schedule=sched,
                 zero_fill_pitch=W, mask=ymask, relu_bwd=True, post_scale=ps, out2=d2)
        t = torch.where(ymask.float() > 0, xr.grad.permute(0, 2, 3, 1).reshape(n * H * W, Cin), torch.zeros(n * H * W, Cin, device=hw.dev))
        torch.testing.assert_close(d2.float(), t, **TOL)
        torch.testing.assert_close(d1.float(), t * ps, **TOL)
        ops.gemm(gh, wk, M, Cin, Cout, out=dx, b_mode=ops.KROW_TAPS, ldb=Cin, R=1, S=1, Cin=Cout, c_rowmap=rowmap, tile=tile, schedule=sched)
    torch.testing.assert_close(dx.float().view(n, H, W, Cin), xr.grad.permute(0, 2, 3, 1), **TOL)
    # wgrad over the pixel reduction, split through slabs
    for split in (1, 2):
        dw = torch.zeros(Cout, k * k * Cin, dtype=torch.float32, device=hw.dev)
        ops.gemm(gh, xh, Cout, k * k * Cin, M, out=dw, a_mode=ops.KROW, lda=Cout, b_mode=ops.KROW_GATHER, b_tab=tab,
                 ldb=0, R=k, S=k, Cin=Cin, H=H, W=W, sH=W * Cin, sW=Cin, accumulate=True, split_k=split, tile=tile, schedule=sched, splitk_ws=buf)
        torch.testing.assert_close(dw.view(Cout, k, k, Cin), wr.grad.permute(0, 2, 3, 1), rtol=2e-2, atol=8e-2)


def test_auto_configuration_of_untabled_shapes(hw):
    """tile = 0 on shapes that are in no table: the launch-cost model's pick (here: 8-wave tiles with tens of K slabs for thin outputs
    over a long reduction -- tests/test_gemm_plan.py says what it picks) goes through the same selection -> workspace -> reduce glue
    the table's entries use; results against torch, and the same bits as the explicitly requested configuration."""
    import ctypes as C
    from clipbert_amd import _lib
    buf = ws(hw, 32 << 20)
    for form, (M, N, K) in (("fwd", (128, 512, 8192)), ("dgrad", (384, 256, 8192))):
        a = hw(rnd(M, K, seed=1).to(BF))
        b = hw(rnd(N, K, seed=2, scale=0.05).to(BF)) if form == "fwd" else hw(rnd(K, N, seed=2, scale=0.05).to(BF))
        bias = hw(rnd(N, seed=3))
        kw = dict(shift=bias, act=ops.ACT_RELU) if form == "fwd" else dict(b_mode=ops.KROW, ldb=N)
        ref = a.float() @ (b.float().t() if form == "fwd" else b.float())
        ref = torch.relu(ref + bias) if form == "fwd" else ref
        auto = torch.empty(M, N, dtype=BF, device=hw.dev)
        ops.gemm(a, b, M, N, K, out=auto, splitk_ws=buf, **kw)
        torch.testing.assert_close(auto.float(), ref, **TOL)
        d = _lib.GemmDesc()
        C.memset(C.byref(d), 0, C.sizeof(d))
        d.dtype, d.M, d.N, d.K, d.b_mode, d.batch, d.split_k = 1, M, N, K, (0 if form == "fwd" else 2), 1, 1
        d.A = d.B = d.C = 1 << 20
        d.a_bytes = d.b_bytes = 1 << 30
        d.lda, d.ldb, d.ldc = K, (K if form == "fwd" else N), N
        d.splitk_ws, d.splitk_ws_bytes = 1 << 20, 32 << 20
        out4 = (C.c_int32 * 4)()
        _lib.check(_lib.get().cb_gemm_plan(C.byref(d), 1, out4), "cb_gemm_plan")
        tile, split, sched, _ = out4
        assert tile >= 5 and split > 1, (form, tuple(out4))              # (the regime this test is about)
        same = torch.empty(M, N, dtype=BF, device=hw.dev)
        ops.gemm(a, b, M, N, K, out=same, splitk_ws=buf, tile=tile, split_k=split, schedule=sched, **kw)
        assert torch.equal(same, auto)
        # a workspace too small for any slab split: the same call must still be right (no split on offer)
        nows = torch.empty(M, N, dtype=BF, device=hw.dev)
        ops.gemm(a, b, M, N, K, out=nows, splitk_ws=torch.empty(4, dtype=torch.float32, device=hw.dev), **kw)
        torch.testing.assert_close(nows.float(), ref, **TOL)


def test_unsupported_shapes_fall_back_to_the_4_wave_kernels(hw):
    """tile 5-7 on a problem the 8-wave kernels do not cover (N % 8 != 0, fp32) must still give the right answer."""
    M, N, K = 70, 36, 72
    x, w = hw(rnd(M, K, seed=1).to(BF)), hw(rnd(N, K, seed=2, scale=0.1).to(BF))
    out = torch.empty(M, N - 1, dtype=BF, device=hw.dev)
    ops.gemm(x, w[:N - 1], M, N - 1, K, out=out, tile=6)
    torch.testing.assert_close(out.float(), x.float() @ w[:N - 1].float().t(), **TOL)
    xf, wf = x.float(), w.float()
    of = torch.empty(M, N, dtype=torch.float32, device=hw.dev)
    ops.gemm(xf, wf, M, N, K, out=of, tile=5)
    torch.testing.assert_close(of, xf @ wf.t(), rtol=1e-4, atol=1e-4)


def test_lazy_dma_emulation():
    """Re-run the emulator cases of this file with LDS-DMA data landing as LATE as the counted waits allow."""
    env = dict(os.environ, EMUL_DMA_LAZY="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "not gpu", "-k",
                        "not lazy_dma", "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=3000)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_inference_gelu_body_equals_the_generic_epilogue_bit_for_bit(hw):
    """round 6: bias + GELU WITHOUT the stored derivative (BertIntermediate in inference) has a specialised epilogue body (combination 17)
    on the 128x256 / 256x128 8-wave tiles and the 4-wave tiles; the 256x256 forward kernel carries no specialised bodies (generic
    epilogue8).  Same packed GELU everywhere: all five tiles agree bit for bit, and with PyTorch within the bf16 bound."""
    M, N, K = 300, 264, 64 * 3 + 8
    x, w, b = hw(rnd(M, K, seed=1).to(BF)), hw(rnd(N, K, seed=2, scale=0.1).to(BF)), hw(rnd(N, seed=3))
    ref = F.gelu(x.float() @ w.float().t() + b)
    outs = []
    for tile in (5, 6, 7, 4, 2):
        y = torch.empty(M, N, dtype=BF, device=hw.dev)
        ops.gemm(x, w, M, N, K, out=y, shift=b, act=ops.ACT_GELU, tile=tile)
        torch.testing.assert_close(y.float(), ref, **TOL)
        outs.append(y)
    for y in outs[1:]:
        assert torch.equal(y, outs[0])
