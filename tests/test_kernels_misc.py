"""Every non-GEMM kernel of libclipbert_hip against plain PyTorch fp32 references (and the oracle's AdamW
restatement).  Each case runs on the host lane-level emulator build (CPU suite) and, marked `gpu`, on the
real library on an MI355X."""
import pytest
import torch
import torch.nn.functional as F

from clipbert_amd import ops
from oracle import clipbert_oracle as O

DT = [torch.float32, torch.bfloat16]


def tol(dt, f32=1e-5, bf=2e-2):
    return dict(rtol=bf, atol=bf) if dt == torch.bfloat16 else dict(rtol=f32 * 10, atol=f32)


DEV = [torch.device("cpu")]


@pytest.fixture(autouse=True)
def _device(hw):
    DEV[0] = hw.dev
    yield
    DEV[0] = torch.device("cpu")


def rnd(*shape, seed=0, scale=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).to(DEV[0])


def zeros(*shape, dtype=torch.float32):
    return torch.zeros(*shape, dtype=dtype, device=DEV[0])


def ones(*shape, dtype=torch.float32):
    return torch.ones(*shape, dtype=dtype, device=DEV[0])


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("D,rows", [(768, 37), (64, 37), (1280, 37), (768, 2350)])      # >= 1024 rows: the 16-wave backward
def test_layernorm_fwd_bwd(hw, dt, D, rows):
    x = rnd(rows, D, seed=1).to(dt)
    g, b = 1 + rnd(D, seed=2, scale=0.1), rnd(D, seed=3, scale=0.1)
    y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-12, save_stats=True)
    xr = x.float().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (D,), gr, br, 1e-12)
    torch.testing.assert_close(y.float(), ref, **tol(dt))
    dy = rnd(rows, D, seed=4).to(dt)
    ref.backward(dy.float())
    dgam, dbet = zeros(D), zeros(D)
    dx, _ = ops.layernorm_bwd(dy, x, g, mean, rstd, dgam, dbet)
    torch.testing.assert_close(dx.float(), xr.grad, **tol(dt, 1e-4))
    torch.testing.assert_close(dgam, gr.grad, **tol(dt, 1e-4, 5e-2))
    torch.testing.assert_close(dbet, br.grad, **tol(dt, 1e-4, 5e-2))


@pytest.mark.parametrize("dt", DT)
def test_embeddings_fwd_bwd(hw, dt):
    B, Lt, Hg, Wg, T, D, V = 4, 6, 2, 3, 2, 128, 50
    Lv = Hg * Wg
    L = Lt + Lv
    ids = torch.randint(0, V, (B, Lt), generator=torch.Generator().manual_seed(1)).to(DEV[0])
    ids[0, -1] = 0
    tabs = {k: rnd(n, D, seed=i).to(dt) for i, (k, n) in enumerate(dict(word=V, pos=16, typ=2, row=5, col=5, vtyp=1).items())}
    g1, b1, g2, b2 = 1 + rnd(D, seed=7, scale=0.1), rnd(D, seed=8, scale=0.1), 1 + rnd(D, seed=9, scale=0.1), rnd(D, seed=10, scale=0.1)
    grid = rnd(2, T, Hg, Wg, D, seed=11).to(dt)
    src_row = torch.tensor([0, 0, 1, 1], dtype=torch.int32, device=DEV[0])
    out = zeros(B * L, D, dtype=dt)
    pre = zeros(B * L, D, dtype=dt)
    mean, rstd = zeros(B * L), zeros(B * L)
    ops.text_embed_fwd(ids, tabs["word"], tabs["pos"], tabs["typ"], g1, b1, out, pre, mean, rstd, Lt, L, 1e-12)
    ops.visual_embed_fwd(grid, src_row, None, tabs["row"], tabs["col"], tabs["vtyp"], g2, b2, out, pre, mean, rstd, B, Lv, Lt, L, 1e-12)
    # reference with autograd
    f = {k: v.float().requires_grad_(True) for k, v in tabs.items()}
    gridr = grid.float().requires_grad_(True)
    te = f["word"][ids] + f["pos"][:Lt].unsqueeze(0) + f["typ"][0]
    gv = gridr[src_row.long()].mean(1) + f["row"][:Hg].view(1, Hg, 1, D) + f["col"][:Wg].view(1, 1, Wg, D)
    ve = gv.reshape(B, Lv, D) + f["vtyp"][0]
    pre_ref = torch.cat([te, ve], 1)
    ref = torch.cat([F.layer_norm(te, (D,), g1, b1, 1e-12), F.layer_norm(ve, (D,), g2, b2, 1e-12)], 1)
    torch.testing.assert_close(pre.float().view(B, L, D), pre_ref, **tol(dt))
    torch.testing.assert_close(out.float().view(B, L, D), ref, **tol(dt))
    # backward from d(pre)
    dpre = rnd(B * L, D, seed=12).to(dt)
    pre_ref.backward(dpre.float().view(B, L, D))
    dword, dpos, dtyp = zeros(V, D), zeros(16, D), zeros(2, D)
    ops.text_embed_bwd(dpre, ids, dword, dpos, dtyp[0], Lt, L, pad_id=0)
    ref_dword = f["word"].grad.clone()
    ref_dword[0] = 0          # padding_idx rows receive no gradient
    torch.testing.assert_close(dword, ref_dword, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(dpos, f["pos"].grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(dtyp, f["typ"].grad, rtol=1e-4, atol=1e-4)
    dgrid, drow, dcol, dvt = zeros(2, T, Hg, Wg, D), zeros(5, D), zeros(5, D), zeros(1, D)
    ops.visual_embed_bwd(dpre, src_row, None, dgrid, drow, dcol, dvt, B, Lv, Lt, L)
    torch.testing.assert_close(dgrid, gridr.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(drow, f["row"].grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(dcol, f["col"].grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(dvt, f["vtyp"].grad, rtol=1e-4, atol=1e-4)


def test_visual_embed_pixel_subsample(hw):
    B, Lt, Hg, Wg, T, D = 2, 3, 3, 3, 1, 64
    sel = torch.tensor([1, 4, 5, 8], dtype=torch.int32, device=DEV[0])
    Lv, L = 4, 7
    grid = rnd(B, T, Hg, Wg, D, seed=1)
    row, col, typ = rnd(4, D, seed=2), rnd(4, D, seed=3), rnd(1, D, seed=4)
    g, b = ones(D), zeros(D)
    out = zeros(B * L, D)
    ops.visual_embed_fwd(grid, None, sel, row, col, typ, g, b, out, None, None, None, B, Lv, Lt, L, 1e-12)
    sd = {"p.row_position_embeddings.weight": row.cpu(), "p.col_position_embeddings.weight": col.cpu(),
          "p.token_type_embeddings.weight": typ.cpu(), "p.LayerNorm.weight": g.cpu(), "p.LayerNorm.bias": b.cpu()}
    ref = O.visual_embeddings(sd, "p", grid.cpu(), 1e-12, sel.long().cpu())
    torch.testing.assert_close(out.view(B, L, D)[:, Lt:].cpu(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("L", [9, 41, 69, 112, 169, 192, 200])      # 65..192: LDS-resident MFMA kernels (bf16); 200: generic
def test_attention_fwd_bwd(hw, dt, L):
    B, H = 2, 2
    qkv = rnd(B * L, 3 * H * 64, seed=1).to(dt)
    mask = ones(B, L)
    mask[0, L - 3:] = 0
    mask[1, 2] = 0
    ctx, lse = ops.attention_fwd(qkv, mask, B, L, H, save_lse=True)
    x = qkv.float().requires_grad_(True)
    q, k, v = [t.view(B, L, H, 64).permute(0, 2, 1, 3) for t in x.view(B, L, 3, H * 64).unbind(2)]
    s = q @ k.transpose(-1, -2) / 8.0 + ((1 - mask) * -10000.0)[:, None, None, :]
    ref = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B * L, H * 64)
    torch.testing.assert_close(ctx.float(), ref, **tol(dt, 1e-5))
    dctx = rnd(B * L, H * 64, seed=2).to(dt)
    ref.backward(dctx.float())
    dqkv = ops.attention_bwd(qkv, mask, ctx, dctx, lse, B, L, H)
    torch.testing.assert_close(dqkv.float(), x.grad, **tol(dt, 1e-4, 3e-2))


@pytest.mark.parametrize("L", [20, 41, 64, 69, 169])
def test_attention_dropout_mfma_matches_generic(hw, L):
    """bf16 one-wave MFMA kernels vs the generic fp32 kernels on the same (bf16-valued) inputs and the same dropout
    stream: same mask indexing in forward and backward, P rounded to bf16 before P.V the only difference."""
    B, H, p = 2, 3, 0.3
    qkv16 = rnd(B * L, 3 * H * 64, seed=1).to(torch.bfloat16)
    dctx16 = rnd(B * L, H * 64, seed=2).to(torch.bfloat16)
    mask = ones(B, L)
    mask[0, L - 5:] = 0
    sp = torch.tensor([11], dtype=torch.int64, device=DEV[0])
    ctx16, lse16 = ops.attention_fwd(qkv16, mask, B, L, H, save_lse=True, dropout_p=p, dropout_seed=5, seed_ptr=sp)
    ctx32, lse32 = ops.attention_fwd(qkv16.float(), mask, B, L, H, save_lse=True, dropout_p=p, dropout_seed=5, seed_ptr=sp)
    torch.testing.assert_close(lse16, lse32, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(ctx16.float(), ctx32, rtol=2e-2, atol=2e-2)
    d16 = ops.attention_bwd(qkv16, mask, ctx16, dctx16, lse16, B, L, H, dropout_p=p, dropout_seed=5, seed_ptr=sp)
    d32 = ops.attention_bwd(qkv16.float(), mask, ctx32, dctx16.float(), lse32, B, L, H, dropout_p=p, dropout_seed=5, seed_ptr=sp)
    torch.testing.assert_close(d16.float(), d32, rtol=3e-2, atol=3e-2)
    # a different replay seed gives a different mask
    ctx_b, _ = ops.attention_fwd(qkv16, mask, B, L, H, dropout_p=p, dropout_seed=6, seed_ptr=sp)
    assert (ctx_b.float() - ctx16.float()).abs().max() > 1e-2


def test_cross_entropy_and_colsum_and_cast_and_act(hw):
    logits = rnd(9, 37, seed=1)
    labels = torch.randint(0, 37, (9,), generator=torch.Generator().manual_seed(2)).to(DEV[0])
    labels[3] = -100
    dloss = rnd(9, seed=3)
    loss, dl = ops.cross_entropy(logits, labels, dloss=dloss, want_grad=True)
    lr = logits.clone().requires_grad_(True)
    ref = F.cross_entropy(lr, labels, reduction="none", ignore_index=-100)
    torch.testing.assert_close(loss, ref, rtol=1e-5, atol=1e-5)
    ref.backward(dloss)
    torch.testing.assert_close(dl, lr.grad, rtol=1e-5, atol=1e-6)
    for dt in DT:
        for n_cols in (70, 72, 776):            # scalar fallback, vector path, several column blocks
            g = rnd(300, n_cols, seed=4).to(dt)
            out = ones(n_cols)
            ops.colsum(g, out)
            torch.testing.assert_close(out, 1 + g.float().sum(0), rtol=1e-4, atol=1e-3)
        for act, fn in [(ops.ACT_GELU, F.gelu), (ops.ACT_TANH, torch.tanh), (ops.ACT_RELU, F.relu)]:
            pre = rnd(1001, seed=5).requires_grad_(True)
            y = fn(pre)
            dy = rnd(1001, seed=6)
            y.backward(dy)
            ref_in = pre.detach() if act == ops.ACT_GELU else y.detach()
            dx = ops.act_bwd(act, dy.to(dt), ref_in.to(dt))
            torch.testing.assert_close(dx.float(), pre.grad, **tol(dt, 1e-5, 3e-2))
    src = rnd(1003, seed=7)
    torch.testing.assert_close(ops.cast(src, torch.empty(1003, dtype=torch.bfloat16, device=DEV[0])), src.bfloat16())


def test_adamw_matches_reference_restatement_with_clipping(hw):
    n = 1003
    p, g, m, v = rnd(n, seed=1), rnd(n, seed=2, scale=3.0), rnd(n, seed=3, scale=0.1), rnd(n, seed=4).abs() * 0.01
    sq = zeros(1)
    ops.sq_sum(g, sq)
    torch.testing.assert_close(sq[0], (g * g).sum(), rtol=1e-5, atol=1e-3)
    # the order-independent variant (what FusedAdamW uses): same value, bit-identical from call to call
    big = rnd(3 * 1024 * 1024 + 5, seed=9)
    ws = zeros(1024)
    outs = []
    for _ in range(3):
        o = zeros(1)
        ops.sq_sum(big, o, ws)
        outs.append(float(o[0]))
    assert outs[0] == outs[1] == outs[2]
    assert abs(outs[0] - float((big.double() ** 2).sum())) / outs[0] < 1e-5
    gc, total = O.clip_grad_norm([g.cpu()], 5.0)
    pr, mr, vr = O.adamw_step(p.cpu(), gc[0], m.cpu(), v.cpu(), step=3, lr=5e-5, beta1=0.9, beta2=0.98, eps=1e-6, weight_decay=1e-3)
    pr, mr, vr = pr.to(DEV[0]), mr.to(DEV[0]), vr.to(DEV[0])
    w16 = torch.empty(n, dtype=torch.bfloat16, device=DEV[0])
    p2, m2, v2 = p.clone(), m.clone(), v.clone()
    hp = torch.tensor(ops.adamw_hyper(5e-5, 0.9, 0.98, 1e-6, 1e-3, 3, max_norm=5.0), device=DEV[0])
    ops.adamw(p2, g, m2, v2, w16, hp, grad_sq_sum=sq)
    torch.testing.assert_close(p2, pr, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(m2, mr, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(v2, vr, rtol=1e-5, atol=1e-8)
    torch.testing.assert_close(w16, pr.bfloat16())
    # bf16 gradients (the reduced data-parallel wire image) == fp32 gradients holding the same bf16 values, bit for bit
    g16 = g.bfloat16()
    sq_a, sq_b, ws = zeros(1), zeros(1), zeros(1024)
    ops.sq_sum(g16, sq_a, ws)
    ops.sq_sum(g16.float(), sq_b, ws)
    assert float(sq_a[0]) == float(sq_b[0])
    pa, ma, va, pb, mb, vb = p.clone(), m.clone(), v.clone(), p.clone(), m.clone(), v.clone()
    ops.adamw(pa, g16, ma, va, None, hp, grad_sq_sum=sq_a)
    ops.adamw(pb, g16.float(), mb, vb, None, hp, grad_sq_sum=sq_b)
    torch.testing.assert_close(pa, pb, rtol=0, atol=0)
    torch.testing.assert_close(va, vb, rtol=0, atol=0)


@pytest.mark.parametrize("dt", DT)
def test_pool_stem_and_relu_bwd(hw, dt):
    x = rnd(2, 8, 9, 7, seed=1).to(dt)                  # NCHW reference
    xh = x.permute(0, 2, 3, 1).contiguous()
    y = ops.maxpool_fwd(xh, 3, 2, 1)
    torch.testing.assert_close(y.float().permute(0, 3, 1, 2), F.max_pool2d(x.float(), 3, 2, 1))
    y2 = ops.maxpool_fwd(xh, 2, 2, 0, relu=True)
    xr = x.float().requires_grad_(True)
    ref2 = F.relu(F.max_pool2d(xr, 2, 2))
    torch.testing.assert_close(y2.float().permute(0, 3, 1, 2), ref2)
    dy = rnd(*ref2.shape, seed=2).to(dt)
    ref2.backward(dy.float())
    dx = ops.maxpool2_bwd(xh, y2, dy.permute(0, 2, 3, 1).contiguous(), relu=True)
    torch.testing.assert_close(dx.float().permute(0, 3, 1, 2), xr.grad)
    # stem pack: fp32 normalised input and fused uint8 path
    img = torch.randint(0, 256, (2, 3, 6, 5), generator=torch.Generator().manual_seed(3), dtype=torch.uint8).to(DEV[0])
    mean, std = (123.675, 116.28, 103.53), (1.0, 2.0, 0.5)
    norm = ops.image_norm(img, mean, std)
    ref_norm = (img.float() - torch.tensor(mean, device=DEV[0]).view(1, 3, 1, 1)) / torch.tensor(std, device=DEV[0]).view(1, 3, 1, 1)
    torch.testing.assert_close(norm, ref_norm, rtol=1e-6, atol=1e-5)
    ref_pack = zeros(2, 12, 11, 4)
    ref_pack[:, 3:9, 3:8, :3] = ref_norm[:, [2, 1, 0]].permute(0, 2, 3, 1)
    for packed in (ops.stem_pack(norm, dt, 3), ops.stem_pack(img, dt, 3, mean, std)):
        torch.testing.assert_close(packed.float(), ref_pack.to(dt).float(), **tol(dt, 1e-5, 1.0))
    # relu + frozen-BN backward
    yy, dyy = rnd(10, 16, seed=4).to(dt), rnd(10, 16, seed=5).to(dt)
    sc, sc2 = rnd(16, seed=6), rnd(16, seed=7)
    g, dz, g2 = ops.relu_scale_bwd(dyy, yy, sc, True, sc2)
    refdz = dyy.float() * (yy.float() > 0)
    torch.testing.assert_close(dz.float(), refdz.to(dt).float())
    torch.testing.assert_close(g.float(), (refdz * sc).to(dt).float(), **tol(dt))
    torch.testing.assert_close(g2.float(), (refdz * sc2).to(dt).float(), **tol(dt))


def test_dropout_kernel_and_seed_pointer(hw):
    x = ones(5001)
    a = ops.dropout(x, 0.1, seed=3)
    b = ops.dropout(x, 0.1, seed=1, seed_ptr=torch.tensor([2], dtype=torch.int64, device=DEV[0]))
    assert torch.equal(a, b)                       # seed + *seed_ptr
    assert abs((a > 0).float().mean().item() - 0.9) < 0.02
    torch.testing.assert_close(a[a > 0], torch.full_like(a[a > 0], 1 / 0.9))
    c = ops.dropout(x, 0.1, seed=4)
    assert not torch.equal(a, c)


@pytest.mark.parametrize("N", [72, 68, 200])
def test_dropout_masks_agree_across_kernels(hw, N):
    """One mask stream per site whatever kernel draws it: the GEMM epilogue (8-wide, 4-wide and element paths), the
    LayerNorm backward (which re-applies the forward's mask to the gradient) and cb_dropout must drop the same elements."""
    M, K, p, seed = 150, 32, 0.3, 9
    sp = torch.tensor([5], dtype=torch.int64, device=DEV[0])
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2)
    plain = torch.empty(M, N, device=DEV[0])
    ops.gemm(x, w, M, N, K, out=plain)
    ref = ops.dropout(plain.view(-1), p, seed=seed, seed_ptr=sp).view(M, N)
    for tile in (1, 2):
        got = torch.empty(M, N, device=DEV[0])
        ops.gemm(x, w, M, N, K, out=got, dropout_p=p, dropout_seed=seed, seed_ptr=sp, tile=tile)
        torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-5)
    if N % 8 == 0:
        D = N
        xx, dy = rnd(M, D, seed=3), rnd(M, D, seed=4)
        g = 1 + rnd(D, seed=5, scale=0.1)
        mean, var = xx.mean(-1), xx.var(-1, unbiased=False)
        rstd = (var + 1e-12).rsqrt()
        dg, db = zeros(D), zeros(D)
        dx, dx2 = ops.layernorm_bwd(dy, xx, g, mean, rstd, dg, db, dropout_p=p, dropout_seed=seed, seed_ptr=sp)
        torch.testing.assert_close(dx2, ops.dropout(dx.view(-1), p, seed=seed, seed_ptr=sp).view(M, D), rtol=1e-6, atol=1e-6)


def test_layernorm_bwd_row_segments(hw):
    B, L, Lt, D = 3, 7, 4, 64
    x, dy = rnd(B * L, D, seed=1), rnd(B * L, D, seed=2)
    g = 1 + rnd(D, seed=3, scale=0.1)
    mean, var = x.mean(-1), x.var(-1, unbiased=False)
    rstd = (var + 1e-12).rsqrt()
    dx = torch.zeros_like(x)
    dg, db = zeros(D), zeros(D)
    ops.layernorm_bwd(dy, x, g, mean, rstd, dg, db, dx=dx, rows=B * Lt, seg=(Lt, L, 0))
    xr = x.clone().requires_grad_(True)
    gr = g.clone().requires_grad_(True)
    y = F.layer_norm(xr, (D,), gr, zeros(D), 1e-12)
    sel = zeros(B, L, 1)
    sel[:, :Lt] = 1
    y.backward(dy * sel.view(B * L, 1))
    torch.testing.assert_close(dx, xr.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dg, gr.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("rows", [37, 1100])
def test_layernorm_bwd_partials_equal_atomics_path(hw, dt, rows):
    """cb_layernorm_bwd_part + cb_ln_partials_reduce (deterministic, no atomics) vs cb_layernorm_bwd: same dx bit for bit, parameter
    gradients equal up to summation order; two jobs reduced in one launch onto offsets of a flat buffer; repeatable bit for bit."""
    D, p, seed = 192, 0.1, 77
    g = 1 + rnd(D, seed=3, scale=0.1)
    flat = zeros(5 * D + 8)
    offs_g = torch.tensor([8, 8 + 2 * D], dtype=torch.int64, device=DEV[0])
    offs_b = torch.tensor([8 + D, 8 + 3 * D], dtype=torch.int64, device=DEV[0])
    nb = ops.ln_part_blocks(rows)
    part = torch.full((2, nb, 2, D), float("nan"), dtype=torch.float32, device=DEV[0])       # every slot must be written
    refs = []
    for job in range(2):
        x, dy = rnd(rows, D, seed=10 + job).to(dt), rnd(rows, D, seed=20 + job).to(dt)
        mean, var = x.float().mean(-1), x.float().var(-1, unbiased=False)
        rstd = (var + 1e-12).rsqrt()
        dg, db = zeros(D), zeros(D)
        dx_a, dx2_a = ops.layernorm_bwd(dy, x, g, mean, rstd, dg, db, dropout_p=p, dropout_seed=seed)
        dx_p, dx2_p = ops.layernorm_bwd_part(dy, x, g, mean, rstd, part[job], dropout_p=p, dropout_seed=seed)
        assert torch.equal(dx_a, dx_p) and torch.equal(dx2_a, dx2_p)
        refs.append((dg, db))
    flat[8 + 4 * D:] = 1.0                                     # neighbours stay untouched
    ops.ln_partials_reduce(part, flat, offs_g, offs_b)
    ops.ln_partials_reduce(part, flat, offs_g, offs_b)         # accumulates
    for job, (dg, db) in enumerate(refs):
        torch.testing.assert_close(flat[8 + 2 * job * D: 8 + (2 * job + 1) * D], 2 * dg, **tol(dt, 1e-4, 5e-2))
        torch.testing.assert_close(flat[8 + (2 * job + 1) * D: 8 + (2 * job + 2) * D], 2 * db, **tol(dt, 1e-4, 5e-2))
    assert float(flat[:8].abs().max()) == 0 and bool((flat[8 + 4 * D:] == 1).all())
    again = zeros(5 * D + 8)
    ops.ln_partials_reduce(part, again, offs_g, offs_b)
    ops.ln_partials_reduce(part, again, offs_g, offs_b)
    assert torch.equal(again[:8 + 4 * D], flat[:8 + 4 * D])


@pytest.mark.parametrize("off,n", [(0, 1), (1, 1), (3, 40), (0, 4096), (5, 4099), (7, 70001), (0, 0)])
def test_zero_fill_touches_exactly_its_range(hw, off, n):
    """cb_zero on a byte range of any alignment: the range becomes zero, its neighbours keep their values (the zero fills of the
    step -- optimizer.zero_grad(), scatter targets, the squared-norm word -- go through it instead of torch fill kernels)"""
    buf = torch.full((off + n + 33,), 0x5A, dtype=torch.uint8, device=DEV[0])
    ops.zero_(buf[off:off + n])
    want = torch.full_like(buf, 0x5A)
    want[off:off + n] = 0
    assert torch.equal(buf, want)
    assert float(ops.zeros((3, 5), torch.float32, DEV[0]).abs().sum()) == 0.0
    g = ones(1000)
    ops.zero_(g[64:512])
    assert float(g.sum()) == 1000 - 448 and float(g[64:512].abs().sum()) == 0.0


def test_head_losses_and_retrieval_scores_against_the_oracle(hw):
    """cb_head_loss (MSE / BCE-with-logits / sigmoid margin ranking, forward + backward) and cb_retrieval_scores against the ORACLE's
    restatement of the reference's heads (oracle/clipbert_oracle.py: retrieval_loss, sequence_classification_forward; pinned to
    src/modeling/modeling.py by tests/test_oracle_vs_reference.py) with autograd through it."""
    from clipbert_amd import clips
    from clipbert_amd import modeling as M
    # sigmoid margin ranking: 5 videos x (1 positive + 3 negatives)
    x = rnd(5, 4, seed=1, scale=2.0)
    xr = x.detach().cpu().clone().requires_grad_(True)
    cfg = dict(loss_type="rank", margin=0.3, num_labels=1)
    want = O.retrieval_loss(xr.view(-1, 1), torch.zeros(20, dtype=torch.long), cfg, sample_size=5)
    up = rnd(5, 3, seed=2).abs()
    (want * up.cpu()).sum().backward()
    xg = x.clone().requires_grad_(True)
    got = M.head_loss_none(ops.LOSS_RANK, xg, group=4, margin=0.3)
    (got * up).sum().backward()
    torch.testing.assert_close(got.detach().cpu(), want.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(xg.grad.cpu(), xr.grad.view(5, 4), rtol=1e-5, atol=1e-6)
    assert (got == 0).any() and (got > 0).any()                        # both sides of the clamp are exercised
    # BCE with logits on soft targets, MSE
    lg, tg = rnd(7, 11, seed=3, scale=3.0), rnd(7, 11, seed=4).sigmoid()
    for kind, ref in ((ops.LOSS_BCE, lambda a, b: F.binary_cross_entropy_with_logits(a, b, reduction="none")),
                      (ops.LOSS_MSE, lambda a, b: F.mse_loss(a, b, reduction="none"))):
        a = lg.detach().cpu().clone().requires_grad_(True)
        w = ref(a, tg.cpu())
        up2 = rnd(7, 11, seed=5)
        (w * up2.cpu()).sum().backward()
        b = lg.clone().requires_grad_(True)
        g = M.head_loss_none(kind, b, tg)
        (g * up2).sum().backward()
        torch.testing.assert_close(g.detach().cpu(), w.detach(), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(b.grad.cpu(), a.grad, rtol=1e-5, atol=1e-6)
    # extreme logits: the BCE stays finite (log1p(exp(-|x|)) form)
    big = torch.tensor([[80.0, -80.0, 0.0]], device=DEV[0])
    l, _ = ops.head_loss(ops.LOSS_BCE, big, torch.tensor([[0.0, 1.0, 0.5]], device=DEV[0]))
    torch.testing.assert_close(l.cpu(), torch.tensor([[80.0, 80.0, 0.6931472]]), rtol=1e-6, atol=1e-6)
    # inference scores (run_video_retrieval.py:681-690), rounded to 4 places by the caller
    two, one = rnd(9, 2, seed=6, scale=2.0), rnd(9, 1, seed=7, scale=2.0)
    assert clips.retrieval_scores(two) == [round(v, 4) for v in torch.softmax(two.cpu(), 1)[:, 1].tolist()]
    assert clips.retrieval_scores(one) == [round(v, 4) for v in torch.sigmoid(one.cpu()).view(-1).tolist()]


def test_zero_many_ranges_any_alignment(hw):
    """cb_zero_ranges: up to four ranges per launch (ops.zero_many chunks longer lists), unaligned heads / tails, neighbours untouched"""
    base = hw(torch.arange(1, 20001, dtype=torch.float32))
    buf = base.clone()
    u8 = hw(torch.full((1000,), 7, dtype=torch.uint8))
    views = [buf[3:1000], buf[1001:1002], buf[5000:12345], buf[19990:], buf[2000:2003]]
    ops.zero_many(views + [u8[5:998], None, buf[0:0]])
    want = base.clone()
    for a, b in ((3, 1000), (1001, 1002), (5000, 12345), (19990, 20000), (2000, 2003)):
        want[a:b] = 0
    assert torch.equal(buf.cpu(), want.cpu())
    assert u8[:5].eq(7).all() and u8[998:].eq(7).all() and u8[5:998].eq(0).all()
