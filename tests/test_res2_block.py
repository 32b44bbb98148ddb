"""cb_res2_block (one res2 bottleneck block in one launch) against (a) the same block computed op by op in fp32 PyTorch with
bf16 rounding at the points where the UNFUSED path stores (y1, y2, the projection-shortcut branch, the block output) and (b) the unfused
product path itself (three / four cb_gemm launches through clipbert_amd.modeling._conv_fwd) -- on the host emulator and, marked gpu, on
the MI355X.  Image sizes that are and are not multiples of the 8 x 8 tile, one and several tiles per workgroup."""
import pytest
import torch
import torch.nn.functional as F

from clipbert_amd import ops

DEV = [torch.device("cpu")]


@pytest.fixture(autouse=True)
def _device(hw):
    DEV[0] = hw.dev
    yield
    DEV[0] = torch.device("cpu")


def _gen(seed):
    return torch.Generator().manual_seed(seed)


def _bf(x):
    return x.to(torch.bfloat16).float()


def _make(cin, seed):
    g = _gen(seed)
    w1 = torch.randn(64, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5
    w2 = torch.randn(64, 64, 3, 3, generator=g) * (2.0 / 576) ** 0.5
    w3 = torch.randn(256, 64, 1, 1, generator=g) * (2.0 / 64) ** 0.5
    wsc = torch.randn(256, cin, 1, 1, generator=g) * (1.0 / cin) ** 0.5 if cin == 64 else None
    ss = lambda c: (0.5 + torch.rand(c, generator=g), torch.randn(c, generator=g) * 0.2)
    return w1, w2, w3, wsc, ss(64), ss(64), ss(256), (ss(256) if cin == 64 else None)


def _reference(x_nhwc, w1, w2, w3, wsc, ss1, ss2, ss3, sssc):
    """fp32 math on the bf16-rounded operands, bf16 rounding where the unfused path stores an activation"""
    x = x_nhwc.float().permute(0, 3, 1, 2)
    bn = lambda y, ss: y * ss[0].view(1, -1, 1, 1) + ss[1].view(1, -1, 1, 1)
    y1 = _bf(F.relu(bn(F.conv2d(x, _bf(w1)), ss1)))
    y2 = _bf(F.relu(bn(F.conv2d(y1, _bf(w2), padding=1), ss2)))
    sc = _bf(bn(F.conv2d(x, _bf(wsc)), sssc)) if wsc is not None else x
    out = _bf(F.relu(bn(F.conv2d(y2, _bf(w3)), ss3) + sc))
    return out.permute(0, 2, 3, 1).contiguous()


def _krsc(w):
    """the KRSC memory image of an OIHW weight, bf16 (what ParamBank.compute() hands the kernels)"""
    return w.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV[0])


@pytest.mark.parametrize("cin", [256, 64])
@pytest.mark.parametrize("n,h,w,maxwg", [(2, 16, 16, 0), (1, 12, 20, 0), (3, 16, 8, 2), (1, 9, 7, 0)])
def test_res2_block_equals_the_block_op_by_op(hw, monkeypatch, cin, n, h, w, maxwg):
    if maxwg:
        monkeypatch.setenv("CB_PERSISTENT_MAXWG", str(maxwg))  # persistent loop: each workgroup walks several tiles
    w1, w2, w3, wsc, ss1, ss2, ss3, sssc = _make(cin, 3 + cin)
    x = (torch.randn(n, h, w, cin, generator=_gen(7)) * 1.5).to(torch.bfloat16)
    if cin == 256:
        x = x.relu()                                     # (a block input is a ReLU output)
    ref = _reference(x, w1, w2, w3, wsc, ss1, ss2, ss3, sssc)
    d = lambda t: None if t is None else tuple(u.to(DEV[0]) for u in t)
    out = ops.res2_block(x.to(DEV[0]), _krsc(w1), _krsc(w2), _krsc(w3), d(ss1), d(ss2), d(ss3), wsc=None if wsc is None else _krsc(wsc), sssc=d(sssc))
    assert out.shape == (n, h, w, 256) and out.dtype == torch.bfloat16
    got = out.float().cpu()
    # bf16 outputs of O(1): one bf16 ulp of slack for the cases where the two fp32 sums round apart (summation order of the MFMA)
    err = (got - ref).abs()
    assert float(err.max()) <= 2 ** -7 * max(1.0, float(ref.abs().max())), float(err.max())
    assert float((err > 0).float().mean()) < 0.05, float((err > 0).float().mean())      # ... and almost every element bit-equal


@pytest.mark.parametrize("first", [True, False])
def test_fused_block_equals_the_unfused_product_path(hw, first):
    """the model's own modules: ops.res2_block against the three / four cb_gemm launches of modeling._conv_fwd on the same block"""
    from clipbert_amd import modeling as M
    from clipbert_amd import synthetic as S
    from oracle import clipbert_oracle as O
    cfg = dict(O.BASE_CONFIG, num_hidden_layers=1, num_labels=2, loss_type="ce", margin=0.1, vocab_size=300, max_position_embeddings=40)
    model = M.ClipBert(cfg, detectron2_model_cfg="R-50-grid.yaml", transformer_cls=M.ClipBertForVideoTextRetrieval)
    model.load_state_dict(S.full_state_dict(cfg, "retrieval", 5), strict=True)
    model.to(DEV[0]).eval()
    model.prepare(dtype=torch.bfloat16, device=DEV[0])
    rt = model.rt
    blk = model.cnn.feature.backbone.res2[0 if first else 1]
    cin = 64 if first else 256
    x = (torch.randn(2, 16, 24, cin, generator=_gen(11))).relu().to(torch.bfloat16).to(DEV[0])
    sc = M._conv_fwd(rt, x, blk.shortcut) if blk.shortcut is not None else x
    y1 = M._conv_fwd(rt, x, blk.conv1, act=M.ACT_RELU)
    y2 = M._conv_fwd(rt, y1, blk.conv2, act=M.ACT_RELU)
    ref = M._conv_fwd(rt, y2, blk.conv3, residual=sc, relu_after=True)
    got = M._res2_block_fused(rt, x, blk)
    err = (got.float() - ref.float()).abs()
    assert float(err.max()) <= 2 ** -7 * max(1.0, float(ref.float().abs().max())), float(err.max())
    assert float((err > 0).float().mean()) < 0.05


@pytest.mark.parametrize("n,h,w,maxwg", [(2, 32, 32, 0), (1, 48, 40, 3), (1, 30, 18, 0)])
def test_stem_pool_equals_conv_then_pool(hw, monkeypatch, n, h, w, maxwg):
    """cb_stem_pool against the two launches it replaces (cb_gemm in its stem form + cb_maxpool_fwd) on the same packed image, and
    against conv2d + max_pool2d in fp32 PyTorch with the convolution output rounded to bf16 where the unfused path stores it"""
    if maxwg:
        monkeypatch.setenv("CB_PERSISTENT_MAXWG", str(maxwg))
    from clipbert_amd import modeling as M
    g = _gen(21)
    frames = torch.randint(0, 256, (n, 3, h, w), generator=g, dtype=torch.uint8)
    wt = torch.randn(64, 3, 7, 7, generator=g) * (2.0 / 147) ** 0.5
    scale, shift = 0.5 + torch.rand(64, generator=g), torch.randn(64, generator=g) * 0.2
    mean, std = (123.675, 116.28, 103.53), (1.0, 1.0, 1.0)
    packed = ops.stem_pack(frames.to(DEV[0]), torch.bfloat16, 3, mean, std, extra_w=2)
    wp_ = torch.zeros(64, 7, 8, 4)
    wp_[:, :, :7, :3] = wt.permute(0, 2, 3, 1)
    wk = wp_.view(64, 224).to(torch.bfloat16).to(DEV[0])
    oh, ow = (h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1
    got = ops.stem_pool(packed, wk, scale.to(DEV[0]), shift.to(DEV[0]), oh, ow).float().cpu()
    # (a) the unfused launches
    hp, wpk = packed.shape[1], packed.shape[2]
    tab = ops.build_pixel_table(n, oh, ow, 2, 0, hp * wpk * 4, wpk * 4, 4, DEV[0])
    y = torch.empty(n, oh, ow, 64, dtype=torch.bfloat16, device=DEV[0])
    ops.gemm(packed, wk, n * oh * ow, 64, 224, out=y.view(-1, 64), a_mode=M.ROWK_GATHER, a_tab=tab, lda=0, ldb=224, R=7, S=1, Cin=32, H=hp, W=wpk,
             sH=wpk * 4, sW=4, scale=scale.to(DEV[0]), shift=shift.to(DEV[0]), act=M.ACT_RELU)
    ref_a = ops.maxpool_fwd(y, 3, 2, 1).float().cpu()
    assert got.shape == ref_a.shape
    assert torch.equal(got, ref_a), float((got - ref_a).abs().max())                  # same K order, same rounding points: bit-equal
    # (a') round 6: the uint8 frames straight into the kernel (cb_stem_pool_u8: ImageNorm + BGR flip + padding in the tile loader): the same bits,
    # with non-trivial pixel statistics too
    got8 = ops.stem_pool_u8(frames.to(DEV[0]), mean, std, wk, scale.to(DEV[0]), shift.to(DEV[0])).float().cpu()
    assert torch.equal(got8, got), float((got8 - got).abs().max())
    mean2, std2 = (120.0, 110.5, 100.25), (57.0, 58.5, 60.0)
    packed2 = ops.stem_pack(frames.to(DEV[0]), torch.bfloat16, 3, mean2, std2, extra_w=2)
    want2 = ops.stem_pool(packed2, wk, scale.to(DEV[0]), shift.to(DEV[0]), oh, ow)
    got2 = ops.stem_pool_u8(frames.to(DEV[0]), mean2, std2, wk, scale.to(DEV[0]), shift.to(DEV[0]))
    assert torch.equal(got2, want2)
    # (b) plain PyTorch: BGR, mean-subtracted, bf16 operands, fp32 math
    x = _bf(frames.float()[:, [2, 1, 0]] - torch.tensor(mean)[[2, 1, 0]].view(1, 3, 1, 1))
    conv = F.conv2d(x, _bf(wt), stride=2, padding=3) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    ref_b = F.max_pool2d(_bf(F.relu(conv)), 3, 2, 1).permute(0, 2, 3, 1)
    err = (got - ref_b).abs()
    assert float(err.max()) <= 2 ** -6 * max(1.0, float(ref_b.abs().max())), float(err.max())
