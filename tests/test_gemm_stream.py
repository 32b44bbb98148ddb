"""cb_gemm tile 8, the streaming structure (clipbert_amd/csrc/gemm_stream_impl.h): persistent workgroups, weights resident in LDS, the
next A tile by LDS-DMA and the epilogue's operands in flight while a tile is stored.  Every instantiation must give, BIT FOR BIT, what
the one-workgroup-per-tile 64x64 kernel gives (same MFMA, same K order, same epilogue code) and agree with a plain PyTorch fp32
reference within the bf16 tolerance.  Runs on the host lane-level emulator (CPU suite; EMUL_DMA_LAZY models the asynchronous
LDS-DMA: a wrong wait count fails deterministically) and, marked `gpu`, on the MI355X."""
import os

import pytest
import torch

from clipbert_amd import ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _plan(d):
    from clipbert_amd import _lib
    import ctypes as C
    out = (C.c_int32 * 4)()
    rc = _lib.get().cb_gemm_plan(C.byref(d), 1, out)
    assert rc == 0
    return list(out)


# (M, N, K, residual, relu): the forward shapes of res2 / res3 at small pixel counts, ragged M included
FWD = [
    (200, 256, 64, True, True),      # variant 0: conv3 of res2 (+ residual, ReLU after)
    (333, 256, 64, False, False),    # variant 0: projection shortcut of res2.0 (FrozenBN only), M not a multiple of the tile
    (130, 512, 128, True, True),     # variant 1: conv3 of res3, N in four column blocks
    (257, 128, 128, False, True),    # variant 1, one column block
    (1000, 256, 48, True, False),    # K tail inside the single K tile
    (300, 384, 104, False, False),   # variant 1 with a K tail in the second K tile, three column blocks
    (700, 512, 64, True, True),      # variant 0, two column blocks
]


@pytest.mark.parametrize("case", FWD, ids=lambda c: "x".join(map(str, c[:3])) + ("+res" if c[3] else ""))
def test_stream_forward_equals_tile_kernel_bitwise(hw, case, monkeypatch):
    M, N, K, res, relu = case
    if hw.name == "emul":
        monkeypatch.setenv("CB_PERSISTENT_MAXWG", "2")         # a couple of workgroups walk many tiles: the persistent loop is exercised
    a, w = hw(rnd(M, K, seed=1).bfloat16()), hw(rnd(N, K, seed=2, scale=0.2).bfloat16())
    scale, shift = hw(rnd(N, seed=3) * 0.1 + 1.0), hw(rnd(N, seed=4) * 0.1)
    r = hw(rnd(M, N, seed=5).bfloat16()) if res else None
    kw = dict(scale=scale, shift=shift, act=ops.ACT_RELU if (relu and not res) else ops.ACT_NONE, residual=r, relu_after=bool(relu and res))
    o_ref, o_str = hw(torch.full((M, N), float("nan")).bfloat16()), hw(torch.full((M, N), float("nan")).bfloat16())
    ops.gemm(a, w, M, N, K, out=o_ref, tile=2, **kw)
    d = ops.gemm_desc(a, w, M, N, K, out=o_str, tile=8, **kw)
    assert _plan(d)[0] == 8
    ops.gemm(a, w, M, N, K, out=o_str, tile=8, **kw)
    assert torch.equal(o_ref.cpu(), o_str.cpu())
    want = (a.float().cpu() @ w.float().cpu().t()) * scale.cpu() + shift.cpu()
    if kw["act"]:
        want = want.relu()
    if res:
        want = want + r.float().cpu()
        if relu:
            want = want.relu()
    torch.testing.assert_close(o_str.float().cpu(), want, rtol=2e-2, atol=2e-2)


def test_stream_is_chosen_for_the_hbm_bound_shapes_and_refused_elsewhere(hw):
    """auto (tile 0): the streaming structure takes a short reduction over many rows; a request for it on a problem it does not cover
    is an error, never a silent fallback"""
    M = 40000 if hw.name == "gpu" else 64
    a, w = hw(rnd(max(M, 64), 64, seed=1).bfloat16()), hw(rnd(256, 64, seed=2).bfloat16())
    o = hw(torch.empty(max(M, 64), 256).bfloat16())
    if hw.name == "gpu":
        assert _plan(ops.gemm_desc(a, w, M, 256, 64, out=o))[0] == 8
    small = ops.gemm_desc(a[:64], w, 64, 256, 64, out=o[:64])
    assert _plan(small)[0] != 8                                 # few rows: one workgroup per tile
    bad = ops.gemm_desc(a[:64], hw(rnd(256, 512, seed=3).bfloat16()), 64, 256, 512, out=o[:64], tile=8)          # K = 512: not covered
    with pytest.raises(RuntimeError):
        ops.gemm_group([bad], o)
