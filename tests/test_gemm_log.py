"""clipbert_amd/gemm_log.py: the problems of a step are filed under the roof that bounds them (bench.py's `roofline.families`), with the
algorithmic flops / bytes the roofline prices them by.  Descriptors only -- nothing is launched (emulator backend for the pointers)."""
import torch

from clipbert_amd import gemm_log, ops


def _bf(*shape):
    return torch.zeros(*shape, dtype=torch.bfloat16)


def test_problems_are_filed_under_the_roof_that_bounds_them(emul):
    frames, px56, px28 = 64, 64 * 56 * 56, 64 * 28 * 28
    scale = torch.ones(512)
    # encoder linear: token rows x 3072 x 768, bias + GELU with the pre-activation as second output
    a, w, o, o2 = _bf(2624, 768), _bf(3072, 768), _bf(2624, 3072), _bf(2624, 3072)
    p = gemm_log._problem(ops.gemm_desc(a, w, 2624, 3072, 768, out=o, shift=torch.zeros(3072), act=ops.ACT_GELU, out2=o2))
    assert p["family"] == "encoder linear (fwd + dgrad)" and p["form"] == "fwd"
    assert p["flop"] == 2.0 * 2624 * 3072 * 768
    assert p["bytes"] == (2624 * 768 + 3072 * 768) * 2 + 2624 * 3072 * 2 + 2624 * 3072 * 2          # A + B + C + the second output
    # res2 conv3: 1x1, K = 64, FrozenBN + residual -> HBM-bound family
    x, w3, y, r = _bf(px56, 64), _bf(256, 64), _bf(px56, 256), _bf(px56, 256)
    p = gemm_log._problem(ops.gemm_desc(x, w3, px56, 256, 64, out=y, scale=scale[:256], shift=scale[:256], residual=r, relu_after=True))
    assert p["family"] == "resnet 1x1 conv, K <= 256 (fwd + dgrad)" and gemm_log.FAMILY_BOUND[p["family"]] == "hbm"
    assert p["bytes"] == (px56 * 64 + 256 * 64) * 2 + 2 * px56 * 256 * 2
    # res3 conv1 of blocks 1..3: K = 512 -> MFMA-priced 1x1 family
    p = gemm_log._problem(ops.gemm_desc(_bf(px28, 512), _bf(128, 512), px28, 128, 512, out=_bf(px28, 128), scale=scale[:128], shift=scale[:128], act=ops.ACT_RELU))
    assert p["family"] == "resnet 1x1 conv, K > 256 (fwd + dgrad)"
    # 3x3 convolution through the pixel table (the gathered input counted once, not once per tap)
    img = _bf(frames, 28, 28, 128)
    tab = ops.build_pixel_table(frames, 28, 28, 1, 1, 28 * 28 * 128, 28 * 128, 128, img.device)
    d = ops.gemm_desc(img, _bf(128, 9 * 128), px28, 128, 9 * 128, out=_bf(px28, 128), a_mode=ops.ROWK_GATHER, a_tab=tab, lda=0, ldb=9 * 128, R=3, S=3, Cin=128,
                      H=28, W=28, sH=28 * 128, sW=128, scale=scale[:128], shift=scale[:128], act=ops.ACT_RELU)
    p = gemm_log._problem(d)
    assert p["family"] == "conv 3x3 / 7x7 (fwd + dgrad)" and p["taps"] == 9
    assert p["bytes"] == (px28 * 128 + 128 * 9 * 128) * 2 + px28 * 128 * 2
    # weight gradient of a ResNet 1x1 convolution (K = pixels) and of an encoder linear (K = token rows): one family
    g, xx, gw = _bf(px28, 512), _bf(px28, 128), torch.zeros(512, 128)
    p = gemm_log._problem(ops.gemm_desc(g, xx, 512, 128, px28, out=gw, a_mode=ops.KROW, lda=512, b_mode=ops.KROW, ldb=128, accumulate=True))
    assert p["family"] == "weight gradients (linear + conv)" and p["form"] == "wgrad"
    assert p["bytes"] == (px28 * 512 + px28 * 128) * 2 + 2 * 512 * 128 * 4                               # fp32 gradient read + written
    # the data gradient of a linear stays with the encoder family
    p = gemm_log._problem(ops.gemm_desc(_bf(2624, 3072), _bf(3072, 768), 2624, 768, 3072, out=_bf(2624, 768), b_mode=ops.KROW, ldb=768))
    assert p["family"] == "encoder linear (fwd + dgrad)" and p["form"] == "dgrad"


def test_the_log_records_single_and_grouped_calls(emul):
    a, w, o = _bf(64, 64), _bf(64, 64), _bf(64, 64)
    with gemm_log.GemmLog() as log:
        ops.gemm(a, w, 64, 64, 64, out=o)
        ops.gemm_group([ops.gemm_desc(a, w, 64, 64, 64, out=_bf(64, 64)), ops.gemm_desc(a, w, 64, 64, 64, out=_bf(64, 64))], a)
    assert [len(ln["problems"]) for ln in log.launches] == [1, 2]
    assert ops.gemm.__module__ == "clipbert_amd.ops"                 # the wrappers are gone again
    fams = log.by_family()
    assert sum(len(v) for v in fams.values()) == 2
    log.launches[0]["replay"](); log.launches[1]["replay"]()         # the recorded calls can be issued again
