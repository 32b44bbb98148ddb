"""The CNN half of the oracle cannot be checked against detectron2 (absent, un-installable).  It is
cross-checked here against an INDEPENDENT implementation of the same published architecture:
HF transformers' ResNetModel configured as ResNet-50 v1 (stride in the first 1x1 of each stage =
detectron2's STRIDE_IN_1X1=True), BatchNorm in eval mode carrying the same statistics
(= FrozenBatchNorm2d, eps 1e-5)."""
import pytest
import torch

from clipbert_amd import synthetic as S
from oracle import clipbert_oracle as O


def _hf_resnet50():
    tr = pytest.importorskip("transformers")
    cfg = tr.ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[256, 512, 1024, 2048],
                          depths=[3, 4, 6, 3], layer_type="bottleneck", hidden_act="relu",
                          downsample_in_first_stage=False, downsample_in_bottleneck=True)
    return tr.ResNetModel(cfg).eval()


def _copy(sd, hf):
    def cb(dst_conv, dst_bn, p):
        dst_conv.weight.data.copy_(sd[p + ".weight"])
        dst_bn.weight.data.copy_(sd[p + ".norm.weight"])
        dst_bn.bias.data.copy_(sd[p + ".norm.bias"])
        dst_bn.running_mean.copy_(sd[p + ".norm.running_mean"])
        dst_bn.running_var.copy_(sd[p + ".norm.running_var"])
        assert abs(dst_bn.eps - O.FROZEN_BN_EPS) < 1e-12
    bb = "cnn.feature.backbone."
    emb = hf.embedder.embedder
    cb(emb.convolution, emb.normalization, bb + "stem.conv1")
    for si, (name, n_blocks, _m, _o, _s) in enumerate(O.RESNET50_STAGES):
        stage = hf.encoder.stages[si]
        for b in range(n_blocks):
            blk = stage.layers[b]
            p = f"{bb}{name}.{b}"
            if (p + ".shortcut.weight") in sd:
                cb(blk.shortcut.convolution, blk.shortcut.normalization, p + ".shortcut")
            for ci in range(3):
                cl = blk.layer[ci]
                cb(cl.convolution, cl.normalization, f"{p}.conv{ci + 1}")


def test_resnet50_matches_independent_implementation():
    sd = S.cnn_state_dict(11)
    hf = _hf_resnet50()
    _copy(sd, hf)
    x = torch.randn(2, 3, 96, 128, generator=S._gen(11, "x")) * 50
    with torch.no_grad():
        ours = O.resnet50_res5(sd, x, "cnn.feature.backbone.")
        theirs = hf(x).last_hidden_state
    assert ours.shape == (2, 2048, 3, 4)
    torch.testing.assert_close(ours, theirs, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("size,res5", [(224, 7), (448, 14), (768, 24)])
def test_resnet50_matches_independent_implementation_at_the_shipped_sizes(size, res5):
    """the three input sizes of the shipped configs (src/configs: 224 pretraining / bench shape, 448 MSRVTT, 768
    TGIF-QA) -> res5 maps of 7 / 14 / 24: padding and stride arithmetic of every stage against the independent implementation."""
    torch.set_num_threads(max(1, min(32, __import__("os").cpu_count() or 1)))
    sd = S.cnn_state_dict(11)
    hf = _hf_resnet50()
    _copy(sd, hf)
    x = torch.randn(1, 3, size, size, generator=S._gen(11, f"x{size}")) * 50
    with torch.no_grad():
        ours = O.resnet50_res5(sd, x, "cnn.feature.backbone.")
        theirs = hf(x).last_hidden_state
        grid = O.grid_feat_backbone(sd, x[None], "cnn.")
    assert ours.shape == (1, 2048, res5, res5)
    assert grid.shape == (1, 1, res5 // 2, res5 // 2, 768)             # MaxPool2d(2, 2) floors: 7 -> 3, 14 -> 7, 24 -> 12
    torch.testing.assert_close(ours, theirs, rtol=1e-4, atol=1e-3)


def test_resnet50_gradients_match_independent_implementation():
    """autograd through both implementations: d(sum of res5 * fixed random tensor) / d(every conv weight of res3..res5 and the input)"""
    sd = {k: v.clone() for k, v in S.cnn_state_dict(11).items()}
    hf = _hf_resnet50()
    _copy(sd, hf)
    x = (torch.randn(2, 3, 96, 96, generator=S._gen(11, "xg")) * 50).requires_grad_(True)
    x2 = x.detach().clone().requires_grad_(True)
    for k, v in sd.items():
        if k.endswith(".weight") and ".norm." not in k:
            v.requires_grad_(True)
    r = torch.randn(2, 2048, 3, 3, generator=S._gen(11, "r"))
    (O.resnet50_res5(sd, x, "cnn.feature.backbone.") * r).sum().backward()
    (hf(x2).last_hidden_state * r).sum().backward()
    # (relative L2: single elements differ where the two fp32 evaluations put a max-pool / ReLU decision on different sides of a tie)
    assert float((x.grad - x2.grad).norm() / x2.grad.norm()) < 5e-3
    bb = "cnn.feature.backbone."
    checked = 0
    for si, (name, n_blocks, _m, _o, _s) in enumerate(O.RESNET50_STAGES):
        for b in range(n_blocks):
            blk = hf.encoder.stages[si].layers[b]
            pairs = [(f"{bb}{name}.{b}.conv{ci + 1}.weight", blk.layer[ci].convolution.weight) for ci in range(3)]
            if f"{bb}{name}.{b}.shortcut.weight" in sd:
                pairs.append((f"{bb}{name}.{b}.shortcut.weight", blk.shortcut.convolution.weight))
            for k, w in pairs:
                g, h = sd[k].grad, w.grad
                assert float((g - h).norm() / h.norm()) < 5e-3, k          # fp32 noise through up to 50 layers; a wrong stride or padding gives O(1)
                checked += 1
    assert checked == 52                                                 # 16 blocks x 3 convolutions + 4 projection shortcuts


def test_detectron2_facts_the_restatement_rests_on():
    """The [3P] facts of SURVEY.md a3, stated once as executable checks on the oracle (each cites the detectron2 @ ffff8ac file it
    restates; the source itself is absent from the image, so these pin the RESTATEMENT, not detectron2)."""
    assert O.DETECTRON2_FACTS["frozen_bn_eps"] == O.FROZEN_BN_EPS == 1e-5
    assert O.RESNET50_STAGES == (("res2", 3, 64, 256, 1), ("res3", 4, 128, 512, 2), ("res4", 6, 256, 1024, 2), ("res5", 3, 512, 2048, 2))
    sd = S.cnn_state_dict(11)
    bb = "cnn.feature.backbone."
    # the stem: 7x7 stride 2 pad 3 + max-pool 3x3 stride 2 pad 1 -> 1/4 resolution
    taps = {}
    x = torch.randn(1, 3, 64, 64, generator=S._gen(11, "f")) * 50
    with torch.no_grad():
        O.resnet50_res5(sd, x, bb, taps)
    assert taps["stem"].shape == (1, 64, 16, 16) and taps["res2"].shape == (1, 256, 16, 16)
    assert taps["res3"].shape == (1, 512, 8, 8) and taps["res4"].shape == (1, 1024, 4, 4) and taps["res5"].shape == (1, 2048, 2, 2)
    # STRIDE_IN_1X1 = True: the stride sits in conv1 (1x1) of a stage's first block, conv2 (3x3) has stride 1 -> moving conv1's
    # weights of res3.0 changes only what the strided sampling sees: check against an explicit strided slice
    with torch.no_grad():
        y_in = taps["res2"]
        w = sd[bb + "res3.0.conv1.weight"]
        direct = torch.nn.functional.conv2d(y_in[:, :, ::2, ::2], w)      # a 1x1 stride-2 conv = 1x1 conv on every second pixel
        strided = torch.nn.functional.conv2d(y_in, w, stride=2)
    assert torch.equal(direct, strided)
    # a projection shortcut exists exactly on the first block of each stage (channel change), nowhere else
    for name, n_blocks, *_ in O.RESNET50_STAGES:
        for b in range(n_blocks):
            assert ((f"{bb}{name}.{b}.shortcut.weight" in sd) == (b == 0)), (name, b)
    # the grid encoder has no bias and pools BEFORE the ReLU (src/modeling/grid_feat.py:43-48): equal to ReLU-then-pool only because
    # max and ReLU commute -- both orders are checked
    with torch.no_grad():
        g = torch.nn.functional.conv2d(taps["res5"], sd["cnn.grid_encoder.0.weight"], None, 1, 1)
        a = torch.relu(torch.nn.functional.max_pool2d(g, 2, 2))
        b_ = torch.nn.functional.max_pool2d(torch.relu(g), 2, 2)
    assert torch.equal(a, b_) and "cnn.grid_encoder.0.bias" not in sd


def test_grid_backbone_shapes_and_bgr_flip():
    sd = S.cnn_state_dict(11)
    v = torch.randn(1, 2, 3, 64, 64, generator=S._gen(11, "v")) * 50
    with torch.no_grad():
        g = O.grid_feat_backbone(sd, v)
        g_flip = O.grid_feat_backbone(sd, v[:, :, [2, 1, 0]])
    assert g.shape == (1, 2, 1, 1, 768)          # 64 -> res5 2x2 -> pool 1x1
    assert (g >= 0).all()
    assert not torch.allclose(g, g_flip)         # channel order matters (grid_feat.py:92-94)
