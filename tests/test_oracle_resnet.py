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


def test_grid_backbone_shapes_and_bgr_flip():
    sd = S.cnn_state_dict(11)
    v = torch.randn(1, 2, 3, 64, 64, generator=S._gen(11, "v")) * 50
    with torch.no_grad():
        g = O.grid_feat_backbone(sd, v)
        g_flip = O.grid_feat_backbone(sd, v[:, :, [2, 1, 0]])
    assert g.shape == (1, 2, 1, 1, 768)          # 64 -> res5 2x2 -> pool 1x1
    assert (g >= 0).all()
    assert not torch.allclose(g, g_flip)         # channel order matters (grid_feat.py:92-94)
