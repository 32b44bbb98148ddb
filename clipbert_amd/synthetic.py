"""Deterministic synthetic weights and inputs (no datasets / checkpoints are reachable here).

Weights follow SURVEY.md section 8(c): Linear/Embedding ~ N(0, 0.02) (transformers.py:559-570 of the
reference), conv weights MSRA-normal, FrozenBN statistics random-but-fixed so that the affine is
non-trivial.  Unlike the reference's init, biases and LayerNorm affines are ALSO randomised
(small) so that a kernel that drops a bias or a gamma cannot pass a parity test.

Inputs follow SURVEY.md section 8(d): uint8 frames ~ U{0..255}; token ids ~ U{1000..30521} with
[CLS]=101 first, [SEP]=102 at a random length, pad 0 afterwards, mask = (pos <= sep).

Every tensor is drawn from its own ``torch.Generator`` seeded by (seed, key) so the values do not
depend on construction order and are identical on every host with this torch build.
"""
import zlib
from typing import Dict, Sequence

import torch

RESNET50_STAGES = (("res2", 3, 64, 256), ("res3", 4, 128, 512),
                   ("res4", 6, 256, 1024), ("res5", 3, 512, 2048))


def _gen(seed: int, key: str) -> torch.Generator:
    g = torch.Generator()
    g.manual_seed((seed * 1000003 + zlib.crc32(key.encode())) % (2 ** 63 - 1))
    return g


def _normal(seed, key, shape, std, mean=0.0):
    return torch.randn(shape, generator=_gen(seed, key), dtype=torch.float32) * std + mean


def _uniform(seed, key, shape, lo, hi):
    return torch.rand(shape, generator=_gen(seed, key), dtype=torch.float32) * (hi - lo) + lo


def cnn_state_dict(seed: int = 42, prefix: str = "cnn.", hidden: int = 768,
                   backbone_out: int = 2048) -> Dict[str, torch.Tensor]:
    """Keys as detectron2 names them inside ClipBert."""
    sd = {}

    def conv(name, cout, cin, k, gain=1.0):
        fan_out = cout * k * k  # detectron2 uses c2_msra_fill: kaiming_normal_(mode="fan_out")
        sd[name + ".weight"] = _normal(seed, name + ".weight", (cout, cin, k, k),
                                       gain * (2.0 / fan_out) ** 0.5)
        sd[name + ".norm.weight"] = _uniform(seed, name + ".norm.weight", (cout,), 0.5, 1.5)
        sd[name + ".norm.bias"] = _normal(seed, name + ".norm.bias", (cout,), 0.1)
        sd[name + ".norm.running_mean"] = _normal(seed, name + ".norm.running_mean", (cout,), 0.1)
        sd[name + ".norm.running_var"] = _uniform(seed, name + ".norm.running_var", (cout,), 0.5, 1.5)

    bb = prefix + "feature.backbone."
    conv(bb + "stem.conv1", 64, 3, 7)
    cin = 64
    for name, n_blocks, mid, cout in RESNET50_STAGES:
        for b in range(n_blocks):
            p = f"{bb}{name}.{b}"
            if cin != cout:
                conv(p + ".shortcut", cout, cin, 1)
            conv(p + ".conv1", mid, cin, 1)
            conv(p + ".conv2", mid, mid, 3)
            conv(p + ".conv3", cout, mid, 1, gain=0.5)   # keep the residual sum bounded
            cin = cout
    k = prefix + "grid_encoder.0.weight"
    # scaled so the grid features are O(0.5): the row/col/type embeddings (std 0.02) then matter
    # to the LayerNorm output at the 1e-2 level and a kernel that drops them fails parity.
    sd[k] = _normal(seed, k, (hidden, backbone_out, 3, 3), (2.0 / (backbone_out * 9)) ** 0.5 / 300.0)
    return sd


def transformer_state_dict(cfg: dict, head: str = "retrieval", seed: int = 42,
                           prefix: str = "transformer.") -> Dict[str, torch.Tensor]:
    """head in {retrieval, multiple_choice, sequence_classification, pretraining}."""
    d, ff, v = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"]
    std = cfg.get("initializer_range", 0.02)
    sd = {}

    def lin(name, out, inp):
        sd[name + ".weight"] = _normal(seed, name + ".weight", (out, inp), std)
        sd[name + ".bias"] = _normal(seed, name + ".bias", (out,), std)

    def emb(name, n, dim):
        sd[name + ".weight"] = _normal(seed, name + ".weight", (n, dim), std)

    def ln(name, dim):
        sd[name + ".weight"] = _normal(seed, name + ".weight", (dim,), 0.05, 1.0)
        sd[name + ".bias"] = _normal(seed, name + ".bias", (dim,), 0.05)

    b = prefix + "bert."
    emb(b + "embeddings.word_embeddings", v, d)
    sd[b + "embeddings.word_embeddings.weight"][cfg.get("pad_token_id", 0)] = 0.0
    emb(b + "embeddings.position_embeddings", cfg["max_position_embeddings"], d)
    emb(b + "embeddings.token_type_embeddings", cfg["type_vocab_size"], d)
    ln(b + "embeddings.LayerNorm", d)
    emb(b + "visual_embeddings.position_embeddings", cfg["max_position_embeddings"], d)  # unused
    emb(b + "visual_embeddings.row_position_embeddings", cfg["max_grid_row_position_embeddings"], d)
    emb(b + "visual_embeddings.col_position_embeddings", cfg["max_grid_col_position_embeddings"], d)
    emb(b + "visual_embeddings.token_type_embeddings", 1, d)
    ln(b + "visual_embeddings.LayerNorm", d)
    for i in range(cfg["num_hidden_layers"]):
        p = f"{b}encoder.layer.{i}."
        lin(p + "attention.self.query", d, d)
        lin(p + "attention.self.key", d, d)
        lin(p + "attention.self.value", d, d)
        lin(p + "attention.output.dense", d, d)
        ln(p + "attention.output.LayerNorm", d)
        lin(p + "intermediate.dense", ff, d)
        lin(p + "output.dense", d, ff)
        ln(p + "output.LayerNorm", d)
    lin(b + "pooler.dense", d, d)
    if head == "pretraining":
        c = prefix + "cls."
        lin(c + "predictions.transform.dense", d, d)
        ln(c + "predictions.transform.LayerNorm", d)
        sd[c + "predictions.bias"] = _normal(seed, c + "predictions.bias", (v,), std)
        # tied / aliased keys exactly as the reference saves them (transformers.py:507-510)
        sd[c + "predictions.decoder.weight"] = sd[b + "embeddings.word_embeddings.weight"]
        sd[c + "predictions.decoder.bias"] = sd[c + "predictions.bias"]
        lin(c + "seq_relationship", 2, d)
    elif head == "regression":
        r = prefix + "regressor."
        lin(r + "0", d, d)
        sd[r + "2.weight"] = _uniform(seed, r + "2.weight", (d,), 0.5, 1.5)
        sd[r + "2.bias"] = _normal(seed, r + "2.bias", (d,), 0.1)
        sd[r + "2.running_mean"] = _normal(seed, r + "2.running_mean", (d,), 0.02)
        sd[r + "2.running_var"] = _uniform(seed, r + "2.running_var", (d,), 0.001, 0.003)     # ELU(N(0, 0.02)-ish) has var ~1e-3
        sd[r + "2.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
        lin(r + "4", 1, d)
    else:
        n_out = 1 if head == "multiple_choice" else cfg["num_labels"]
        lin(prefix + "classifier.0", 2 * d, d)
        lin(prefix + "classifier.2", n_out, 2 * d)
    return sd


def full_state_dict(cfg: dict, head: str = "retrieval", seed: int = 42):
    sd = cnn_state_dict(seed, "cnn.", cfg["hidden_size"], cfg["backbone_channel_in_size"])
    sd.update(transformer_state_dict(cfg, head, seed, "transformer."))
    return sd


def synthetic_frames(n_videos: int, n_frames: int, size: int, seed: int = 42) -> torch.Tensor:
    """uint8 (Bv, n_frames, 3, size, size) RGB."""
    return torch.randint(0, 256, (n_videos, n_frames, 3, size, size), generator=_gen(seed, "frames"),
                         dtype=torch.uint8)


def synthetic_text(n_pairs: int, max_len: int, seed: int = 42, vocab: int = 30522):
    """ids (n, L) int64 and mask (n, L) int64."""
    g = _gen(seed, "text")
    ids = torch.randint(min(1000, vocab // 2), vocab, (n_pairs, max_len), generator=g, dtype=torch.long)
    lo = min(8, max_len)
    sep = torch.randint(lo - 1, max_len, (n_pairs,), generator=g, dtype=torch.long)
    pos = torch.arange(max_len).unsqueeze(0)
    ids[:, 0] = min(101, vocab - 2)
    ids[pos == sep.unsqueeze(1)] = min(102, vocab - 1)
    mask = (pos <= sep.unsqueeze(1)).long()
    return ids * mask, mask


def synthetic_labels(n: int, n_classes: int, seed: int = 42) -> torch.Tensor:
    return torch.randint(0, n_classes, (n,), generator=_gen(seed, "labels"), dtype=torch.long)


PIXEL_MEAN: Sequence[float] = (123.675, 116.28, 103.53)   # src/configs/*.json img_pixel_mean
PIXEL_STD: Sequence[float] = (1.0, 1.0, 1.0)
