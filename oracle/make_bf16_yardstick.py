"""bf16 yardstick: what does bf16 STORAGE alone cost on the golden cases?  TEST INFRASTRUCTURE ONLY.

Run anywhere (CPU, needs only the committed goldens):   python -m oracle.make_bf16_yardstick [case ...]

For every golden case the ORACLE is run end to end in its two bf16 modes (oracle/clipbert_oracle.py: ``precision("bf16")`` = every op's
output rounded, ``precision("bf16_fused")`` = rounded only where a fused implementation must store; fp32 accumulation and statistics in
both) and the SAME error figures that tests/test_parity_record.py measures on the product are measured on it, against the same fp32
goldens produced by the reference's classes.  For the retrieval training case all parameter gradients of both modes are compared with
the oracle's fp32 autograd (per-tensor relative L2 error, cosine of the flat gradient).

Output: tests/golden/bf16_yardstick.json -- a COMMITTED CONSTANT.  The product's bf16 parity bounds are derived from it
(tests/test_parity_record.py: product error <= 1.5 x yardstick) and from nothing measured on the product.  Regenerate only when a golden
changes; say why in the commit.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import clipbert_oracle as O          # noqa: E402
from oracle import make_golden as G              # noqa: E402

OUT = os.path.join(G.GOLDEN_DIR, "bf16_yardstick.json")
MODES = ("bf16", "bf16_fused")


def top2_margin(x: np.ndarray) -> np.ndarray:
    t = np.sort(x, axis=-1)[..., -2:]
    return t[..., 1] - t[..., 0]


def forward_errors(name: str, mode: str) -> dict:
    """the figures of tests/test_parity_record.py::test_bf16_error_against_reference_goldens..., measured on the oracle in ``mode``"""
    gold = np.load(os.path.join(G.GOLDEN_DIR, name + ".npz"))
    cfg, head, sd, batch = G.build_case(name)
    rec = {}
    with torch.no_grad(), O.precision(mode):
        b = dict(batch)
        b["visual_inputs"] = O._r(b["visual_inputs"])         # the stem's input image is stored in bf16
        if name in G.CASES:
            out = O.clipbert_forward(sd, b, cfg, head)
            if head == "pretraining":
                mlm = out["mlm_scores"].numpy()
                rec["itm_scores"] = float(np.abs(out["itm_scores"].numpy() - gold["itm_scores"]).max())
                rec["mlm_scores_strided"] = float(np.abs(mlm[..., ::509] - gold["mlm_scores_strided"]).max())
                rec["mlm_argmax_agreement"] = float((mlm.argmax(-1) == gold["mlm_argmax"]).mean())
                if "mlm_margin_min" in gold.files:
                    rec["golden_mlm_margin_min"] = float(gold["mlm_margin_min"])
            else:
                lg = out["logits"].numpy()
                rec["logits"] = float(np.abs(lg - gold["logits"]).max())
                rec["logit_scale"] = float(np.abs(gold["logits"]).max())
                if head == "multiple_choice":
                    rec["answer_ids_agree"] = float((lg.argmax(-1) == gold["logits"].argmax(-1)).mean())
                    rec["golden_margin_min"] = float(top2_margin(gold["logits"]).min())
        else:
            c = G.CLIP_CASES[name]
            vis = b["visual_inputs"].view(c["n_videos"], c["n_clips"], c["n_frames"], *b["visual_inputs"].shape[2:])
            per_clip = []
            for k in range(c["n_clips"]):
                bb = dict(visual_inputs=vis[:, k], text_input_ids=batch["text_input_ids"], text_input_mask=batch["text_input_mask"],
                          n_examples_list=list(batch["n_examples_list"]))
                per_clip.append(O.clipbert_forward(sd, bb, cfg, head)["logits"])
            st = torch.stack(per_clip)
            if c["mode"] == "train":
                rec["logits"] = float(np.abs(st.numpy() - gold["stack"]).max())
                rec["logit_scale"] = float(np.abs(gold["stack"]).max())
                pooled = O.aggregate_clip_logits(per_clip, c["pool"])
                if c["pool"] == "lse":
                    loss = O.lse_train_loss(pooled, batch["labels"])
                else:
                    loss = torch.nn.functional.cross_entropy(pooled.view(-1, cfg["num_labels"]), batch["labels"].view(-1), reduction="none")
                rec["loss"] = abs(float(loss.mean()) - float(gold["loss"].mean()))
                if head == "multiple_choice":
                    gm = gold["stack"].mean(0).reshape(-1, cfg["num_labels"])
                    om = st.numpy().mean(0).reshape(-1, cfg["num_labels"])
                    rec["answer_ids_agree"] = float((om.argmax(-1) == gm.argmax(-1)).mean())
                    rec["golden_margin_min"] = float(top2_margin(gm).min())
            else:
                scores = O.lse_inference_scores(O.aggregate_clip_logits(per_clip, c["pool"])).tolist()
                rec["scores"] = float(max(abs(a - r) for a, r in zip(scores, gold["scores"].tolist())))
    return rec


def gradient_errors(name: str = "retrieval_ce") -> dict:
    """all parameter gradients of the training forward + backward: each bf16 mode against the oracle's own fp32 autograd"""
    cfg, head, sd, batch = G.build_case(name)

    def grads(mode):
        sdr = {k: v.clone().requires_grad_(v.is_floating_point() and ".norm." not in k) for k, v in sd.items()}
        with O.precision(mode):
            b = dict(batch)
            b["visual_inputs"] = O._r(b["visual_inputs"])
            O.clipbert_forward(sdr, b, cfg, head)["loss"].mean().backward()
        return {k: v.grad for k, v in sdr.items() if v.requires_grad and v.grad is not None}

    def trained(k):          # detectron2 freezes the stem and res2 (FREEZE_AT = 2, src/modeling/grid_feat.py:59-66): no gradient exists there
        return "stem" not in k and "res2" not in k

    ref = grads("fp32")
    out = {}
    for mode in MODES:
        g = grads(mode)
        dot = n1 = n2 = 0.0
        per = {}
        for k, r in ref.items():
            a = g[k].double()
            r = r.double()
            if trained(k):
                dot += float((a * r).sum()); n1 += float((a * a).sum()); n2 += float((r * r).sum())
            if float(r.norm()) > 1e-8:
                per[k] = float((a - r).norm() / r.norm())
        vals = np.array([v for k, v in per.items() if trained(k)])
        worst = max((k for k in per if trained(k)), key=per.get)
        out[mode] = {"flat_gradient_cosine": dot / (n1 ** 0.5 * n2 ** 0.5), "median_tensor_rel_l2": float(np.median(vals)),
                     "p90_tensor_rel_l2": float(np.quantile(vals, 0.9)), "worst_tensor_rel_l2": per[worst], "worst_tensor": worst,
                     "tensors": int(len(vals)), "scope": "parameters outside the frozen stem / res2 (per_tensor_rel_l2 lists every tensor)",
                     "per_tensor_rel_l2": {k: round(v, 5) for k, v in per.items()}}
    return out


def main():
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    only = set(sys.argv[1:])
    data = {}
    if os.path.exists(OUT):
        with open(OUT) as fh:
            data = json.load(fh)
    data["_comment"] = ("bf16-storage error of the CPU oracle (two rounding granularities) against the fp32 goldens; generated by "
                        "oracle/make_bf16_yardstick.py; the product's bf16 bounds are multiples of these (tests/test_parity_record.py)")
    for name in list(G.CASES) + list(G.CLIP_CASES):
        if only and name not in only:
            continue
        data[name] = {mode: forward_errors(name, mode) for mode in MODES}
        print(name, json.dumps(data[name]), flush=True)
    if not only or "grad" in only:
        data["grad_retrieval_ce_vs_oracle_autograd"] = gradient_errors()
        print({m: {k: v for k, v in d.items() if k != "per_tensor_rel_l2"} for m, d in data["grad_retrieval_ce_vs_oracle_autograd"].items()})
    with open(OUT, "w") as fh:
        json.dump(data, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
