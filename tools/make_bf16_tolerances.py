#!/usr/bin/env python
"""tests/golden/bf16_tolerances.json from a parity record written by tests/test_parity_record.py on an MI355X
(gpurun_out/r03_bf16_parity.json, committed as profiles/r03_bf16_parity.json): every bound = 2 x the measured value.

  python tools/make_bf16_tolerances.py [record.json]

Error-like numbers (max |delta| of logits / scores / loss, relative L2 of gradients): 2 x measured, rounded UP to 3 significant digits.
Agreement-like numbers (argmax agreement, gradient cosine): the shortfall from 1 may double (1 - 2 (1 - measured))."""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SKIP = {"logit_scale", "answer_ids_agree_raw", "answer_ids_decided", "tensors", "worst_tensor"}


def up3(v):
    if v <= 0:
        return 0.0
    e = math.floor(math.log10(v)) - 2
    return round(math.ceil(v / 10 ** e - 1e-9) * 10 ** e, 12)


def main():
    rec_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r03_bf16_parity.json")
    rec = json.load(open(rec_path))
    out = {"_comment": "2x the bf16 max |delta| measured on an MI355X against the reference goldens (profiles/r03_bf16_parity.json); "
                       "see tests/test_parity_record.py, tools/make_bf16_tolerances.py"}
    for case, vals in sorted(rec.items()):
        tol = {}
        for k, v in vals.items():
            if k in SKIP or not isinstance(v, (int, float)):
                continue
            if k == "flat_gradient_cosine":
                tol["flat_gradient_cosine_min"] = math.floor((1 - 2 * (1 - v)) * 1e4) / 1e4
            elif k.endswith("agreement") or k.endswith("agree"):
                tol[k] = math.floor((1 - 2 * (1 - v)) * 1e3) / 1e3
            else:
                tol[k] = up3(2 * v)
        out[case] = tol
    with open(os.path.join(ROOT, "tests", "golden", "bf16_tolerances.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
