#!/usr/bin/env python
"""Log every cb_gemm call (shape, form, epilogue operands) of ONE eager step of the bench workload (bench.py --mode <mode>, default
train = the metric configuration), for joining with a rocprofv3 kernel trace / counter collection of the same process
(tools/join_gemm_trace.py, tools/pmc_traffic.py): the LAST len(calls) cb_gemm dispatches of the trace are this step's."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import tune_gemm  # noqa: E402
from clipbert_amd import ops  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "train"
recorded = tune_gemm.record_calls(mode)           # the second of bench.py's two eager warm-up steps (it ends with a synchronize)
calls = []
for (a, b, Mm, N, K), kw in recorded:
    form = "wgrad" if kw.get("a_mode", 0) == ops.KROW else ("dgrad" if kw.get("b_mode", 0) in (ops.KROW, ops.KROW_TAPS) else "fwd")
    conv = kw.get("a_mode", 0) == ops.ROWK_GATHER or kw.get("b_mode", 0) == ops.KROW_GATHER
    # additional M x N operand passes of the epilogue, in elements of the activation type (C itself is counted by the reader)
    extra = sum(1 for k in ("residual", "mask", "out2", "gelu_grad_pre") if kw.get(k) is not None)
    extra += 1 if (kw.get("accumulate") and form != "wgrad") else 0
    calls.append(dict(M=Mm, N=N, K=K, form=form, conv=bool(conv), split=kw.get("split_k", 1), R=kw.get("R", 1), S=kw.get("S", 1),
                      batch=kw.get("batch", 1), esz=a.element_size(), c_esz=kw["out"].element_size(), extra_mn=extra))
torch.cuda.synchronize()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(calls, open(os.path.join(ROOT, "gpurun_out", "gemm_calls.json"), "w"))
print("logged", len(calls), "gemm calls")
