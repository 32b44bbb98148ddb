#!/usr/bin/env python
"""The encoder's forward products at the inference row's batch (256 pairs x 41 tokens = 10496 rows), per tile, hot:
    python tools/infer_shapes_probe.py            (CB_GEMM_FAST_EPI=0 for the generic epilogue)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gemm_yardstick import graph_time  # noqa: E402

from clipbert_amd import ops  # noqa: E402

if os.environ.get("CB_LIB_VARIANT"):          # a diagnostic copy of the library (python -m clipbert_amd.build --variant NAME --csrc DIR)
    from clipbert_amd import _lib
    from clipbert_amd.build import variant_path
    import ctypes
    _lib._LIB = _lib.bind(ctypes.CDLL(variant_path(os.environ["CB_LIB_VARIANT"])), strict=False)      # (an older ABI: symbols added since are absent)

dev, dt = "cuda", torch.bfloat16
g = torch.Generator(device=dev).manual_seed(1)
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g, dtype=torch.float32) - 0.5).to(dt)
ops.splitk_workspace(torch.device(dev))
M = int(sys.argv[1]) if len(sys.argv) > 1 else 10496
for name, N, K, kw in (("QKV (bias)", 2304, 768, {}), ("FFN1 (bias + GELU)", 3072, 768, dict(act=ops.ACT_GELU)),
                       ("attn.out (bias + residual)", 768, 768, dict(res=True)), ("FFN2 (bias + residual)", 768, 3072, dict(res=True))):
    x, w, bias = rnd(M, K), rnd(N, K), rnd(N).float()
    y = torch.empty(M, N, device=dev, dtype=dt)
    extra = dict(shift=bias)
    if kw.get("res"):
        extra["residual"] = rnd(M, N)
    if "act" in kw:
        extra["act"] = kw["act"]
    line = [f"{name} {M}x{N}x{K}: plan {ops.gemm_plan(x, w, M, N, K, out=y, **extra)}"]
    for tile, sched in ((0, 0), (5, 1), (5, 3), (6, 1), (6, 3), (7, 1), (7, 3), (4, 0), (1, 0)):
        try:
            t = graph_time(lambda: ops.gemm(x, w, M, N, K, out=y, tile=tile, schedule=sched, **extra), 16)
            line.append(f"t{tile}/m{sched}: {t:.1f}")
        except Exception as e:      # noqa: BLE001
            line.append(f"t{tile}/m{sched}: {str(e)[:30]}")
    print("  ".join(line), flush=True)
