"""bf16 max |delta| of one full-size golden case on the GPU: python tools/golden_bf16_error.py <case>  (try CB_GEMM_NO_MODEL=1 / CB_GEMM_TRACE=1:
another tile or K split is another fp32 summation order and re-draws the bf16 rounding noise downstream)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from clipbert_amd import ops, modeling as M  # noqa: E402
from oracle import make_golden as G  # noqa: E402
from test_gpu_full import GOLDEN, build_model, to_dev  # noqa: E402
name = sys.argv[1]
gold = np.load(os.path.join(GOLDEN, name + ".npz"))
cfg, head, sd, batch = G.build_case(name)
model = build_model(cfg, head, sd, torch.bfloat16)
with torch.no_grad():
    lg = model(to_dev(batch))["logits"].float().cpu().numpy()
print(f"{name}: max |delta| vs golden {np.abs(lg.reshape(gold['logits'].shape) - gold['logits']).max():.6e}", flush=True)
