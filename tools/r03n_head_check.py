"""Why did the bf16 error of the retrieval_rank golden move (5.5e-4 -> 1.3e-3) when 1-column head outputs became contiguous?
(1) cb_gemm M x 1 / M x 2 outputs with ldc = N vs 4 against torch on the GPU; (2) the golden case with both layouts."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from clipbert_amd import ops, modeling as M  # noqa: E402
from oracle import make_golden as G  # noqa: E402
from test_gpu_full import GOLDEN, build_model, to_dev  # noqa: E402
dev = torch.device("cuda", 0)
torch.manual_seed(0)
for m in (4, 5, 64):
    K = 1536
    for N in (1, 2):
        a = (torch.randn(m, K, device=dev) * 0.5).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.02).bfloat16(); b = torch.randn(N, device=dev) * 0.01
        ref = a.float() @ w.float().t() + b
        outs = []
        for ld in (N, 4):
            store = torch.full((m, ld), 7.0, dtype=torch.float32, device=dev); y = store[:, :N]
            ops.gemm(a, w, m, N, K, out=y, shift=b)
            outs.append(y.clone())
            print(f"M={m} N={N} ld={ld}: max err {float((y - ref).abs().max()):.3e}", flush=True)
        print("   identical:", torch.equal(outs[0], outs[1]))
for name in ("retrieval_rank", "tgif_mc"):
    gold = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg, head, sd, batch = G.build_case(name)
    for pad in (False, True):
        M._PAD_SMALL_HEADS = pad
        for rep in range(2):
            model = build_model(cfg, head, sd, torch.bfloat16)
            with torch.no_grad():
                lg = model(to_dev(batch))["logits"].float().cpu().numpy()
            print(f"{name}: padded={pad} run {rep}: max |delta| vs golden {np.abs(lg.reshape(gold['logits'].shape) - gold['logits']).max():.6e}", flush=True)
