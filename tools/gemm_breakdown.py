#!/usr/bin/env python
"""Log every cb_gemm call (shape, form) of ONE eager training step of the bench workload, for joining with a
rocprofv3 kernel trace of the same process (tools/join_gemm_trace.py)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from clipbert_amd import modeling as M, ops, synthetic as S
from clipbert_amd.optim import FusedAdamW

dev = torch.device("cuda", 0)
cfg = dict(bench.BASE_CONFIG)
model = M.ClipBert(cfg, detectron2_model_cfg="R-50-grid.yaml", transformer_cls=M.ClipBertForVideoTextRetrieval)
model.load_state_dict(S.full_state_dict(cfg, "retrieval", 42), strict=True)
model.to(dev).train()
model.prepare(dtype=torch.bfloat16, device=dev, overlap_wgrad=False)
opt = FusedAdamW(model.rt.bank, lr=5e-5, betas=(0.9, 0.98), weight_decay=1e-3, max_grad_norm=5.0)
bv = 16
vis = bench.ops_image_norm_host(S.synthetic_frames(bv, 2, 224, 42)).to(dev)
ids, mask = S.synthetic_text(bv * 2, 32, 42)
ids, mask = ids.to(dev), mask.to(dev)
labels = torch.tensor([1, 0] * bv, dtype=torch.long, device=dev)

def step():
    opt.zero_grad()
    out = model(dict(visual_inputs=vis, text_input_ids=ids, text_input_mask=mask, n_examples_list=[2] * bv))
    _, loss = model.transformer.calc_loss(out["logits"], labels, sample_size=bv)
    loss.mean().backward()
    opt.step()

for _ in range(2):
    step()
torch.cuda.synchronize()
calls = []
orig = ops.gemm
def logged(a, b, Mm, N, K, **kw):
    form = "wgrad" if kw.get("a_mode", 0) == ops.KROW else ("dgrad" if kw.get("b_mode", 0) in (ops.KROW, ops.KROW_TAPS) else "fwd")
    conv = kw.get("a_mode", 0) == ops.ROWK_GATHER or kw.get("b_mode", 0) == ops.KROW_GATHER
    calls.append(dict(M=Mm, N=N, K=K, form=form, conv=bool(conv), split=kw.get("split_k", 1), R=kw.get("R", 1), batch=kw.get("batch", 1),
                      esz=a.element_size(), c_esz=kw["out"].element_size()))
    return orig(a, b, Mm, N, K, **kw)
ops.gemm = logged
step()
torch.cuda.synchronize()
ops.gemm = orig
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(calls, open(os.path.join(ROOT, "gpurun_out", "gemm_calls.json"), "w"))
print("logged", len(calls), "gemm calls")
