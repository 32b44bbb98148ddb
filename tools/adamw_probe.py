"""cb_adamw over a 96 M-element parameter range (the step's largest group) timed in a hipGraph: GB/s against the 30 bytes per element
it must move (16 read + 14 written).  Round 2: 5.7-6.0 TB/s (the streaming-copy rate of this part is ~6.3 TB/s); a variant with two
4-element chunks per thread (eight loads in flight, half the blocks) measured 5.6 TB/s and was dropped."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clipbert_amd import ops

dev = torch.device("cuda", 0)
n = 96 * 1024 * 1024 + 7
p, g, m, v = (torch.randn(n, device=dev) * 0.01 for _ in range(4))
v.abs_()
w16 = torch.empty(n, dtype=torch.bfloat16, device=dev)
hp = torch.tensor(ops.adamw_hyper(1e-4, 0.9, 0.98, 1e-6, 1e-3, 10, 5.0, 1.0) + [0.0] * 6, dtype=torch.float32, device=dev)
sq = torch.tensor([4.0], device=dev)
fn = lambda: ops.adamw(p, g, m, v, w16, hp, sq)
fn(); torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    for _ in range(5):
        fn()
gr.replay(); torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 5)
print(f"cb_adamw, {n} elements: {best * 1e3:.1f} us per launch, {n * 30 / best / 1e6:.0f} GB/s")
