#!/usr/bin/env python
"""cb_res2_block next to the three / four cb_gemm launches it replaces, on the metric's res2 shape (64 frames, 56 x 56): us per block,
hot (16 launches back to back in a hipGraph) and cold (behind a 384 MB flush).   python tools/res2_probe.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from gemm_yardstick import hot_cold

def main():
    from clipbert_amd.bench import step as bench_step
    from clipbert_amd import modeling as M
    st = bench_step.build(videos=2)
    rt = st.model.rt
    res2 = st.model.cnn.feature.backbone.res2
    for bi, cin in ((0, 64), (1, 256)):
        blk = res2[bi]
        x = torch.randn(64, 56, 56, cin, device="cuda").relu().bfloat16()
        def unfused():
            sc = M._conv_fwd(rt, x, blk.shortcut) if blk.shortcut is not None else x
            y1 = M._conv_fwd(rt, x, blk.conv1, act=M.ACT_RELU)
            y2 = M._conv_fwd(rt, y1, blk.conv2, act=M.ACT_RELU)
            return M._conv_fwd(rt, y2, blk.conv3, residual=sc, relu_after=True)
        a, b = unfused(), M._res2_block_fused(rt, x, blk)
        torch.cuda.synchronize()
        err = (a.float() - b.float()).abs()
        byts = 200704 * (cin * 1.5625 + 256) * 2
        hu, hf = hot_cold(unfused, x), hot_cold(lambda: M._res2_block_fused(rt, x, blk), x)
        print(f"res2 block {bi} (cin {cin}): unfused {hu[0]} / {hu[1]} us (hot / cold), fused {hf[0]} / {hf[1]} us = {byts / hf[1] / 1e6:.2f} TB/s of x(+halo) + out; "
              f"max |diff| {float(err.max()):.4f}, differing elements {float((err > 0).float().mean()) * 100:.2f} %", flush=True)

if __name__ == "__main__":
    main()
