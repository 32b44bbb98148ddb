#!/usr/bin/env python
"""Does a hipMemsetAsync issued by the library INSIDE a hipGraph capture clear its buffer on every replay?  (round 5: the captured
training step of rounds 2-4 fed stale bytes into the ResNet backward from its second replay on.)  Uses cb_maxpool2_bwd of the ROUND-4
library build (memset + scatter kernel) and of the current one (self-zeroing kernel) on a 7x7 map whose last row / column no pooling
window covers:  python tools/memset_capture_probe.py [--lib r04]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--lib", default=""); a = ap.parse_args()
    from clipbert_amd import _lib, ops
    if a.lib:
        from clipbert_amd.build import variant_path
        _lib._LIB = _lib.load(variant_path(a.lib))
    dev = "cuda"
    n, h, c = 16, 7, 768
    x = torch.randn(n, h, h, c, device=dev).bfloat16()
    y = torch.randn(n, 3, 3, c, device=dev).bfloat16().abs()
    dy = torch.randn(n, 3, 3, c, device=dev).bfloat16()
    outs = []
    def step():
        junk = torch.empty(n, h, h, c, device=dev, dtype=torch.bfloat16)
        junk.fill_(1e30)                                    # whatever used the block before
        del junk
        dx = ops.maxpool2_bwd(x, y, dy, relu=True)          # (allocated from the same pool: lands on the freed block)
        outs.append(dx)
        return dx
    step(); torch.cuda.synchronize()
    print("eager: uncovered row max |dx| =", float(outs[-1][:, 6].abs().max()), " col:", float(outs[-1][:, :, 6].abs().max()))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        dx = step()
    for i in range(3):
        g.replay(); torch.cuda.synchronize()
        print(f"replay{i}: uncovered row max |dx| = {float(dx[:, 6].abs().max()):.3e}  col: {float(dx[:, :, 6].abs().max()):.3e}  covered finite: {bool(torch.isfinite(dx[:, :6, :6]).all())}")
        dx.fill_(1e30)                                      # poison between replays: the next replay must clear it again
        torch.cuda.synchronize()

if __name__ == "__main__":
    main()
