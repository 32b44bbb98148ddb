#!/usr/bin/env python
"""Per-launch time of the encoder's N = 768 products by tile, UNSTAMPED (product library): a hipGraph of 48 launches of one product
(rotating operand / output buffers), replayed; time per launch from events around the replays.  Companion of tools/tail_probe.py.

    python tools/tile_time_probe.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from clipbert_amd import ops
    from clipbert_amd.ops import KROW
    dev = torch.device("cuda", 0)
    dt = torch.bfloat16
    ops.splitk_workspace(dev)
    M = 2624
    NB = 6
    seed = torch.zeros(1, dtype=torch.int64, device=dev)
    TILES = {0: "auto", 2: "64x64", 3: "128x64", 1: "128x128", 4: "128x128o2"}
    cases = [("fwd  N768 K768  bias+res+drop", 768, 768, False, True), ("fwd  N768 K3072 bias+res+drop", 768, 3072, False, True),
             ("dgrad N768 K768  plain", 768, 768, True, False), ("dgrad N768 K2304 residual", 768, 2304, True, False),
             ("dgrad N768 K3072 residual", 768, 3072, True, False)]
    for label, N, K, dgrad, drop in cases:
        A = [torch.randn(M, K, device=dev).to(dt) for _ in range(NB)]
        W = [(torch.randn(K, N, device=dev) * 0.02).to(dt) if dgrad else (torch.randn(N, K, device=dev) * 0.02).to(dt) for _ in range(NB)]
        R = [torch.randn(M, N, device=dev).to(dt) for _ in range(NB)]
        Y = [torch.empty(M, N, dtype=dt, device=dev) for _ in range(NB)]
        bias = torch.zeros(N, dtype=torch.float32, device=dev)
        res = {}
        for tile, tname in TILES.items():
            def g(i):
                j = i % NB
                kw = dict(out=Y[j], tile=tile)
                if dgrad:
                    kw.update(b_mode=KROW)
                    if "residual" in label:
                        kw.update(residual=R[j])
                else:
                    kw.update(shift=bias, residual=R[j])
                    if drop:
                        kw.update(dropout_p=0.1, dropout_seed=77 + i, seed_ptr=seed)
                ops.gemm(A[j], W[j], M, N, K, **kw)
            for i in range(3):
                g(i)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            L = 48
            with torch.cuda.graph(gr):
                for i in range(L):
                    g(i)
            for _ in range(3):
                gr.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            reps = 10
            for _ in range(reps):
                gr.replay()
            e1.record()
            torch.cuda.synchronize()
            res[tname] = e0.elapsed_time(e1) * 1000.0 / (reps * L)
        print(f"{label}: " + "  ".join(f"{k} {v:.1f}" for k, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
