#!/usr/bin/env python
"""Why do the SECOND workgroups of a CU finish 10 us after the first ones in the step's 64x64-tile launches (profiles/r05s)?

One encoder product (M x N x K = 2624 x 768 x 768, bias + residual + dropout -> the 64x64 tile, 492 workgroups = 2 on 236 CUs) on the
stamps build (python -m clipbert_amd.build --stamps) in several surroundings; for every launch the end time of the workgroups in wave
slot 0 and wave slot 1 of their SIMD (first / second workgroup of a CU), relative to the launch's first entry.

    python tools/tail_probe.py [--lib stamps]
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HDR, REC, WGS = 3, 6, 512


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default="stamps")
    ap.add_argument("--M", type=int, default=2624)
    ap.add_argument("--N", type=int, default=768)
    ap.add_argument("--K", type=int, default=768)
    a = ap.parse_args()
    from clipbert_amd import _lib, ops
    from clipbert_amd.build import variant_path
    lib = _lib.load(variant_path(a.lib))
    _lib._LIB = lib
    lib.cb_debug_stamps_begin.argtypes = [C.c_void_p, C.c_int64]
    lib.cb_debug_stamps_area_words.restype = C.c_int64
    lib.cb_debug_stamps_count.restype = C.c_int64
    lib.cb_debug_stamps_desc.argtypes = [C.c_int64, C.c_char_p, C.c_int64]
    dev = torch.device("cuda", 0)
    dt = torch.bfloat16
    M, N, K = a.M, a.N, a.K
    words = int(lib.cb_debug_stamps_area_words())
    n_areas = 64
    buf = torch.zeros(n_areas * words, dtype=torch.int64, device=dev)
    ops.splitk_workspace(dev)
    NB = 6
    A = [torch.randn(M, K, device=dev).to(dt) for _ in range(NB)]
    W = [(torch.randn(N, K, device=dev) * 0.02).to(dt) for _ in range(NB)]
    R = [torch.randn(M, N, device=dev).to(dt) for _ in range(NB)]
    Y = [torch.empty(M, N, dtype=dt, device=dev) for _ in range(NB)]
    bias = torch.zeros(N, dtype=torch.float32, device=dev)
    src = torch.randn(M, K, device=dev).to(dt)
    big = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    seed = torch.zeros(1, dtype=torch.int64, device=dev)

    def g(i, plain=False, tile=0):
        j = i % NB
        if plain:
            ops.gemm(A[j], W[j], M, N, K, out=Y[j], shift=bias, tile=tile)
        else:
            ops.gemm(A[j], W[j], M, N, K, out=Y[j], shift=bias, residual=R[j], dropout_p=0.1, dropout_seed=1234 + i, seed_ptr=seed, tile=tile)

    def begin():
        buf.zero_()
        buf.view(n_areas, words)[:, 0] = -1
        torch.cuda.synchronize()
        lib.cb_debug_stamps_begin(C.c_void_p(buf.data_ptr()), buf.numel() * 8)

    def report(label):
        torch.cuda.synchronize()
        n = int(lib.cb_debug_stamps_count())
        raw = buf.view(n_areas, words)[:n].cpu().numpy().astype(np.uint64)
        descs = []
        for i in range(n):
            b = C.create_string_buffer(1024)
            lib.cb_debug_stamps_desc(i, b, 1024)
            descs.append(json.loads(b.value.decode()))
        lib.cb_debug_stamps_begin(None, 0)
        print(f"== {label}: {n} stamped launches")
        for i in range(n):
            ar = raw[i]
            tmin, tmax, nwg = int(ar[0]), int(ar[1]), int(ar[2])
            if nwg == 0:
                continue
            d = descs[i]
            rec = ar[HDR:HDR + min(nwg, WGS) * REC].reshape(-1, REC).astype(np.int64)
            rec = rec[rec[:, 0] > 0]
            end = (rec[:, 4] - tmin) / 100.0
            epi = (rec[:, 3] - rec[:, 2]) / 100.0
            kl = (rec[:, 2] - rec[:, 1]) / 100.0
            slot = (rec[:, 5] & 0xf)
            parts = []
            for s in sorted(set(slot.tolist())):
                sel = slot == s
                parts.append(f"slot{s}: n {int(sel.sum())} k {np.median(kl[sel]):.1f} epi {np.median(epi[sel]):.1f} end med {np.median(end[sel]):.1f} max {end[sel].max():.1f}")
            print(f"  #{i} {d['M']}x{d['N']}x{d['K']} tile {d['tile']} wgs {nwg} wall {(tmax - tmin) / 100.0:.1f} | " + " | ".join(parts))

    for i in range(3):
        g(i)
    torch.cuda.synchronize()

    begin()
    for i in range(4):
        g(0)
    report("S1 back to back, same buffers (hot)")

    begin()
    for i in range(6):
        g(i)
    report("S2 back to back, rotating buffers")

    begin()
    for i in range(4):
        big.fill_(i)
        g(i)
    report("S3 each launch after a 256 MB fill (cold caches)")

    begin()
    for i in range(4):
        A[i % NB].copy_(src)
        g(i)
    report("S4 each launch after a kernel that writes its A operand")

    begin()
    for i in range(4):
        g(i, plain=True)
    report("S5 bias only (no residual / dropout)")

    begin()
    for i in range(4):
        g(i, tile=3)
    report("S6 tile 128x64 (252 workgroups)")

    # captured: the same sequences as a hipGraph
    lib.cb_debug_stamps_begin(None, 0)
    torch.cuda.synchronize()
    buf.zero_(); buf.view(n_areas, words)[:, 0] = -1
    torch.cuda.synchronize()
    lib.cb_debug_stamps_begin(C.c_void_p(buf.data_ptr()), buf.numel() * 8)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(6):
            g(i)
        for i in range(3):
            A[i % NB].copy_(src)
            g(i)
    for _ in range(3):
        buf.view(n_areas, words)[:, 1:].zero_(); buf.view(n_areas, words)[:, 0] = -1
        gr.replay()
    report("S7 hipGraph replay: 6 back to back (rotating), then 3 after a writer of A")


if __name__ == "__main__":
    main()
