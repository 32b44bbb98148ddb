"""The opt-in LDS-DMA ring kernels (CB_GEMM_DMA=1) against the register-staged kernels on the encoder's GEMM shapes, one process
per configuration (the switches are read once).  Round 2 also ran 4- and 5-stage rings on the 128x128 tile (results in
DESIGN.md section 7 and the comment in csrc/gemm_impl.h); those instantiations were removed again."""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [("fwd", 2624, 3072, 768), ("fwd", 2624, 2304, 768), ("fwd", 2624, 768, 3072), ("fwd", 5440, 3072, 768), ("fwd", 8192, 8192, 1024),
          ("fwd", 8192, 8192, 4096), ("fwd", 10496, 3072, 768), ("fwd", 10496, 2304, 768), ("fwd", 10496, 768, 3072), ("fwd", 10496, 768, 768),
          ("dgrad", 2624, 3072, 768), ("dgrad", 2624, 768, 3072)]


def child():
    import torch
    from clipbert_amd import ops
    from tools.tune_gemm import time_config
    dev = torch.device("cuda", 0)
    tile = int(os.environ["PROBE_TILE"])
    for form, M, N, K in SHAPES:
        a = (torch.rand(M, K, device=dev) - 0.5).bfloat16()
        if form == "fwd":
            b = (torch.rand(N, K, device=dev) - 0.5).bfloat16(); kw = {}
        else:
            b = (torch.rand(K, N, device=dev) - 0.5).bfloat16(); kw = dict(b_mode=ops.KROW, ldb=N)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        us, err = time_config((a, b, M, N, K), dict(kw, out=out), tile, 1)
        print(f"  {form:5s} {M:5d}x{N:5d}x{K:5d}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF" if us else f"  {form} {M}x{N}x{K}: {err}", flush=True)


if __name__ == "__main__":
    if os.environ.get("PROBE_TILE"):
        child()
    else:
        for name, env in [("register ring 128x128 occ2 (tile 4)", dict(PROBE_TILE="4")),
                          ("register ring 128x128 PF=2 (tile 1)", dict(PROBE_TILE="1")),
                          ("register ring 64x64 (tile 2)", dict(PROBE_TILE="2")),
                          ("DMA ring 128x128, 3 stages", dict(PROBE_TILE="1", CB_GEMM_DMA="1", CB_GEMM_DMA_KROW="1")),
                          ("DMA ring 128x128, 2 stages, two blocks per CU (tile 4)", dict(PROBE_TILE="4", CB_GEMM_DMA="1", CB_GEMM_DMA_KROW="1")),
                          ("DMA ring 64x64, 4 stages", dict(PROBE_TILE="2", CB_GEMM_DMA="1", CB_GEMM_DMA_KROW="1")),
                          ("DMA ring 64x64, 8 stages", dict(PROBE_TILE="2", CB_GEMM_DMA="1", CB_GEMM_DMA_KROW="1", CB_GEMM_DMA_DEEP="1"))]:
            print(name, flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, CB_GEMM_NO_TUNED="1", **env))
