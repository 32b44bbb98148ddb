"""LayerNorm forward / backward timings at the encoder's token counts (hipGraph timing, tools/gemm_bench.timeit).
CB_LN_BWD_BLOCKS caps the backward's grid (its 2*D atomics per block bound the useful number of blocks)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clipbert_amd import ops
from tools.gemm_bench import timeit, dev
g = torch.ones(768, device=dev); b = torch.zeros(768, device=dev)
for rows in (1312, 2624, 5440):
    x = torch.randn(rows, 768, device=dev).bfloat16(); dy = torch.randn_like(x)
    y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-12, save_stats=True)
    dg, db = torch.zeros(768, device=dev), torch.zeros(768, device=dev)
    dx2 = torch.empty_like(x)
    print(f"rows {rows}: ln fwd {timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-12, save_stats=True)):.1f} us, "
          f"ln bwd {timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dg, db)):.1f} us, "
          f"ln bwd + dropped copy {timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dg, db, dropout_p=0.1, dropout_seed=3, dx2=dx2)):.1f} us "
          f"(CB_LN_BWD_BLOCKS={os.environ.get('CB_LN_BWD_BLOCKS', 'default')})", flush=True)
