import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Collection order under `-m gpu -x`: kernel parity first, then whole-model / golden parity, then the host-logic rows on the
# GPU, and subprocess / control-flow tests LAST -- a brittle late test must never mask the parity rows behind it.
_ORDER = ["test_abi", "test_kernels_gemm", "test_kernels_gemm8", "test_kernels_misc", "test_gemm_fuzz", "test_model_small", "test_gpu_full", "test_parity_record",
          "test_clips", "test_tasks", "test_checkpoint", "test_loops", "test_comm"]


def _rank(item):
    name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    if name.startswith("test_zz"):
        return len(_ORDER) + 1
    return _ORDER.index(name) if name in _ORDER else len(_ORDER)


def pytest_collection_modifyitems(config, items):
    import torch
    items.sort(key=_rank)                      # stable: the order inside a file is kept
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def emul_lib():
    """The product HIP sources compiled against the host lane-level emulator (tests/emul)."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "emul"))
    import build_emul
    import ctypes
    from clipbert_amd import _lib
    path = build_emul.build()
    return _lib.bind(ctypes.CDLL(path), strict=False)


@pytest.fixture()
def emul(emul_lib, monkeypatch):
    """Route clipbert_amd.ops through the emulator build (CPU tensors, host pointers) for one test."""
    from clipbert_amd import _lib, ops
    monkeypatch.setattr(_lib, "_LIB", emul_lib)
    monkeypatch.setattr(ops, "_ALLOW_HOST_POINTERS", True)
    return emul_lib


class _HW:
    def __init__(self, name, dev):
        self.name, self.dev = name, dev

    def __call__(self, t):
        """move a tensor (or None) to the backend's device"""
        return None if t is None else t.to(self.dev)


@pytest.fixture(params=["emul", pytest.param("gpu", marks=pytest.mark.gpu)])
def hw(request, monkeypatch):
    """Backend under test: the host lane-level emulator build (CPU tensors) or the real library on cuda:0."""
    import torch
    if request.param == "emul":
        lib = request.getfixturevalue("emul_lib")
        from clipbert_amd import _lib, ops
        monkeypatch.setattr(_lib, "_LIB", lib)
        monkeypatch.setattr(ops, "_ALLOW_HOST_POINTERS", True)
        return _HW("emul", torch.device("cpu"))
    return _HW("gpu", torch.device("cuda", 0))
