import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
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
