import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _ensure_built():
    """Test modules import the package at collection time: build the in-tree libraries first
    if this is a fresh checkout (hipcc cross-compiles without a GPU)."""
    lib_path = os.path.join(ROOT, "heightmap-ray-marcher_amd", "libhmrm.so")
    oracle_path = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
    if not (os.path.exists(lib_path) and os.path.exists(oracle_path)):
        import __graft_entry__
        __graft_entry__.build()


_ensure_built()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hmrm():
    """The product package (loads the in-tree libhmrm.so; import fails if it is not built)."""
    lib_path = os.path.join(ROOT, "heightmap-ray-marcher_amd", "libhmrm.so")
    if not os.path.exists(lib_path):
        import __graft_entry__
        __graft_entry__.build()
    return importlib.import_module("heightmap-ray-marcher_amd")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle binding (test infrastructure)."""
    from oracle import oracle_py
    oracle_py.build()
    return oracle_py


@pytest.fixture(scope="session")
def stb_ref():
    """The reference's own stb build (oracle/_ref), or None where it was never built."""
    import stb_ref as _s
    return _s.load()
