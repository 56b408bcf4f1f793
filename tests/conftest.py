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


# ---- the GPU suite's time budget (the driver gives `pytest -m gpu` 900 s) ----
# HMRM_SUITE_BUDGET_S (default 420): wall time the whole suite aims to stay inside.  The time-boxed fuzz slices share
# HMRM_FUZZ_BUDGET_S (default 270) in fixed proportions, and the two heaviest optional cases (a map with a side of
# 2^24 cells, a frame 524 325 rows tall) run only while the suite is inside its budget -- skipped with a message
# otherwise, never silently.  The suite's wall time is written to gpurun_out/gpu_suite_wall.txt at the end.
import time

_SUITE_T0 = time.time()
SUITE_BUDGET_S = float(os.environ.get("HMRM_SUITE_BUDGET_S", "420"))
FUZZ_BUDGET_S = float(os.environ.get("HMRM_FUZZ_BUDGET_S", "270"))


def suite_elapsed_s() -> float:
    return time.time() - _SUITE_T0


def skip_if_over_budget(cost_s: float, what: str):
    """For the optional heavy cases: run only if the suite would stay inside its budget."""
    if suite_elapsed_s() + cost_s > SUITE_BUDGET_S:
        pytest.skip(f"{what}: suite at {suite_elapsed_s():.0f} s of its {SUITE_BUDGET_S:.0f} s budget "
                    f"(HMRM_SUITE_BUDGET_S), the case costs ~{cost_s:.0f} s")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionfinish(session, exitstatus):
    markexpr = getattr(session.config.option, "markexpr", "") or ""
    if markexpr.strip() != "gpu":
        return
    try:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "gpu_suite_wall.txt"), "w") as f:
            f.write(f"pytest -m gpu: {suite_elapsed_s():.1f} s wall, exit status {int(exitstatus)}, {session.testscollected} collected, "
                    f"{session.testsfailed} failed; budget {SUITE_BUDGET_S:.0f} s (fuzz slices {FUZZ_BUDGET_S:.0f} s)\n")
    except OSError:
        pass


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
