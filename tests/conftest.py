import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The tests bind libmidyn.so: build it when it is missing or was built from other sources (a fresh
    checkout has no binary -- *.so is git-ignored; hipcc cross-compiles gfx950 without a GPU)."""
    import importlib

    try:
        entry = importlib.import_module("__graft_entry__")
        if not entry.library_is_current():   # content hash of the sources vs the stamp written by build()
            entry.build()
    except Exception as err:  # pylint: disable=broad-except
        print(f"[conftest] could not build libmidyn.so: {err}", file=sys.stderr)


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
        return cache[name]

    return load


def assert_close(a, b, tol=1e-12):
    """max|a-b| <= tol * (1 + max|b|)  -- the parity criterion of SURVEY.md 8(d)."""
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, f"shape {a.shape} != {b.shape}"
    err = float(np.max(np.abs(a - b))) if a.size else 0.0
    scale = 1.0 + (float(np.max(np.abs(b))) if b.size else 0.0)
    assert err <= tol * scale, f"max|diff|={err:.3e} > {tol:.1e}*(1+{scale - 1:.3e})"


@pytest.fixture
def per_launch_routes():
    """For the tests that pin the launch-per-product routes (work-list MFMA / streaming kernels) through their launch
    counters: the one-launch kernels that would otherwise take sweeps on very sparse stacks (ell_sweep_kernel) are
    switched off for the duration of the test."""
    import qiskit_dynamics_amd as q

    ctx = q.default_context()
    ctx.set_option("ell_sweep", 0)
    try:
        yield
    finally:
        ctx.set_option("ell_sweep", 1)
