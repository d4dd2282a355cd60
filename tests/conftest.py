import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def assert_close(a, b, rtol, name="", scale=None):
    """|a-b| <= rtol*|b| + rtol*rms(b)  (BASELINE.md §2 tolerance form)."""
    import numpy as np

    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    if scale is None:
        scale = float(np.sqrt(np.mean(b * b))) if b.size else 0.0
    err = np.abs(a - b)
    tol = rtol * np.abs(b) + rtol * scale
    bad = err > tol
    if bad.any():
        i = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError(
            f"{name}: {int(bad.sum())}/{bad.size} outside rtol={rtol} (scale={scale:.3e}); worst at {i}: "
            f"got {a[i]!r} want {b[i]!r} err {err[i]:.3e}"
        )
