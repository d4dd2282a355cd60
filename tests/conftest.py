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


def check_and_sync_f16x2_state(L, ora, max_outliers=12, max_fraction=5e-4):
    """End-to-end parameter check of a learner running the f16x2 arithmetic (~22-bit operands; its gradients are
    pinned at 1e-4 of their rms at kernel level).  Adam's g / (sqrt(v) + eps) turns a 1e-4-of-rms gradient difference
    on an entry near eps (1e-5) into a visible step, so: all but a handful of parameters (at most 12, or 0.05 % of a large
    network's) at 1e-4, every one at 1e-3;
    then the learner takes over the oracle's parameters and Adam moments (rounded to f32, which the oracle adopts
    too) so that every update is compared from an IDENTICAL state and the few amplified entries do not compound."""
    import numpy as np
    import torch

    Pa = L.Pa
    for name, got, want in (("actor", L.p[:Pa], ora.pa), ("critic", L.p[Pa:], ora.pc)):
        got = got.cpu().numpy().astype(np.float64)
        bad = np.abs(got - want) > 1e-4 * (np.abs(want) + np.sqrt(np.mean(want * want)))
        assert bad.sum() <= max(max_outliers, max_fraction * bad.size), f"{name} params: {int(bad.sum())} entries outside 1e-4"
        assert_close(got, want, 1e-3, f"{name} params (hard bound)")
    for dst, a_, c_ in ((L.p, ora.pa, ora.pc), (L.m, ora.ma, ora.mc), (L.v, ora.va, ora.vc)):
        dst[:Pa].copy_(torch.from_numpy(a_.astype(np.float32)))
        dst[Pa:].copy_(torch.from_numpy(c_.astype(np.float32)))
    ora.pa, ora.pc = L.p[:Pa].cpu().numpy().astype(np.float64), L.p[Pa:].cpu().numpy().astype(np.float64)
    ora.ma, ora.mc = L.m[:Pa].cpu().numpy().astype(np.float64), L.m[Pa:].cpu().numpy().astype(np.float64)
    ora.va, ora.vc = L.v[:Pa].cpu().numpy().astype(np.float64), L.v[Pa:].cpu().numpy().astype(np.float64)
