"""Shared helpers for parity tests."""
import numpy as np

# Float32 tolerances (stated in DESIGN.md):
#  operators: normwise relative error <= 1e-5, max-abs error relative to the
#  reference's max-abs <= 1e-4; solver state after 3 epochs: normwise 1e-3;
#  costs: rtol 1e-4.
OP_NORMWISE = 1e-5
OP_MAXABS = 1e-4
SOLVER_NORMWISE = 1e-3
COST_RTOL = 1e-4


def relerr(a, b):
    """Normwise relative error ||a - b|| / ||b||."""
    a = np.asarray(a)
    b = np.asarray(b)
    den = np.linalg.norm(b.ravel())
    return float(np.linalg.norm((a - b).ravel()) / (den if den > 0 else 1.0))


def maxerr(a, b):
    """max|a - b| / max|b|."""
    a = np.asarray(a)
    b = np.asarray(b)
    den = np.abs(b).max()
    return float(np.abs(a - b).max() / (den if den > 0 else 1.0))


def assert_close(a, b, normwise=OP_NORMWISE, maxabs=OP_MAXABS, what=""):
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    r, m = relerr(a, b), maxerr(a, b)
    assert r <= normwise and m <= maxabs, (
        f"{what}: normwise {r:.3e} (tol {normwise:.1e}), "
        f"max-abs {m:.3e} (tol {maxabs:.1e})")
