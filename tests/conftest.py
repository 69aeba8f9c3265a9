import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


@pytest.fixture(scope="session")
def oracle():
    from oracle import awq_oracle

    awq_oracle.build()
    return awq_oracle


def product_tol(ref32):
    """The stated fp16-product tolerance (SURVEY.md 8c / BASELINE.md 4):
    |y - ref| <= 1e-3*|ref| + 1e-3*rms(ref)."""
    ref32 = np.asarray(ref32, np.float64)
    rms = float(np.sqrt(np.mean(ref32 ** 2))) if ref32.size else 0.0
    return 1e-3 * np.abs(ref32) + 1e-3 * rms


def assert_product_close(y, ref32, what="", wsigma=None):
    """|y - ref| <= 1e-3*|ref| + 1e-3*rms(ref) + 1 fp16 ulp of the output (+ 6 sigma of the
    reference's own fp16 weight-rounding noise when `wsigma` is given: the oracle multiplies
    fp16-ROUNDED weights, the MFMA kernels apply the scale after an exact integer dot product, so
    the two differ by that noise -- oracle.weight_rounding_sigma)."""
    y = np.asarray(y, np.float64)
    ref32 = np.asarray(ref32, np.float64)
    tol = product_tol(ref32)
    if wsigma is not None:
        tol = tol + 6.0 * np.asarray(wsigma, np.float64)
    # one fp16 ulp of slack for the final rounding of the output itself
    ulp = np.maximum(np.abs(ref32), 2.0 ** -14) * 2.0 ** -10
    bad = np.abs(y - ref32) > tol + ulp
    if bad.any():
        idx = np.argwhere(bad)
        where = ", ".join(f"{tuple(int(v) for v in i)}: {y[tuple(i)]:.4f} vs {ref32[tuple(i)]:.4f}" for i in idx[:48])
        raise AssertionError(f"{what}: {int(bad.sum())}/{bad.size} outside tolerance, max err {np.abs(y - ref32).max()}; at {where}")
