"""CPU model of csrc/gemv_batch.hip's index algebra (tools/emulate_gemv_batch.py): DMA lane map + global-side swizzle, the LDS image
of a piece, fragment reads, nibble order against the pair-permuted activations, the K split over the waves of a block, passes,
partial-tile reduction and store map -- against x @ dequant(W)^T.  No GPU; the kernel itself is pinned by tests/test_gpu_parity.py."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("emulate_gemv_batch", os.path.join(ROOT, "tools", "emulate_gemv_batch.py"))
emu = importlib.util.module_from_spec(spec)
spec.loader.exec_module(emu)


@pytest.mark.parametrize("M,K,N,form,cap", [(5, 512, 40, 0, 0), (8, 1024, 72, 1, 0), (16, 1280, 33, 2, 0), (20, 768, 48, 0, 0), (32, 1152, 100, 0, 0),
                                            (7, 2432, 24, 2, 0), (9, 4224, 20, 1, 0), (6, 1024, 200, 0, 1), (17, 4224, 88, 0, 2),
                                            (12, 8320, 56, 1, 1)])
def test_model_matches_dequant_matmul(M, K, N, form, cap):
    """form: 0 = the launcher's choice, 1 = activations through the swizzled LDS staging area, 2 = direct fragment loads"""
    x, qw, qz, sc = emu.random_case(M, K, N, seed=M + K + N)
    y = emu.run(x, qw, qz, sc, form=form, blocks_cap=cap).astype(np.float32)
    ref = emu.reference(x, qw, qz, sc)
    assert np.abs(y - ref).max() <= 2e-3 * np.abs(ref).max() + 1e-3


def test_model_one_hot_rows_select_weights():
    """an activation row that is one-hot at k returns column k of the dequantised weights exactly (every nibble position)"""
    M, K, N = 16, 256, 32
    _, qw, qz, sc = emu.random_case(M, K, N, seed=3)
    for base in range(0, 16, 8):
        x = np.zeros((M, K), dtype=np.float16)
        ks = [(37 * m + base) % K for m in range(M)]
        for m, k in enumerate(ks):
            x[m, k] = 1.0
        y = emu.run(x, qw, qz, sc)
        ref = emu.reference(x, qw, qz, sc).astype(np.float16)
        assert np.array_equal(y.view(np.uint16), ref.view(np.uint16))


def test_prefill_attention_model():
    """tools/emulate_prefill_attn.py: the numpy model of one block of csrc/prefill_attn.hip in the form the product builds (row sums
    from the matrix pipe) reproduces fp64 attention on ragged / chunked / soft-capped / ALiBi cases, NaN-poisoned rows past the context."""
    import subprocess
    import sys

    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "emulate_prefill_attn.py"), "--mfma-rowsum"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-800:] + r.stderr[-800:]
