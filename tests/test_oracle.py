"""Pins the CPU oracle (oracle/) against the reference-generated golden vectors (tests/golden/).

CPU only.  The fixtures were produced by tests/golden/make_golden.py running the reference's own
dequantize_gemm / WQLinear_GEMM naive forward / from_linear packers.
"""
import numpy as np
import pytest

from conftest import assert_product_close, golden

FULLRANGE = ["fullrange_K256_N64_g128", "fullrange_K128_N32_g32"]
PACKED = ["packed_K512_N64_g128", "packed_K256_N32_g64", "packed_K128_N32_g32"]


def test_fp16_conversion_exhaustive(oracle):
    """software f2h/h2f in the C oracle == numpy's IEEE fp16 for every fp16 bit pattern and for
    fp32 values around every rounding boundary."""
    lib = oracle.lib()
    bits = np.arange(65536, dtype=np.uint16)
    vals = bits.view(np.float16).astype(np.float32)
    for b in range(0, 65536, 7):
        f = lib.awq_oracle_h2f(int(b))
        if np.isnan(vals[b]):
            assert np.isnan(f)
        else:
            assert f == vals[b]
    rng = np.random.default_rng(0)
    probe = np.concatenate([
        rng.standard_normal(4000).astype(np.float32) * 10.0 ** rng.integers(-9, 5, 4000),
        np.array([0.0, -0.0, 65504, 65519.9, 65520, 1e9, -1e9, 2.0 ** -24, 2.0 ** -25, 2.0 ** -25 * 1.0001,
                  2.0 ** -14, 2.0 ** -14 * 0.9999, 6.1e-5, np.inf, -np.inf], np.float32),
        (vals[:31744:3].astype(np.float64) * (1 + 2.0 ** -11)).astype(np.float32),  # exact ties
    ])
    with np.errstate(over="ignore"):
        want = probe.astype(np.float16).view(np.uint16)
    for v, w in zip(probe, want):
        assert lib.awq_oracle_f2h(float(v)) == int(w), (v, hex(int(w)))


def test_kat_a6(oracle):
    """SURVEY.md Appendix A.6 known-answer vector (computed with the reference oracle)."""
    g = golden("kat_a6")
    assert g["qweight"].tolist() == [[1985229328], [-19088744]]
    w = oracle.unpack_gemm(g["qweight"])
    assert w.tolist() == [[0, 4, 1, 5, 2, 6, 3, 7], [8, 12, 9, 13, 10, 14, 11, 15]]
    assert np.array_equal(w, g["w_int"])
    W = oracle.dequant_gemm(g["qweight"], g["qzeros"], g["scales"], 2)
    assert W.tolist() == [[-0.5, 3, 0, 1, 1, 5, 2, -6], [3.5, 11, 16, 3, 9, 13, 10, -14]]
    assert np.array_equal(W.view(np.uint16), g["W"].view(np.uint16))
    _, y = oracle.linear_gemm(g["x"], g["qweight"], g["qzeros"], g["scales"], 2)
    assert y.tolist() == [[6.5, 25, 32, 7, 19, 31, 22, -34]]
    assert np.array_equal(y, g["y"])


@pytest.mark.parametrize("name", FULLRANGE + PACKED)
def test_unpack_and_dequant_bit_exact(oracle, name):
    g = golden(name)
    pre = "gemm_" if name.startswith("packed") else ""
    qw, qz, s = g[pre + "qweight"], g[pre + "qzeros"], g[pre + "scales"]
    gs = int(g["group_size"])
    assert np.array_equal(oracle.unpack_gemm(qw), g["w_int"])
    assert np.array_equal(oracle.unpack_gemm(qz), g["z_int"])
    W = oracle.dequant_gemm(qw, qz, s, gs)
    assert np.array_equal(W.view(np.uint16), g["W"].view(np.uint16)), "dequantize_gemm must be bit exact"
    Wt = oracle.torch_dequantize_gemm(*[__import__("torch").from_numpy(a) for a in (qw, qz, s)], gs)
    assert np.array_equal(Wt.numpy().view(np.uint16), g["W"].view(np.uint16))


@pytest.mark.parametrize("name", FULLRANGE)
def test_product_vs_reference_forward(oracle, name):
    g = golden(name)
    gs = int(g["group_size"])
    y32, y16 = oracle.linear_gemm(g["x"], g["qweight"], g["qzeros"], g["scales"], gs, g["bias"])
    assert_product_close(g["y"], y32, "reference forward vs oracle fp32")
    y32n, _ = oracle.linear_gemm(g["x"], g["qweight"], g["qzeros"], g["scales"], gs)
    assert_product_close(g["y_nobias"], y32n, "reference forward (no bias) vs oracle fp32")
    # the rounded oracle output is within 1 fp16 ulp (+ bias double rounding) of the reference's
    d = np.abs(y16.astype(np.float32) - g["y"].astype(np.float32))
    ulp = np.maximum(np.abs(y32), 2.0 ** -14) * 2.0 ** -10
    assert (d <= 2 * ulp + 1e-6).all()


@pytest.mark.parametrize("name", PACKED)
def test_gemv_layout_restatement(oracle, name):
    """Appendix A.3: reference-packed GEMV buffers dequantise to exactly the GEMM-layout W."""
    g = golden(name)
    gs = int(g["group_size"])
    K = g["w_int"].shape[0]
    zw = oracle.zeros_width(K, gs)
    assert g["gemv_qzeros"].shape[1] == zw and g["gemv_scales"].shape[1] == 8 * zw
    W = oracle.dequant_gemv(g["gemv_qweight"], g["gemv_qzeros"], g["gemv_scales"], gs)
    assert np.array_equal(W.view(np.uint16), g["W"].view(np.uint16))


@pytest.mark.parametrize("name", PACKED)
def test_gemvfast_layout_restatement(oracle, name):
    """Appendix A.4: closed-form index map == pack_intweight; w*s+qzeros dequant close to (w-z)*s."""
    g = golden(name)
    gs = int(g["group_size"])
    assert np.array_equal(oracle.unpack_gemvfast(g["fast_qweight"]), g["w_int"])
    W = oracle.dequant_gemvfast(g["fast_qweight"], g["fast_scales"], g["fast_qzeros"], gs).astype(np.float32)
    ref = g["W"].astype(np.float32)
    smax = np.abs(g["gemm_scales"].astype(np.float32)).max()
    # -(s*z) is rounded to fp16 once when stored, the sum once more: <= 2 half-ulps of |s|*15
    assert np.abs(W - ref).max() <= 15 * smax * 2.0 ** -10


def test_zeros_width_table(oracle):
    """awq/modules/linear/gemv.py:12-24 values quoted in SURVEY.md A.3."""
    for K, g_, want in [(4096, 128, 4), (11008, 128, 11), (14336, 128, 14), (8192, 128, 8), (28672, 128, 28),
                        (4096, 64, 8), (4096, 32, 16), (256, 64, 2), (128, 32, 4)]:
        assert oracle.zeros_width(K, g_) == want


def test_moe_align_docstring_example(oracle):
    """Worked example of awq/modules/fused/moe.py:111-119 (expert ids shifted to 0-based)."""
    topk = np.array([[1, 2, 3], [0, 1, 3], [0, 2, 3], [0, 1, 2]], np.int32)
    sorted_ids, expert_ids, n = oracle.moe_align(topk, 4, 4)
    assert n == 16
    assert sorted_ids[:16].tolist() == [3, 6, 9, 12, 0, 4, 10, 12, 1, 7, 11, 12, 2, 5, 8, 12]
    assert expert_ids[:4].tolist() == [0, 1, 2, 3]
    assert (sorted_ids[16:] == 12).all()


def test_silu_and_mul(oracle):
    rng = np.random.default_rng(1)
    gu = rng.standard_normal((5, 2, 64)).astype(np.float16)
    out = oracle.silu_and_mul(gu)
    g32, u32 = gu[..., :32].astype(np.float32), gu[..., 32:].astype(np.float32)
    want = (g32 / (1 + np.exp(-g32)) * u32).astype(np.float16)
    assert np.abs(out.astype(np.float32) - want.astype(np.float32)).max() <= 2e-3


def test_exact_product_and_weight_rounding_sigma(oracle):
    """linear_gemm_exact (no fp16 rounding of W) and the fp16-W oracle differ by the reference's own
    weight-rounding noise, bounded by weight_rounding_sigma (used by the GPU parity tests)."""
    g = golden("fullrange_K256_N64_g128")
    y32, _ = oracle.linear_gemm(g["x"], g["qweight"], g["qzeros"], g["scales"], 128, g["bias"])
    yex = oracle.linear_gemm_exact(g["x"], g["qweight"], g["qzeros"], g["scales"], 128, g["bias"])
    sig = oracle.weight_rounding_sigma(g["x"], g["W"])
    assert yex.shape == y32.shape and sig.shape == y32.shape
    d = np.abs(yex - y32.astype(np.float64))
    assert (d <= 6 * sig + 1e-6 * np.abs(yex) + 1e-9).all()
    assert d.max() > 0  # the two oracles are genuinely different arithmetic
    # with power-of-two scales and small integers W is exactly representable: both oracles agree
    K, N = 128, 16
    rng = np.random.default_rng(0)
    qw = rng.integers(-2**31, 2**31 - 1, (K, N // 8), dtype=np.int64).astype(np.int32)
    qz = rng.integers(-2**31, 2**31 - 1, (1, N // 8), dtype=np.int64).astype(np.int32)
    s = np.full((1, N), 0.5, np.float16)
    x = rng.integers(-4, 5, (2, K)).astype(np.float16)
    a, _ = oracle.linear_gemm(x, qw, qz, s, 128)
    b = oracle.linear_gemm_exact(x, qw, qz, s, 128)
    assert np.array_equal(a.astype(np.float64), b)


@pytest.mark.parametrize("name", PACKED)
def test_exact_products_of_the_gemv_and_gemvfast_layouts(oracle, name):
    """matmul_exact_gemv on reference-packed GEMV buffers == linear_gemm_exact on the GEMM-layout buffers of the SAME integers
    (bit for bit: the same float64 arithmetic on the same integers and scales); matmul_exact_gemvfast == x @ (w s + qzeros) from
    the fixture's integer weights, and within the weight-rounding noise of the fp16-weight product (these
    are the references the default-path GPU kernels are held to at 1 ulp + 1e-4 rms)."""
    g = golden(name)
    gs = int(g["group_size"])
    a = oracle.linear_gemm_exact(g["x"], g["gemm_qweight"], g["gemm_qzeros"], g["gemm_scales"], gs)
    b = oracle.matmul_exact_gemv(g["x"], g["gemv_qweight"], g["gemv_qzeros"], g["gemv_scales"], gs)
    assert np.allclose(a, b, rtol=1e-12, atol=1e-12), np.abs(a - b).max()
    G = g["w_int"].shape[0] // gs
    Wf = g["w_int"].astype(np.float64) * np.repeat(g["fast_scales"][:G].astype(np.float64), gs, axis=0) \
        + np.repeat(g["fast_qzeros"][:G].astype(np.float64), gs, axis=0)
    c = oracle.matmul_exact_gemvfast(g["x"], g["fast_qweight"], g["fast_scales"], g["fast_qzeros"], gs)
    assert np.allclose(c, g["x"].astype(np.float64) @ Wf, rtol=1e-12, atol=1e-12)
    W16 = oracle.dequant_gemvfast(g["fast_qweight"], g["fast_scales"], g["fast_qzeros"], gs)
    y32, _ = oracle.matmul(g["x"], W16)
    assert (np.abs(c - y32) <= 6 * oracle.weight_rounding_sigma(g["x"], W16) + 1e-6 * np.abs(c) + 1e-9).all()
