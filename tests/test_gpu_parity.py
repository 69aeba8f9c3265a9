"""Parity of the gfx950 HIP path against the CPU oracle and the reference-generated golden
vectors.  Everything here goes through the C ABI (autoawq_amd.ops -> libawq_hip.so).

Bars: integer unpack and the dequantised fp16 weight are BIT-EXACT; the product is
within |y - ref| <= 1e-3*|ref| + 1e-3*rms(ref) of the exact (double-accumulated) product of the
oracle's fp16 weights, plus one fp16 ulp for the output rounding.
"""
import numpy as np
import pytest
import torch

from conftest import assert_close_to_exact, assert_product_close, golden

pytestmark = pytest.mark.gpu

MAX_INT32 = 0x7FFFFFFF
MIN_INT32 = -MAX_INT32 - 1


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from autoawq_amd import _lib, ops as _ops

    _lib.lib()  # fails loudly if the extension is missing
    return _ops


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def fullrange_case(K, N, g, M, seed, realistic=False):
    """tests/test_dequantization.py:15-38 recipe (full-range int32 words, randn scales); the
    `realistic` variant uses small positive scales like quantised checkpoints."""
    gen = torch.Generator().manual_seed(seed)
    qweight = torch.randint(MIN_INT32, MAX_INT32, (K, N // 8), dtype=torch.int32, generator=gen)
    qzeros = torch.randint(MIN_INT32, MAX_INT32, (K // g, N // 8), dtype=torch.int32, generator=gen)
    if realistic:
        scales = (torch.rand((K // g, N), generator=gen) * 0.02 + 0.005).half()
    else:
        scales = torch.randn((K // g, N), generator=gen).half()
    x = torch.randn((M, K), generator=gen).half()
    bias = torch.randn((N,), generator=gen).half()
    return qweight, qzeros, scales, x, bias


# ------------------------------------------------------------------ integer unpack: bit exact

@pytest.mark.parametrize("name", ["kat_a6", "fullrange_K256_N64_g128", "fullrange_K128_N32_g32"])
def test_unpack_golden_bit_exact(ops, name):
    g = golden(name)
    assert np.array_equal(ops.unpack_int4(dev(g["qweight"])).cpu().numpy(), g["w_int"])
    assert np.array_equal(ops.unpack_int4(dev(g["qzeros"])).cpu().numpy(), g["z_int"])


def test_unpack_every_nibble_position_and_sign_bit(ops, oracle):
    rows = []
    for pos in range(8):
        for v in range(16):
            rows.append([v << (4 * pos), 0xFFFFFFFF ^ (v << (4 * pos))])
    q = np.array(rows, dtype=np.uint32).view(np.int32)
    assert np.array_equal(ops.unpack_int4(dev(q)).cpu().numpy(), oracle.unpack_gemm(q))
    big = torch.randint(MIN_INT32, MAX_INT32, (4096, 512), dtype=torch.int32, generator=torch.Generator().manual_seed(7))
    assert np.array_equal(ops.unpack_int4(big.cuda()).cpu().numpy(), oracle.unpack_gemm(big.numpy()))
    assert ops.unpack_int4(torch.empty((0, 4), dtype=torch.int32, device="cuda")).shape == (0, 32)


# ------------------------------------------------------------------ dequant: bit exact

@pytest.mark.parametrize("name", ["kat_a6", "fullrange_K256_N64_g128", "fullrange_K128_N32_g32",
                                  "packed_K512_N64_g128", "packed_K256_N32_g64", "packed_K128_N32_g32"])
def test_dequant_golden_bit_exact(ops, name):
    g = golden(name)
    pre = "gemm_" if name.startswith("packed") else ""
    W = ops.dequantize_weights(dev(g[pre + "qweight"]), dev(g[pre + "scales"]), dev(g[pre + "qzeros"]))
    assert np.array_equal(W.cpu().numpy().view(np.uint16), g["W"].view(np.uint16))


@pytest.mark.parametrize("K,N,g", [(4096, 1792, 128),   # the reference's own test shape
                                   (4096, 4096, 128), (11008, 4096, 128), (4096, 11008, 128),
                                   (512, 64, 32), (512, 72, 64), (768, 40, 768)])
def test_dequant_bit_exact_vs_oracle(ops, oracle, K, N, g):
    qw, qz, s, _, _ = fullrange_case(K, N, g, 1, seed=K + N)
    # sprinkle special scales: tiny (subnormal products), huge (overflow to inf), zero, negative zero
    s.view(-1)[:8] = torch.tensor([6.1e-5, -6.1e-5, 65504, -65504, 1e-7, 0.0, -0.0, 3.0e-4]).half()
    W = ops.dequantize_weights(qw.cuda(), s.cuda(), qz.cuda()).cpu().numpy()
    ref = oracle.dequant_gemm(qw.numpy(), qz.numpy(), s.numpy(), g)
    assert np.array_equal(W.view(np.uint16), ref.view(np.uint16))


# ------------------------------------------------------------------ product

def gemm_variants(ops):
    v = {"auto": 0, "naive": ops.gemm_flags(ops.KERNEL_NAIVE)}
    for nlog in (2, 3, 4):
        v[f"valu_n{nlog}"] = ops.gemm_flags(ops.KERNEL_VALU, nlog=nlog)
        v[f"valu_n{nlog}_s1"] = ops.gemm_flags(ops.KERNEL_VALU, nlog=nlog, splitk=1)
        v[f"valu_n{nlog}_s7"] = ops.gemm_flags(ops.KERNEL_VALU, nlog=nlog, splitk=7)
    v["valu_n3_s64"] = ops.gemm_flags(ops.KERNEL_VALU, nlog=3, splitk=64)
    v["valu_n3_plain_loads"] = ops.gemm_flags(ops.KERNEL_VALU, nlog=3, no_nt=True)
    for wpl in (2, 4):
        v[f"mfma_w{wpl}"] = ops.gemm_flags(ops.KERNEL_MFMA_GEMV, nlog=wpl)
        v[f"mfma_w{wpl}_s1"] = ops.gemm_flags(ops.KERNEL_MFMA_GEMV, nlog=wpl, splitk=1)
        v[f"mfma_w{wpl}_s5_u2"] = ops.gemm_flags(ops.KERNEL_MFMA_GEMV, nlog=wpl, splitk=5, unit=2)
        v[f"mfma_w{wpl}_s3_two_pass"] = ops.gemm_flags(ops.KERNEL_MFMA_GEMV, nlog=wpl, splitk=3, two_pass=True)
        v[f"mfma_w{wpl}_v2_u4"] = ops.gemm_flags(ops.KERNEL_MFMA_GEMV, nlog=wpl, waves=2, unit=4)
    v["mfma_w2_v8_u8"] = ops.gemm_flags(ops.KERNEL_MFMA_GEMV, nlog=2, waves=8, unit=8)
    v["mfma_w2_s8_u2"] = ops.gemm_flags(ops.KERNEL_MFMA_GEMV, nlog=2, splitk=8, unit=2)  # the wide (gate|up) tuning branch, forced
    v["mfma_w2_s16_v8"] = ops.gemm_flags(ops.KERNEL_MFMA_GEMV, nlog=2, splitk=16, waves=8)  # the narrow-matrix default, forced
    v["mfma_w2_v4_u8_s64"] = ops.gemm_flags(ops.KERNEL_MFMA_GEMV, nlog=2, waves=4, unit=8, splitk=64)
    return v


@pytest.mark.parametrize("name", ["kat_a6", "fullrange_K256_N64_g128", "fullrange_K128_N32_g32"])
def test_gemm_golden(ops, oracle, name):
    """reference forward outputs (tests/golden) vs the HIP path, all kernel variants that take the shape"""
    g = golden(name)
    gs = int(g["group_size"]) if "group_size" in g else 2
    bias = g.get("bias")
    y32, _ = oracle.linear_gemm(g["x"], g["qweight"], g["qzeros"], g["scales"], gs, bias)
    want_ref = g["y"].astype(np.float32)
    for vname, flags in gemm_variants(ops).items():
        try:
            y = ops.gemm_forward(dev(g["x"]), dev(g["qweight"]), dev(g["scales"]), dev(g["qzeros"]),
                                 dev(bias) if bias is not None else None, flags=flags)
        except Exception as e:  # variant does not take this shape (e.g. N % 32 != 0): must say so
            assert "no kernel" in str(e), (vname, e)
            continue
        y = y.cpu().numpy().astype(np.float32)
        assert_product_close(y, y32, f"{name}/{vname} vs oracle")
        # and against what the reference itself produced on CPU (fp16, so 2 ulp of slack)
        ulp = np.maximum(np.abs(y32), 2.0 ** -14) * 2.0 ** -10
        rms = np.sqrt((y32.astype(np.float64) ** 2).mean())
        assert (np.abs(y - want_ref) <= 3 * ulp + 1e-3 * np.abs(y32) + 1e-3 * rms).all(), vname


@pytest.mark.parametrize("K,N,g", [(4096, 4096, 128), (4096, 11008, 128), (11008, 4096, 128),
                                   (4096, 12288, 128), (4096, 22016, 128),  # every Linear bench.py times, at every M
                                   (1024, 8192, 128), (512, 96, 64), (512, 1056, 32),
                                   (256, 40, 128), (2048, 2048, 2048)])
@pytest.mark.parametrize("M", [1, 2, 3, 4, 5, 8, 13, 16])
def test_gemm_vs_oracle_all_variants(ops, oracle, K, N, g, M):
    qw, qz, s, x, bias = fullrange_case(K, N, g, M, seed=K * 3 + N + M, realistic=(N % 64 == 0))
    y32, _ = oracle.linear_gemm(x.numpy(), qw.numpy(), qz.numpy(), s.numpy(), g, bias.numpy())
    W = oracle.dequant_gemm(qw.numpy(), qz.numpy(), s.numpy(), g)
    wsig = oracle.weight_rounding_sigma(x.numpy(), W)
    yex = oracle.linear_gemm_exact(x.numpy(), qw.numpy(), qz.numpy(), s.numpy(), g, bias.numpy())
    rms = float(np.sqrt(np.mean(yex ** 2)))
    dq, dz, ds, dx, db = qw.cuda(), qz.cuda(), s.cuda(), x.cuda(), bias.cuda()
    ran = 0
    for vname, flags in gemm_variants(ops).items():
        try:
            y = ops.gemm_forward(dx, dq, ds, dz, db, flags=flags)
        except Exception as e:
            assert "no kernel" in str(e), (vname, e)
            continue
        ran += 1
        yn = y.cpu().numpy().astype(np.float64)
        if ops.last_kernel() in ("naive", "gemv_valu", "gemm_skinny"):  # reference-order numerics: fp16-rounded weights
            assert_product_close(yn, y32, f"K{K} N{N} g{g} M{M} {vname}")
        else:
            # MFMA kernels: exact integer dot product, scale applied in fp32 -> within the
            # reference's weight-rounding noise of the oracle, and within 1 ulp + 1e-4*rms of
            # the exact-arithmetic product
            assert_product_close(yn, y32, f"K{K} N{N} g{g} M{M} {vname}", wsigma=wsig)
            ulp = np.maximum(np.abs(yex), 2.0 ** -14) * 2.0 ** -10
            bad = np.abs(yn - yex) > ulp + 1e-4 * rms + 1e-5 * np.abs(yex)
            assert not bad.any(), f"K{K} N{N} g{g} M{M} {vname}: {int(bad.sum())} off the exact product"
    assert ran >= 2
    assert ops.workspace_is_clean(dx.device), "split-K tickets must be re-armed (zero) after every call"


@pytest.mark.parametrize("K,N,g", [(512, 256, 128), (1024, 384, 64), (256, 136, 32), (4096, 512, 128), (2048, 2048, 2048)])
@pytest.mark.parametrize("M", [17, 64, 100, 128, 300])
@pytest.mark.parametrize("bn", [1, 2])
def test_tiled_gemm_vs_oracle(ops, oracle, K, N, g, M, bn):
    """fused dequant + MFMA GEMM (M > 16): reference-order numerics (fp16-rounded weights, fp32
    accumulation), ragged M and N tiles, with bias; also the asymmetric one-hot check that would
    expose a transposed fragment."""
    qw, qz, s, x, bias = fullrange_case(K, N, g, M, seed=K + N + M, realistic=(N % 64 == 0))
    y32, _ = oracle.linear_gemm(x.numpy(), qw.numpy(), qz.numpy(), s.numpy(), g, bias.numpy())
    fl = ops.gemm_flags(ops.KERNEL_TILED, nlog=bn)
    for sk in (0, 1, 3):  # auto split-K (in-launch exchange), none, odd
        y = ops.gemm_forward(x.cuda(), qw.cuda(), s.cuda(), qz.cuda(), bias.cuda(), flags=ops.gemm_flags(ops.KERNEL_TILED, nlog=bn, splitk=sk))
        assert ops.last_kernel() == "gemm_tiled"
        assert_product_close(y.cpu().numpy().astype(np.float64), y32, f"tiled K{K} N{N} g{g} M{M} bn{bn} s{sk}")
    assert ops.workspace_is_clean(y.device)
    assert torch.equal(y, ops.gemm_forward(x.cuda(), qw.cuda(), s.cuda(), qz.cuda(), bias.cuda(), flags=ops.gemm_flags(ops.KERNEL_TILED, nlog=bn, splitk=3)))
    # one-hot rows pick rows of the bit-exact dequantised W (row m selects k = 7*m + 3)
    W = ops.dequantize_weights(qw.cuda(), s.cuda(), qz.cuda())
    e = torch.zeros((M, K), dtype=torch.float16, device="cuda")
    ks = (torch.arange(M, device="cuda") * 7 + 3) % K
    e[torch.arange(M, device="cuda"), ks] = 1.0
    out = ops.gemm_forward(e, qw.cuda(), s.cuda(), qz.cuda(), flags=fl)
    diff = torch.nonzero(out != W[ks])
    assert diff.numel() == 0, f"one-hot rows differ at {diff[:48].tolist()} ({diff.shape[0]} elements)"


@pytest.mark.parametrize("K,N,g", [(512, 256, 128), (1024, 384, 64), (256, 136, 64), (4096, 512, 128), (2048, 2048, 2048), (192, 264, 192)])
@pytest.mark.parametrize("M", [17, 128, 300, 1000])
@pytest.mark.parametrize("bm", [1, 2])
def test_regb_gemm_vs_oracle(ops, oracle, K, N, g, M, bm):
    """Prefill GEMM with the weight operand decoded in registers (csrc/gemm_regb.hip): the same numerics as the
    reference's dequantise-then-matmul (exact fp16 weights, fp32 accumulation) -- against the oracle with bias, ragged
    M and N tiles, groups of 64 / 128 / 192 / K rows; row-exact against the bit-exact dequantised W with one-hot
    activations (which would expose a wrong K slot, byte selector, swizzle or column order), and bitwise repeatable."""
    qw, qz, s, x, bias = fullrange_case(K, N, g, M, seed=K + N + M, realistic=(N % 64 == 0))
    y32, _ = oracle.linear_gemm(x.numpy(), qw.numpy(), qz.numpy(), s.numpy(), g, bias.numpy())
    fl = ops.gemm_flags(ops.KERNEL_REGB, nlog=bm)
    y = ops.gemm_forward(x.cuda(), qw.cuda(), s.cuda(), qz.cuda(), bias.cuda(), flags=fl)
    assert ops.last_kernel() == "gemm_regb"
    assert_product_close(y.cpu().numpy().astype(np.float64), y32, f"regb K{K} N{N} g{g} M{M} bm{bm}")
    assert torch.equal(y, ops.gemm_forward(x.cuda(), qw.cuda(), s.cuda(), qz.cuda(), bias.cuda(), flags=fl))
    W = ops.dequantize_weights(qw.cuda(), s.cuda(), qz.cuda())
    e = torch.zeros((M, K), dtype=torch.float16, device="cuda")
    ks = (torch.arange(M, device="cuda") * 7 + 3) % K
    e[torch.arange(M, device="cuda"), ks] = 1.0
    out = ops.gemm_forward(e, qw.cuda(), s.cuda(), qz.cuda(), flags=fl)
    diff = torch.nonzero(out != W[ks])
    assert diff.numel() == 0, f"one-hot rows differ at {diff[:48].tolist()} ({diff.shape[0]} elements)"


@pytest.mark.parametrize("K,N,g", [(512, 256, 128), (1024, 384, 64), (256, 136, 64), (4096, 512, 128), (2048, 2048, 2048), (192, 264, 192), (4096, 4096, 128)])
@pytest.mark.parametrize("M", [9, 13, 16, 17, 32, 33, 50, 64])
def test_skinny_gemm_vs_oracle(ops, oracle, K, N, g, M):
    """Batched-decode GEMM (csrc/gemm_skinny.hip, 17 <= M <= 64): exact fp16 weights, fp32 accumulation -- against the
    oracle with bias for the auto split and forced K splits (one slice, as many as the reducers, more), ragged M / N tiles,
    groups of 64 / 128 / 192 / K rows; row-exact against the bit-exact dequantised W with one-hot activations; bitwise
    repeatable; exchange region back in its initial state."""
    qw, qz, s, x, bias = fullrange_case(K, N, g, M, seed=K + N + M, realistic=(N % 64 == 0))
    y32, _ = oracle.linear_gemm(x.numpy(), qw.numpy(), qz.numpy(), s.numpy(), g, bias.numpy())
    ran = 0
    for sk in (0, 1, 2, 4, 5, 8):
        fl = ops.gemm_flags(ops.KERNEL_SKINNY, splitk=sk)
        try:
            y = ops.gemm_forward(x.cuda(), qw.cuda(), s.cuda(), qz.cuda(), bias.cuda(), flags=fl)
        except Exception as e:  # a split the shape cannot take (fewer 64-row steps than slices, fewer slices than reducers)
            assert "no kernel" in str(e) or "unsupported" in str(e).lower(), e
            continue
        ran += 1
        assert ops.last_kernel() == "gemm_skinny"
        assert_product_close(y.cpu().numpy().astype(np.float64), y32, f"skinny K{K} N{N} g{g} M{M} s{sk}")
        assert torch.equal(y, ops.gemm_forward(x.cuda(), qw.cuda(), s.cuda(), qz.cuda(), bias.cuda(), flags=fl))
    assert ran >= 2
    assert ops.workspace_is_clean(y.device)
    W = ops.dequantize_weights(qw.cuda(), s.cuda(), qz.cuda())
    e = torch.zeros((M, K), dtype=torch.float16, device="cuda")
    ks = (torch.arange(M, device="cuda") * 7 + 3) % K
    e[torch.arange(M, device="cuda"), ks] = 1.0
    out = ops.gemm_forward(e, qw.cuda(), s.cuda(), qz.cuda(), flags=ops.gemm_flags(ops.KERNEL_SKINNY))
    diff = torch.nonzero(out != W[ks])
    assert diff.numel() == 0, f"one-hot rows differ at {diff[:48].tolist()} ({diff.shape[0]} elements)"


def test_regb_gemm_refuses_what_it_cannot_run(ops):
    """K not a multiple of 64 / groups of 32 rows: the launcher says UNSUPPORTED (the caller's fallback is gemm_tiled)."""
    from autoawq_amd import _lib

    for K, N, g in [(160, 64, 32), (96, 64, 32)]:
        qw, qz, s, x, _ = fullrange_case(K, N, g, 32, seed=3, realistic=True)
        with pytest.raises(_lib.AwqHipError):
            ops.gemm_forward(x.cuda(), qw.cuda(), s.cuda(), qz.cuda(), flags=ops.gemm_flags(ops.KERNEL_REGB))


def test_tiled_splitk_more_blocks_than_cus(ops):
    """M = 300 at 2048 x 2048 with a workspace large enough for S = 8: 48 tiles x 8 slices = 384
    blocks, every producer wave streams 16 chunks of 16 bytes into the exchange.  This is the
    configuration in which a rewritten store-data register (profiles/r01_store_hazard.txt) showed;
    one-hot rows make every slice's contribution individually visible."""
    K = N = 2048
    M = 300
    qw, qz, s, x, _ = fullrange_case(K, N, K, M, seed=77, realistic=True)
    dq, ds, dz = qw.cuda(), s.cuda(), qz.cuda()
    ops.workspace(dq.device, 16384 + (64 << 20))
    W = ops.dequantize_weights(dq, ds, dz)
    ks = (torch.arange(M, device="cuda") * 7 + 3) % K
    e = torch.zeros((M, K), dtype=torch.float16, device="cuda")
    e[torch.arange(M, device="cuda"), ks] = 1.0
    for bn in (1, 2):
        for _ in range(5):
            out = ops.gemm_forward(e, dq, ds, dz, flags=ops.gemm_flags(ops.KERNEL_TILED, nlog=bn))
            diff = torch.nonzero(out != W[ks])
            assert diff.numel() == 0, f"bn{bn}: one-hot rows differ at {diff[:32].tolist()}"
        one = ops.gemm_forward(x.cuda(), dq, ds, dz, flags=ops.gemm_flags(ops.KERNEL_TILED, nlog=bn, splitk=1))
        for _ in range(5):
            y = ops.gemm_forward(x.cuda(), dq, ds, dz, flags=ops.gemm_flags(ops.KERNEL_TILED, nlog=bn))
            assert (y.float() - one.float()).abs().max() <= 2e-3 * one.float().abs().max()
    assert ops.workspace_is_clean(dq.device)


@pytest.mark.parametrize("M", [2048, 4300])
def test_tiled_large_grid_k32_tile(ops, M):
    """Grids of >= 512 tiles take the K-step-32 instantiations -- 128 x 256 on four 64 x 128 waves
    (M = 2048 here) or, when 256-row tiles still fill the chip, 256 x 256 on eight (M = 4300: ragged last
    tile): same numerics as every other tile -- exact fp16 weights, fp32 accumulation -- checked against
    the dequantised-weights product and, row-exact, with one-hot activations."""
    K, N = 384, 8192
    qw, qz, s, x, bias = fullrange_case(K, N, 128, M, seed=5, realistic=True)
    dq, ds, dz = qw.cuda(), s.cuda(), qz.cuda()
    W = ops.dequantize_weights(dq, ds, dz)
    y = ops.gemm_forward(x.cuda(), dq, ds, dz, bias.cuda(), flags=ops.gemm_flags(ops.KERNEL_TILED, nlog=2))
    ref = x.cuda().float() @ W.float() + bias.cuda().float()
    err = (y.float() - ref).abs()
    assert bool((err <= 2e-3 * ref.abs() + 2e-3 * ref.abs().mean()).all()), float(err.max())
    ks = (torch.arange(M, device="cuda") * 5 + 1) % K
    e = torch.zeros((M, K), dtype=torch.float16, device="cuda")
    e[torch.arange(M, device="cuda"), ks] = 1.0
    assert torch.equal(ops.gemm_forward(e, dq, ds, dz, flags=ops.gemm_flags(ops.KERNEL_TILED, nlog=2)), W[ks])


def test_auto_dispatch_by_m(ops):
    qw, qz, s, x, _ = fullrange_case(512, 256, 128, 40, seed=9, realistic=True)
    dq, dz, ds = qw.cuda(), qz.cuda(), s.cuda()
    for M, want in [(1, "gemv_mfma"), (8, "gemv_mfma"), (9, "gemm_skinny"), (16, "gemm_skinny"), (17, "gemm_skinny"), (40, "gemm_skinny")]:
        ops.gemm_forward(x[:M].cuda(), dq, ds, dz)
        assert ops.last_kernel() == want, (M, ops.last_kernel())
        assert ops.auto_kernel(M, 512, 256, 128) == {"gemv_mfma": ops.KERNEL_MFMA_GEMV, "gemm_skinny": ops.KERNEL_SKINNY}[want]
    assert ops.auto_kernel(100, 512, 256, 128) == ops.KERNEL_TILED and ops.auto_kernel(40, 512, 256, 32) == ops.KERNEL_TILED
    assert ops.auto_kernel(16, 11008, 4096, 128) == ops.KERNEL_MFMA_GEMV and ops.auto_kernel(16, 4096, 22016, 128) == ops.KERNEL_MFMA_GEMV
    # prefill sizes whose 128 x 256 tiles give every CU a block go to the register-decoded kernel (host-only query)
    assert ops.auto_kernel(1024, 4096, 11008, 128) == ops.KERNEL_REGB and ops.auto_kernel(512, 4096, 11008, 128) == ops.KERNEL_TILED
    assert ops.auto_kernel(2048, 11008, 4096, 128) == ops.KERNEL_REGB and ops.auto_kernel(1024, 11008, 4096, 128) == ops.KERNEL_TILED
    assert ops.auto_kernel(4096, 4096, 4096, 32) == ops.KERNEL_TILED   # groups of 32 rows: not the register-decoded kernel


def test_gemm_deterministic_and_counters_rearmed(ops):
    """split-K slabs are summed in fixed slab order: bitwise identical across 20 runs, and the
    tickets come back to zero so the next (different-shape) call works."""
    qw, qz, s, x, _ = fullrange_case(4096, 4096, 128, 1, seed=11, realistic=True)
    dq, dz, ds, dx = qw.cuda(), qz.cuda(), s.cuda(), x.cuda()
    for flags in (0, ops.gemm_flags(ops.KERNEL_VALU, nlog=3, splitk=32), ops.gemm_flags(ops.KERNEL_MFMA_GEMV, nlog=2, splitk=16), ops.gemm_flags(ops.KERNEL_MFMA_GEMV, nlog=4, splitk=8, waves=4, unit=2)):
        first = ops.gemm_forward(dx, dq, ds, dz, flags=flags)
        for _ in range(20):
            assert torch.equal(ops.gemm_forward(dx, dq, ds, dz, flags=flags), first)
    assert ops.workspace_is_clean(dx.device), "split-K tickets must be zero after every call"
    qw2, qz2, s2, x2, _ = fullrange_case(1024, 8192, 128, 2, seed=12, realistic=True)
    a = ops.gemm_forward(x2.cuda(), qw2.cuda(), s2.cuda(), qz2.cuda())
    b = ops.gemm_forward(x2.cuda(), qw2.cuda(), s2.cuda(), qz2.cuda(), flags=ops.gemm_flags(ops.KERNEL_NAIVE))
    assert (a.float() - b.float()).abs().max() <= 2e-3 * b.float().abs().max()


def test_gemm_linearity_and_zero_input_at_full_size(ops):
    """size-independent properties on the BASELINE shape 4096x11008: f(0) = 0 exactly,
    f(2x) = 2 f(x) exactly (power-of-two scaling commutes with every rounding), one-hot x picks
    a row of the bit-exact dequantised W."""
    qw, qz, s, x, _ = fullrange_case(4096, 11008, 128, 1, seed=21, realistic=True)
    dq, dz, ds, dx = qw.cuda(), qz.cuda(), s.cuda(), x.cuda()
    assert int(ops.gemm_forward(torch.zeros_like(dx), dq, ds, dz).abs().max()) == 0
    y1 = ops.gemm_forward(dx, dq, ds, dz)
    y2 = ops.gemm_forward(dx * 2, dq, ds, dz)
    assert torch.equal(y2, y1 * 2)
    W = ops.dequantize_weights(dq, ds, dz)
    for k in (0, 1, 127, 128, 4095):
        e = torch.zeros_like(dx)
        e[0, k] = 1.0
        assert torch.equal(ops.gemm_forward(e, dq, ds, dz)[0], W[k])


# ------------------------------------------------------------------ module semantics on device

def test_module_forward_semantics(ops, oracle):
    from autoawq_amd import WQLinear_GEMM

    g = golden("fullrange_K256_N64_g128")
    m = WQLinear_GEMM(4, 128, 256, 64, True, "cuda")
    m.qweight, m.qzeros, m.scales, m.bias = dev(g["qweight"]), dev(g["qzeros"]), dev(g["scales"]), dev(g["bias"])
    y32, _ = oracle.linear_gemm(g["x"], g["qweight"], g["qzeros"], g["scales"], 128, g["bias"])
    x = dev(g["x"])
    out3 = m(x.view(1, 3, 256))
    assert out3.shape == (1, 3, 64) and out3.dtype == torch.float16
    assert_product_close(out3[0].cpu().numpy().astype(np.float32), y32, "3-D fp16")
    out2 = m(x)  # 2-D in -> 2-D out (gemm.py:287)
    assert out2.shape == (3, 64)
    assert torch.equal(out2, out3[0])
    for dt in (torch.float32, torch.bfloat16):  # cast to fp16 and back (gemm.py:256-258,284-285)
        xd = x.to(dt)
        o = m(xd)
        assert o.dtype == dt and torch.equal(o, m(xd.half()).to(dt))
    assert m(torch.empty((0, 5, 256), device="cuda", dtype=torch.float16)).shape == (0, 5, 64)
    m.bias = None
    y32n, _ = oracle.linear_gemm(g["x"], g["qweight"], g["qzeros"], g["scales"], 128)
    assert_product_close(m(x).cpu().numpy().astype(np.float32), y32n, "no bias")


def test_module_prefill_branch_and_backward(ops, oracle):
    """>= 1024 tokens takes the dequant + fp16 GEMM branch (gemm.py:48-54); backward = dequant +
    matmul with W^T (gemm.py:88-114)."""
    from autoawq_amd import WQLinear_GEMM

    qw, qz, s, _, bias = fullrange_case(512, 256, 128, 1, seed=5, realistic=True)
    m = WQLinear_GEMM(4, 128, 512, 256, True, "cuda", training=True)
    m.qweight, m.qzeros, m.scales, m.bias = qw.cuda(), qz.cuda(), s.cuda(), bias.cuda()
    x = torch.randn((2, 640, 512), generator=torch.Generator().manual_seed(1)).half().cuda().requires_grad_(True)
    y = m(x)
    y32, _ = oracle.linear_gemm(x.detach().cpu().numpy().reshape(-1, 512), qw.numpy(), qz.numpy(), s.numpy(), 128,
                                bias.numpy())
    assert_product_close(y.detach().cpu().numpy().astype(np.float32).reshape(-1, 256), y32, "prefill branch")
    y.float().sum().backward()
    W = oracle.dequant_gemm(qw.numpy(), qz.numpy(), s.numpy(), 128).astype(np.float32)
    want = np.broadcast_to(W.sum(axis=1), (2, 640, 512))
    got = x.grad.float().cpu().numpy()
    assert np.abs(got - want).max() <= 2e-2 * np.abs(want).max()


def test_fused_qkv_concatenation(ops):
    """fuse_qkv semantics (awq/utils/fused_utils.py:87-96): concatenating packed buffers along N
    gives exactly the concatenated outputs; pinned against the reference's own fused output."""
    from autoawq_amd import WQLinear_GEMM

    g = golden("fused_qkv_K256_g128")
    x = dev(g["x"])
    outs = []
    for t, n in zip("qkv", (64, 32, 32)):
        m = WQLinear_GEMM(4, 128, 256, n, True, "cuda")
        m.qweight, m.qzeros, m.scales, m.bias = (dev(g[f"{t}_qweight"]), dev(g[f"{t}_qzeros"]),
                                                 dev(g[f"{t}_scales"]), dev(g[f"{t}_bias"]))
        outs.append(m(x))
    f = WQLinear_GEMM(4, 128, 256, 128, True, "cuda")
    f.qweight = torch.cat([dev(g[f"{t}_qweight"]) for t in "qkv"], dim=1)
    f.qzeros = torch.cat([dev(g[f"{t}_qzeros"]) for t in "qkv"], dim=1)
    f.scales = torch.cat([dev(g[f"{t}_scales"]) for t in "qkv"], dim=1)
    f.bias = torch.cat([dev(g[f"{t}_bias"]) for t in "qkv"], dim=0)
    assert np.array_equal(f.qweight.cpu().numpy(), g["fused_qweight"])
    yf = f(x)
    ref = g["y"].astype(np.float32)
    # the reference's fp16 CPU matmul output carries its own rounding: 3 ulp + the stated 1e-3*rms
    tol = 3 * np.maximum(np.abs(ref), 2.0 ** -14) * 2.0 ** -10 + 1e-3 * np.sqrt((ref.astype(np.float64) ** 2).mean())
    assert (np.abs(yf.cpu().numpy().astype(np.float32) - ref) <= tol).all()
    assert (np.abs(torch.cat(outs, -1).cpu().numpy().astype(np.float32) - ref) <= tol).all()
    assert torch.equal(torch.cat(outs, -1), yf), "column-concatenated buffers must give bitwise the concatenated outputs"


def test_awq_ext_shim_positional_api(ops):
    """The reference's call forms (gemm.py:51-58) run unchanged against the shim."""
    from autoawq_amd import awq_ext

    g = golden("fullrange_K256_N64_g128")
    qw, s, qz = dev(g["qweight"]), dev(g["scales"]), dev(g["qzeros"])
    W = awq_ext.dequantize_weights_cuda(qw, s, qz, 0, 0, 0, False)
    assert np.array_equal(W.cpu().numpy().view(np.uint16), g["W"].view(np.uint16))
    x = dev(g["x"]).view(1, 3, 256)
    out = awq_ext.gemm_forward_cuda(x.reshape(-1, x.shape[-1]), qw, s, qz, 8)
    ref = g["y_nobias"].astype(np.float32)
    ulp = np.maximum(np.abs(ref), 2.0 ** -14) * 2.0 ** -10
    rms = np.sqrt((ref.astype(np.float64) ** 2).mean())
    assert (np.abs(out.cpu().numpy().astype(np.float32) - ref) <= 3 * ulp + 1e-3 * np.abs(ref) + 1e-3 * rms).all()


# ------------------------------------------------------------------ GEMV layout (WQLinear_GEMV)

def gemv_case(K, N, g, M, seed):
    """Random GEMV-layout buffers: full-range packed words, zero-padded scales / zero nibbles past
    K/g groups exactly like the reference packer leaves them (gemv.py:97-106,143-152)."""
    from autoawq_amd.utils.packing import calculate_zeros_width

    gen = torch.Generator().manual_seed(seed)
    zw = calculate_zeros_width(K, g)
    G = K // g
    qw = torch.randint(MIN_INT32, MAX_INT32, (N, K // 8), dtype=torch.int32, generator=gen)
    zn = torch.randint(0, 16, (N, zw * 8), dtype=torch.int32, generator=gen)
    zn[:, G:] = 0
    qz = torch.zeros((N, zw), dtype=torch.int32)
    for i in range(8):
        qz |= zn[:, i::8] << (4 * i)
    sc = torch.zeros((N, zw * 8), dtype=torch.float16)
    sc[:, :G] = (torch.rand((N, G), generator=gen) * 0.02 + 0.005).half()
    x = torch.randn((M, K), generator=gen).half()
    return qw, qz, sc, x


@pytest.mark.parametrize("name", ["packed_K512_N64_g128", "packed_K256_N32_g64", "packed_K128_N32_g32"])
def test_gemv_layout_golden(ops, oracle, name):
    """reference-packed GEMV buffers: dequant bit-exact == GEMM-layout W^T; product vs the
    reference's own forward output."""
    g = golden(name)
    gs = int(g["group_size"])
    qw, qz, sc = dev(g["gemv_qweight"]), dev(g["gemv_qzeros"]), dev(g["gemv_scales"])
    Wt = ops.dequantize_weights_gemv(qw, sc, qz, gs)
    assert np.array_equal(Wt.cpu().numpy().view(np.uint16), np.ascontiguousarray(g["W"].T).view(np.uint16))
    y = ops.gemv_forward(dev(g["x"]), qw, sc, qz, gs).cpu().numpy().astype(np.float64)
    y32, _ = oracle.matmul(g["x"], g["W"])
    wsig = oracle.weight_rounding_sigma(g["x"], g["W"])
    assert_product_close(y, y32, f"{name} gemv layout", wsigma=wsig)


GEMV_KERNEL_TILE16, GEMV_KERNEL_ROWS = 1, 2  # AWQ_GEMV_KERNEL_* of include/awq_hip.h


def gemv_rows_takes(M, K, g):
    """AUTO dispatch of awq_gemv_forward (capi.hip awq_gemv_auto_kernel): the row-streaming kernel at batches 1 and 2, and at
    batches 3 .. 4 while K <= 6144 (round 4; profiles/r03_gemv_rows_sweep.txt)"""
    return g % 128 == 0 and K % g == 0 and K >= 128 and (M <= 2 or (M <= 4 and K <= 6144)) and not gemv_batch_takes(M, K, g)


def gemv_batch_takes(M, K, g):
    """AUTO dispatch of awq_gemv_forward from five rows (round 5): csrc/gemv_batch.hip wherever it takes the shape; four rows while
    K > 2048 and three while K > 6144 too (profiles/r05_sweep_small_batch.txt)"""
    return g == 128 and (M >= 5 or (M == 4 and K > 2048) or (M == 3 and K > 6144)) and K % 128 == 0 and K >= 128


@pytest.mark.parametrize("K,N,g", [(4096, 4096, 128), (11008, 4096, 128), (4096, 11008, 128), (1024, 72, 64),
                                   (512, 40, 32), (2048, 200, 2048), (256, 16, 128)])
@pytest.mark.parametrize("M", [1, 2, 5, 8, 16, 33])
def test_gemv_layout_vs_oracle(ops, oracle, K, N, g, M):
    qw, qz, sc, x = gemv_case(K, N, g, M, seed=K + 3 * N + M)
    W = oracle.dequant_gemv(qw.numpy(), qz.numpy(), sc.numpy(), g)      # [K, N] fp16, reference rounding
    Wt = ops.dequantize_weights_gemv(qw.cuda(), sc.cuda(), qz.cuda(), g)
    assert np.array_equal(Wt.cpu().numpy().view(np.uint16), np.ascontiguousarray(W.T).view(np.uint16))
    y32, _ = oracle.matmul(x.numpy(), W)
    wsig = oracle.weight_rounding_sigma(x.numpy(), W)
    tile16 = ops.gemm_flags(kernel=GEMV_KERNEL_TILE16)
    for flags in (0, tile16, tile16 | ops.gemm_flags(waves=4, unit=8), tile16 | ops.gemm_flags(waves=16, unit=4)):
        y = ops.gemv_forward(x.cuda(), qw.cuda(), sc.cuda(), qz.cuda(), g, flags=flags)
        if flags == 0 and (M <= 2 or gemv_batch_takes(M, K, g) or (M <= 16 and K <= 4096)):  # (wider K: the wrapper splits the batch to fit the tile kernel's LDS)
            assert ops.last_kernel() == ("gemv_rows" if gemv_rows_takes(M, K, g) else ("gemv_batch" if gemv_batch_takes(M, K, g) else "gemv_nk"))
        elif flags:
            assert ops.last_kernel() == "gemv_nk"
        assert_product_close(y.cpu().numpy().astype(np.float64), y32, f"gemv K{K} N{N} g{g} M{M} f{flags:x}", wsigma=wsig)
    # bitwise reproducible, one-hot rows select rows of the bit-exact W, zero in -> zero out
    y1 = ops.gemv_forward(x.cuda(), qw.cuda(), sc.cuda(), qz.cuda(), g)
    assert torch.equal(y1, ops.gemv_forward(x.cuda(), qw.cuda(), sc.cuda(), qz.cuda(), g))
    e = torch.zeros((M, K), dtype=torch.float16, device="cuda")
    ks = (torch.arange(M, device="cuda") * 37 + 5) % K
    e[torch.arange(M, device="cuda"), ks] = 1.0
    assert torch.equal(ops.gemv_forward(e, qw.cuda(), sc.cuda(), qz.cuda(), g), Wt.t()[ks])
    assert int(ops.gemv_forward(torch.zeros_like(e), qw.cuda(), sc.cuda(), qz.cuda(), g).abs().max()) == 0


# the row-streaming kernel (csrc/gemv_rows.hip): 7B shapes incl. the fused qkv / gate|up widths bench.py --layout gemv
# times, the 70B TP = 8 shard shapes of BASELINE configs[3] (8192 -> 1280, 1024 -> 8192, 8192 -> 7168, 3584 -> 8192), the
# unsharded 70B shapes (8192 -> 10240; K = 28672: two waves side by side on a row), odd sizes (N not a multiple of the
# rows per super-unit, one-line rows, a group wider than a slot, g = 256)
ROWS_SHAPES = [(4096, 4096, 128), (4096, 12288, 128), (4096, 22016, 128), (11008, 4096, 128), (8192, 1280, 128),
               (1024, 8192, 128), (8192, 7168, 128), (3584, 8192, 128), (8192, 10240, 128), (28672, 1024, 128),
               (13824, 5120, 128), (2048, 200, 2048), (256, 16, 128), (1280, 10, 256), (384, 7, 128), (4096, 4099, 128)]


@pytest.mark.parametrize("K,N,g", ROWS_SHAPES)
def test_gemv_rows_kernel_vs_oracle(ops, oracle, K, N, g):
    """every batch size the kernel takes (1..4), the default configuration and forced ones: slots per wave (whole rows
    or several waves side by side), waves per block, super-units in flight, blocks per CU"""
    small = K * N <= 4096 * 4096
    qw, qz, sc, x4 = gemv_case(K, N, g, 4, seed=K + 5 * N)
    W = oracle.dequant_gemv(qw.numpy(), qz.numpy(), sc.numpy(), g)
    Wt = ops.dequantize_weights_gemv(qw.cuda(), sc.cuda(), qz.cuda(), g)
    assert np.array_equal(Wt.cpu().numpy().view(np.uint16), np.ascontiguousarray(W.T).view(np.uint16))
    qwc, qzc, scc = qw.cuda(), qz.cuda(), sc.cuda()
    rows = ops.gemm_flags(kernel=GEMV_KERNEL_ROWS)
    forced = [0] + [ops.gemm_flags(nlog=sl) for sl in (1, 2, 3, 4, 6, 8)] + [
        ops.gemm_flags(waves=8, unit=1, splitk=1), ops.gemm_flags(waves=4, unit=2, splitk=2), ops.gemm_flags(waves=8, unit=2, splitk=3),
        ops.gemm_flags(waves=2, unit=1, splitk=1)]
    ran = 0
    yex4 = oracle.matmul_exact_gemv(x4.numpy(), qw.numpy(), qz.numpy(), sc.numpy(), g)
    for M in (1, 2, 3, 4) if small else (1, 2):
        x = x4[:M].contiguous()
        y32, _ = oracle.matmul(x.numpy(), W)
        wsig = oracle.weight_rounding_sigma(x.numpy(), W)
        # the default configuration (what every caller runs) also within 1 ulp + 1e-4 rms of the EXACT product: the 6-sigma widening
        # below is never the only bound on the headline kernel
        assert_close_to_exact(ops.gemv_forward(x.cuda(), qwc, scc, qzc, g, flags=rows).cpu().numpy(), yex4[:M], f"rows K{K} N{N} g{g} M{M} vs exact")
        for f in forced if (small or M == 1) else forced[:1]:
            try:
                y = ops.gemv_forward(x.cuda(), qwc, scc, qzc, g, flags=rows | f)
            except Exception as e:  # a forced slot count the register budget or the 8-wave limit rules out
                assert "code -3" in str(e), e
                assert f != 0, f"K{K} N{N} M{M}: the default configuration must run"
                continue
            assert ops.last_kernel() == "gemv_rows"
            ran += 1
            assert_product_close(y.cpu().numpy().astype(np.float64), y32, f"rows K{K} N{N} g{g} M{M} f{f:x}", wsigma=wsig)
            assert torch.equal(y, ops.gemv_forward(x.cuda(), qwc, scc, qzc, g, flags=rows | f)), "not bitwise reproducible"
        e = torch.zeros((M, K), dtype=torch.float16, device="cuda")
        ks = (torch.arange(M, device="cuda") * 1237 + K - 3) % K
        e[torch.arange(M, device="cuda"), ks] = 1.0
        assert torch.equal(ops.gemv_forward(e, qwc, scc, qzc, g, flags=rows), Wt.t()[ks]), "one-hot rows must select rows of W"
        assert int(ops.gemv_forward(torch.zeros_like(e), qwc, scc, qzc, g, flags=rows).abs().max()) == 0
        y1, y2 = ops.gemv_forward(x.cuda(), qwc, scc, qzc, g, flags=rows).float(), ops.gemv_forward(2 * x.cuda(), qwc, scc, qzc, g, flags=rows).float()
        normal = y1.abs() >= 2.0 ** -13  # (an fp16 SUBNORMAL output is rounded on a coarser grid than its double)
        assert torch.equal(y2[normal], 2 * y1[normal]), "f(2x) != 2 f(x)"
    assert ran >= (8 if small else 2)


# decoder-block prologue / epilogue of the row-streaming kernel (awq_gemv_forward_ex): the 7B block's four projections, the
# 70B ones (K = 8192: one row per super-unit, so (gate, up) pairs are dealt in twos; 28672 -> 8192 needs two waves per row:
# refused), ragged sizes
@pytest.mark.parametrize("K,N,g", [(4096, 12288, 128), (4096, 4096, 128), (4096, 22016, 128), (11008, 4096, 128), (8192, 10240, 128),
                                   (8192, 57344, 128), (8192, 8192, 128), (2048, 202, 2048), (384, 6, 128), (13824, 5120, 128)])
def test_gemv_rows_block_fusions_vs_unfused_and_oracle(ops, oracle, K, N, g):
    """norm prologue, residual epilogue and silu-pairs epilogue against the SEPARATE launches they replace (awq_rmsnorm_forward,
    awq_gemv_forward, torch add, awq_silu_and_mul) -- bit-identical where the arithmetic is the same operation by operation
    (residual, silu pairs); the norm's row statistic is summed in a different order, so a few inputs may round differently --
    and against the CPU oracle end to end."""
    qw, qz, sc, x4 = gemv_case(K, N, g, 1, seed=3 * K + N)
    qwc, qzc, scc = qw.cuda(), qz.cuda(), sc.cuda()
    gen = torch.Generator().manual_seed(K + N)
    x = x4[:1].contiguous()
    nw = (torch.rand((K,), generator=gen) + 0.5).half()

    def same_bits(a, b):  # bit for bit, an overflowed output (inf -> NaN through silu) counts as equal to itself
        return torch.allclose(a.float(), b.float(), rtol=0, atol=0, equal_nan=True)

    res = torch.randn((1, N), generator=gen).half()
    eps = 1e-5
    W = oracle.dequant_gemv(qw.numpy(), qz.numpy(), sc.numpy(), g)
    xc, nwc, resc = x.cuda(), nw.cuda(), res.cuda()
    rows = ops.gemm_flags(kernel=GEMV_KERNEL_ROWS)

    def oracle_norm(xx):
        x32 = xx.numpy().astype(np.float32)
        inv = np.float32(1.0) / np.sqrt(np.mean(x32.astype(np.float64) ** 2, axis=-1, keepdims=True).astype(np.float32) + np.float32(eps))
        return ((x32 * inv) * nw.numpy().astype(np.float32)).astype(np.float16)

    # ---- plain residual epilogue: fp16(fp16(W x) + res), bit for bit
    y_plain = ops.gemv_forward(xc, qwc, scc, qzc, g, flags=rows)
    got = ops.gemv_forward_ex(xc, qwc, scc, qzc, g, add_residual=resc)
    assert ops.last_kernel() == "gemv_rows"
    assert same_bits(got, y_plain + resc), "residual epilogue differs from the separate add"
    # ---- norm prologue (K <= 12288: with eight 1-KiB slots per wave the kernel has no registers left for it and refuses)
    if K > 12288:
        from autoawq_amd._lib import AwqHipError
        with pytest.raises(AwqHipError, match="code -3"):
            ops.gemv_forward_ex(xc, qwc, scc, qzc, g, norm_weight=nwc, norm_eps=eps)
        if N % 2 == 0:
            yp = ops.gemv_forward_ex(xc, qwc, scc, qzc, g, silu_pairs=True)
            assert yp.shape == (1, N // 2) and ops.last_kernel() == "gemv_rows"
        return
    xn = ops.rmsnorm(xc, nwc, eps)
    want = ops.gemv_forward(xn, qwc, scc, qzc, g, flags=rows)
    got = ops.gemv_forward_ex(xc, qwc, scc, qzc, g, norm_weight=nwc, norm_eps=eps)
    y32, _ = oracle.matmul(oracle_norm(x), W)
    wsig = oracle.weight_rounding_sigma(oracle_norm(x), W)
    assert_product_close(got.cpu().numpy().astype(np.float64), y32, f"rows+norm K{K} N{N}", wsigma=wsig)
    same = float((got == want).float().mean())
    assert same >= 0.5, f"norm prologue: only {same:.3f} of the outputs equal the separate launches'"  # (typically 0.95-1.0)
    d = (got.float() - want.float()).abs()
    assert bool((d <= 2e-3 * want.float().abs() + 2e-3 * want.float().abs().mean()).all()), float(d.max())
    got_nr = ops.gemv_forward_ex(xc, qwc, scc, qzc, g, norm_weight=nwc, norm_eps=eps, add_residual=resc)
    assert same_bits(got_nr, got + resc)
    # ---- silu pairs (rows (2 i, 2 i + 1) = (gate_i, up_i)): == awq_silu_and_mul on the separate outputs
    if N % 2 == 0:
        def unfused_silu(yy):  # awq_silu_and_mul takes widths that are multiples of 8: zero padding
            D, D8 = N // 2, (N // 2 + 7) // 8 * 8
            gu = torch.zeros((1, 2 * D8), dtype=torch.float16, device="cuda")
            gu[:, :D], gu[:, D8:D8 + D] = yy[:, 0::2], yy[:, 1::2]
            return ops.silu_and_mul(gu)[:, :D].contiguous()

        want_p = unfused_silu(y_plain)
        got_p = ops.gemv_forward_ex(xc, qwc, scc, qzc, g, silu_pairs=True)
        assert same_bits(got_p, want_p), "silu-pairs epilogue differs from awq_silu_and_mul on the separate outputs"
        want_np = unfused_silu(want)
        got_np = ops.gemv_forward_ex(xc, qwc, scc, qzc, g, norm_weight=nwc, norm_eps=eps, silu_pairs=True)
        d = (got_np.float() - want_np.float()).abs()
        assert bool((d <= 4e-3 * want_np.float().abs() + 4e-3 * want_np.float().abs().mean()).all()), float(d.max())
        # oracle: silu(fp16 gate) * fp16 up from the fp32 product
        gt, up = y32[:, 0::2].astype(np.float16).astype(np.float32), y32[:, 1::2].astype(np.float16).astype(np.float32)
        ref = (gt / (1.0 + np.exp(-gt))) * up
        dd = np.abs(got_np.cpu().numpy().astype(np.float32) - ref)
        assert bool((dd <= 2e-2 * np.abs(ref) + 2e-2 * np.abs(ref).mean()).all()), float(dd.max())
    # bitwise reproducible
    assert torch.equal(got, ops.gemv_forward_ex(xc, qwc, scc, qzc, g, norm_weight=nwc, norm_eps=eps))


def test_gemv_forward_ex_refuses_what_the_rows_kernel_cannot_take(ops):
    """batch > 1, two waves per row (K = 28672), silu pairs with a residual or an odd N: an error code, never a wrong answer"""
    from autoawq_amd._lib import AwqHipError

    qw, qz, sc, x4 = gemv_case(28672, 256, 128, 2, seed=9)
    with pytest.raises(AwqHipError, match="code -3"):
        ops.gemv_forward_ex(x4[:1].cuda(), qw.cuda(), sc.cuda(), qz.cuda(), 128, norm_weight=torch.ones(28672, dtype=torch.float16, device="cuda"))
    qw, qz, sc, x4 = gemv_case(512, 64, 128, 2, seed=9)
    with pytest.raises(AwqHipError, match="code -3"):
        ops.gemv_forward_ex(x4.cuda(), qw.cuda(), sc.cuda(), qz.cuda(), 128, add_residual=torch.zeros((2, 64), dtype=torch.float16, device="cuda"))
    with pytest.raises(AwqHipError):
        ops.gemv_forward_ex(x4[:1].cuda(), qw.cuda(), sc.cuda(), qz.cuda(), 128, silu_pairs=True,
                            add_residual=torch.zeros((1, 32), dtype=torch.float16, device="cuda"))


@pytest.mark.parametrize("K,N", [(4096, 11008), (4096, 22016), (4096, 4096), (11008, 4096), (8192, 1280), (2048, 4099), (1024, 200), (3584, 8192)])
@pytest.mark.parametrize("M", [2, 3, 5, 8, 9, 16])
def test_gemv_lds_kernel_vs_oracle(ops, oracle, K, N, M):
    """csrc/gemv_lds.hip (GEMV layout, weights through LDS by DMA into MFMA 16x16x32): every batch size and shape its LDS
    budget admits (M K <= 32768 and the activations' fragment form + the ring fit 160 KB), waves per tile 1 / 2 / 4, the
    4-wave two-pieces-in-flight variant, ragged N, ragged last piece (K = 11008, 3584), one-hot / zero / 2x properties."""
    g = 128
    qw, qz, sc, x = gemv_case(K, N, g, M, seed=K + 11 * N + M)
    W = oracle.dequant_gemv(qw.numpy(), qz.numpy(), sc.numpy(), g)
    Wt = ops.dequantize_weights_gemv(qw.cuda(), sc.cuda(), qz.cuda(), g)
    y32, _ = oracle.matmul(x.numpy(), W)
    wsig = oracle.weight_rounding_sigma(x.numpy(), W)
    qwc, qzc, scc = qw.cuda(), qz.cuda(), sc.cuda()
    lds = ops.gemm_flags(kernel=3)
    ran = 0
    for f in (0, ops.gemm_flags(splitk=1), ops.gemm_flags(splitk=2), ops.gemm_flags(splitk=4), ops.gemm_flags(unit=2), ops.gemm_flags(unit=3, splitk=2)):
        try:
            y = ops.gemv_forward(x.cuda(), qwc, scc, qzc, g, flags=lds | f)
        except Exception as e:
            assert "code -3" in str(e), e
            continue
        assert ops.last_kernel() == "gemv_lds"
        ran += 1
        assert_product_close(y.cpu().numpy().astype(np.float64), y32, f"lds K{K} N{N} M{M} f{f:x}", wsigma=wsig)
        assert torch.equal(y, ops.gemv_forward(x.cuda(), qwc, scc, qzc, g, flags=lds | f)), "not bitwise reproducible"
    takes = M * K <= 32768 and (K + 1023) // 1024 * (32 * (M + 1) * 64 + 1024) + 8 * 9472 + 8192 <= 160 * 1024
    assert (ran > 0) == takes, (ran, takes)
    if ran:
        e = torch.zeros((M, K), dtype=torch.float16, device="cuda")
        ks = (torch.arange(M, device="cuda") * 977 + K - 5) % K
        e[torch.arange(M, device="cuda"), ks] = 1.0
        assert torch.equal(ops.gemv_forward(e, qwc, scc, qzc, g, flags=lds), Wt.t()[ks]), "one-hot rows must select rows of W"
        assert int(ops.gemv_forward(torch.zeros_like(e), qwc, scc, qzc, g, flags=lds).abs().max()) == 0
        y1, y2 = ops.gemv_forward(x.cuda(), qwc, scc, qzc, g, flags=lds).float(), ops.gemv_forward(2 * x.cuda(), qwc, scc, qzc, g, flags=lds).float()
        normal = y1.abs() >= 2.0 ** -13  # (an fp16 SUBNORMAL output is rounded on a coarser grid than its double)
        assert torch.equal(y2[normal], 2 * y1[normal]), "f(2x) != 2 f(x)"
    # AUTO (round 5): the batched kernel from five rows, else the row-streaming / tile kernels; this kernel stays reachable by flag
    ops.gemv_forward(x.cuda(), qwc, scc, qzc, g)
    want = "gemv_rows" if gemv_rows_takes(M, K, g) else ("gemv_batch" if gemv_batch_takes(M, K, g) else "gemv_nk")
    if M <= 2 or M >= 5 or K <= 4096:
        assert ops.last_kernel() == want, (ops.last_kernel(), want)


def test_gemv_rows_refuses_what_it_cannot_take(ops):
    """g = 64 / 32 (a lane quad would straddle groups), M > 4: AWQ_ERR_UNSUPPORTED when forced, the tile kernel on AUTO"""
    rows = ops.gemm_flags(kernel=GEMV_KERNEL_ROWS)
    for K, N, g, M in [(1024, 72, 64, 1), (512, 40, 32, 1), (4096, 64, 128, 5)]:
        qw, qz, sc, x = gemv_case(K, N, g, M, seed=1)
        with pytest.raises(Exception, match="code -3"):
            ops.gemv_forward(x.cuda(), qw.cuda(), sc.cuda(), qz.cuda(), g, flags=rows)
        ops.gemv_forward(x.cuda(), qw.cuda(), sc.cuda(), qz.cuda(), g)
        assert ops.last_kernel() == ("gemv_batch" if gemv_batch_takes(M, K, g) else "gemv_nk")


GEMV_KERNEL_BATCH = 5
BATCH_SHAPES = [(4096, 12288), (4096, 4096), (4096, 22016), (11008, 4096),       # the four 7B Linears
                (8192, 1280), (1024, 8192), (8192, 7168), (3584, 8192),           # the four 70B TP = 8 shards
                (4096, 11008), (2048, 4099), (1024, 200), (256, 16), (14336, 4096), (16512, 72)]


@pytest.mark.parametrize("K,N", BATCH_SHAPES)
def test_gemv_batch_kernel_vs_oracle(ops, oracle, K, N):
    """csrc/gemv_batch.hip (GEMV layout, round 5: activations as MFMA A fragments in registers, a tile's K range split over the
    eight waves of one block, weights by LDS-DMA): the four 7B and the four 70B-shard shapes, ragged N, one and several passes
    over K (11008: two / three, 14336, 16512: three / five), few tiles (N = 16, 72, 200: idle owners at the barriers), every batch
    1 .. 64 at the benched shape and a ragged sample elsewhere (17, 33 ...: two 16-row tiles with a ragged second one, balanced
    chunks above 32), both ways the activations reach the registers (LDS staging area | direct fragment loads) and every ring depth;
    against the CPU oracle (the
    reference's dequantised fp16 weights, fp32 product), bitwise reproducible, one-hot rows select rows of the bit-exact W, zero
    in -> zero out, f(2x) == 2 f(x)."""
    g = 128
    # (round 6: one launch up to 128 rows -- 33 .. 64 rows as two row parts of a block, 65 .. 128 as four; every M 33 .. 128 at the benched shape)
    all_m = list(range(1, 129)) if (K, N) == (4096, 11008) else [1, 3, 4, 5, 8, 12, 13, 16, 17, 24, 31, 32, 33, 48, 64, 65, 96, 100, 127, 128]
    if K * N > 4096 * 12288:
        all_m = [4, 5, 16, 17, 32, 64, 97, 128]
    MX = 128
    qw, qz, sc, xall = gemv_case(K, N, g, MX, seed=K + 7 * N)
    W = oracle.dequant_gemv(qw.numpy(), qz.numpy(), sc.numpy(), g)
    qwc, qzc, scc, xc = qw.cuda(), qz.cuda(), sc.cuda(), xall.cuda()
    Wt = ops.dequantize_weights_gemv(qwc, scc, qzc, g)
    y32_all, _ = oracle.matmul(xall.numpy(), W)
    wsig_all = oracle.weight_rounding_sigma(xall.numpy(), W)
    yex_all = oracle.matmul_exact_gemv(xall.numpy(), qw.numpy(), qz.numpy(), sc.numpy(), g)
    bt = ops.gemm_flags(kernel=GEMV_KERNEL_BATCH)
    for M in all_m:
        x = xc[MX - M:]  # (a row offset: the row parts of the 33 .. 128-row calls fall elsewhere for every M)
        y32, wsig = y32_all[MX - M:], wsig_all[MX - M:]
        # beside the 6-sigma-widened bound below: the default form within 1 ulp + 1e-4 rms of the EXACT product
        assert_close_to_exact(ops.gemv_forward(x, qwc, scc, qzc, g, flags=bt).cpu().numpy(), yex_all[MX - M:], f"batch K{K} N{N} M{M} vs exact")
        # forced forms: activations through the LDS staging area (unit=1; refused where it does not fit: M > ~12) or by direct
        # fragment loads (unit=2), ring depths 1 .. 3
        variants = [0] if M not in (5, 8, 12, 16, 17, 32, 64, 128) else [0, ops.gemm_flags(unit=1, splitk=1), ops.gemm_flags(unit=1, splitk=2),
                                                                    ops.gemm_flags(unit=2, splitk=1), ops.gemm_flags(unit=2, splitk=2), ops.gemm_flags(unit=2, splitk=3)]
        # the row parts (round 6): inside the block (waves=1) or across 2 / 3 / 4 blocks of one XCD (refused where a part would exceed 32 rows)
        if M in (17, 24, 32, 33, 48, 64, 65, 80, 96, 100, 127, 128):
            variants = variants + [ops.gemm_flags(waves=w) for w in (1, 2, 3, 4)]
        for f in variants:
            try:
                y = ops.gemv_forward(x, qwc, scc, qzc, g, flags=bt | f)
            except Exception as e:
                w = (f >> 24) & 0xF
                staged_refused = ((f >> 20) & 0xF) == 1 and M > 8  # the staged form: only where M KiB per wave fit
                parts_refused = (w >= 2 and (M + w - 1) // w > 32) or (w == 1 and M > 32)  # (in-block parts: where their tile lists overflow LDS)
                assert "code -3" in str(e) and (staged_refused or parts_refused), (e, M, f)
                continue
            assert ops.last_kernel() == "gemv_batch"
            assert_product_close(y.cpu().numpy().astype(np.float64), y32, f"batch K{K} N{N} M{M} f{f:x}", wsigma=wsig)
            assert torch.equal(y, ops.gemv_forward(x, qwc, scc, qzc, g, flags=bt | f)), "not bitwise reproducible"
        ya = ops.gemv_forward(x, qwc, scc, qzc, g)  # AUTO takes it from five rows (four while K > 2048, three while K > 6144)
        if gemv_batch_takes(M, K, g):
            assert ops.last_kernel() == "gemv_batch" and torch.equal(ya, ops.gemv_forward(x, qwc, scc, qzc, g, flags=bt))
        else:
            assert ops.last_kernel() in ("gemv_rows", "gemv_nk"), ops.last_kernel()
    for M in (5, 16, 20, 64, 100):
        e = torch.zeros((M, K), dtype=torch.float16, device="cuda")
        ks = (torch.arange(M, device="cuda") * 977 + K - 5) % K
        e[torch.arange(M, device="cuda"), ks] = 1.0
        assert torch.equal(ops.gemv_forward(e, qwc, scc, qzc, g, flags=bt), Wt.t()[ks]), "one-hot rows must select rows of W"
        assert int(ops.gemv_forward(torch.zeros_like(e), qwc, scc, qzc, g, flags=bt).abs().max()) == 0
        x = xc[:M]
        y1, y2 = ops.gemv_forward(x, qwc, scc, qzc, g, flags=bt).float(), ops.gemv_forward(2 * x, qwc, scc, qzc, g, flags=bt).float()
        normal = y1.abs() >= 2.0 ** -13
        assert torch.equal(y2[normal], 2 * y1[normal]), "f(2x) != 2 f(x)"


def test_gemv_batch_refuses_what_it_cannot_take(ops):
    """group sizes other than 128: AWQ_ERR_UNSUPPORTED when forced, the older decode kernels on AUTO"""
    bt = ops.gemm_flags(kernel=GEMV_KERNEL_BATCH)
    for K, N, g, M in [(1024, 72, 64, 8), (512, 40, 32, 8), (2048, 200, 2048, 8)]:
        qw, qz, sc, x = gemv_case(K, N, g, M, seed=1)
        with pytest.raises(Exception, match="code -3"):
            ops.gemv_forward(x.cuda(), qw.cuda(), sc.cuda(), qz.cuda(), g, flags=bt)
        ops.gemv_forward(x.cuda(), qw.cuda(), sc.cuda(), qz.cuda(), g)
        assert ops.last_kernel() == "gemv_nk"


@pytest.mark.parametrize("K,N,g", [(4096, 11008, 128), (11008, 4096, 128), (1024, 200, 128), (512, 264, 64), (256, 8, 32), (2048, 4104, 2048), (8192, 1280, 128)])
def test_repack_gemv_to_gemm_bit_exact(ops, oracle, K, N, g):
    """csrc/repack.hip (the prefill route of WQLinear_GEMV: GEMV-layout buffers -> a GEMM-layout temporary of the call): the packed
    words, zero words and scales it writes equal, bit for bit, what the torch repack of utils/convert.py produces (pinned against
    reference-written checkpoints in tests/test_checkpoint.py) -- ragged tiles in both directions, every group size; and the
    product on the repacked buffers equals the product on a GEMM-layout packing of the same integers."""
    from autoawq_amd import WQLinear_GEMV
    from autoawq_amd.utils.convert import convert_linear

    qw, qz, sc, x = gemv_case(K, N, g, 40, seed=K + N + g)
    m = WQLinear_GEMV(4, g, K, N, False, "cuda")
    m.qweight, m.qzeros, m.scales = qw.cuda(), qz.cuda(), sc.cuda()
    want = convert_linear(m, "gemm")
    rq, rs, rz = ops.repack_gemv_to_gemm(m.qweight, m.scales, m.qzeros, g)
    assert torch.equal(rq, want.qweight) and torch.equal(rz, want.qzeros)
    assert torch.equal(rs.view(torch.int16), want.scales.view(torch.int16))
    y = ops.gemv_prefill_repack(x.cuda(), m.qweight, m.scales, m.qzeros, g)
    assert torch.equal(y, ops.gemm_forward(x.cuda(), want.qweight, want.scales, want.qzeros))
    W = oracle.dequant_gemv(qw.numpy(), qz.numpy(), sc.numpy(), g)
    y32, _ = oracle.matmul(x.numpy(), W)
    assert_product_close(y.cpu().numpy().astype(np.float64), y32, f"repack route K{K} N{N} g{g}", wsigma=oracle.weight_rounding_sigma(x.numpy(), W))


def test_gemv_module_forward_semantics(ops, oracle):
    from autoawq_amd import WQLinear_GEMV

    g = golden("packed_K512_N64_g128")
    m = WQLinear_GEMV(4, 128, 512, 64, True, "cuda")
    m.qweight, m.qzeros, m.scales, m.bias = (dev(g["gemv_qweight"]), dev(g["gemv_qzeros"]), dev(g["gemv_scales"]),
                                             dev(g["bias"]))
    x = dev(g["x"])
    y32, _ = oracle.matmul(g["x"], g["W"], g["bias"])
    wsig = oracle.weight_rounding_sigma(g["x"], g["W"])
    out = m(x.view(1, 4, 512))
    assert out.shape == (1, 4, 64) and out.dtype == torch.float16
    # bias is added after the fp16 rounding of the product (gemv.py:183-185): one more ulp
    tol_ulp = np.maximum(np.abs(y32), 2.0 ** -14) * 2.0 ** -10
    assert (np.abs(out[0].cpu().numpy().astype(np.float64) - y32) <= product_tol_(y32) + 6 * wsig + 2 * tol_ulp).all()
    o32 = m(x.float())
    assert o32.dtype == torch.float32 and o32.shape == (4, 64)
    big = torch.randn((70, 512), generator=torch.Generator().manual_seed(2)).half()
    yb, _ = oracle.matmul(big.numpy(), g["W"], g["bias"])
    ob = m(big.cuda())  # 70 rows: still below prefill_min_rows -- the batched-decode kernel, one launch (row parts across blocks)
    assert (np.abs(ob.cpu().numpy().astype(np.float64) - yb) <= product_tol_(yb) + 3 * np.maximum(np.abs(yb), 2.0 ** -14) * 2.0 ** -10).all()


def product_tol_(ref32):
    from conftest import assert_close_to_exact, product_tol
    return product_tol(ref32)


# ------------------------------------------------------------------ fused MLP / MoE

def test_silu_and_mul_vs_oracle(ops, oracle):
    gen = torch.Generator().manual_seed(4)
    gu = (torch.randn((37, 2 * 1408), generator=gen) * 3).half()
    gu.view(-1)[:6] = torch.tensor([0.0, -0.0, 65504, -65504, 1e-4, -20.0]).half()
    want = oracle.silu_and_mul(gu.numpy())
    got = ops.silu_and_mul(gu.cuda()).cpu().numpy()
    w32, g32 = want.astype(np.float32), got.astype(np.float32)
    fin = np.isfinite(w32)
    assert np.array_equal(np.isfinite(g32), fin)
    ulp = np.maximum(np.abs(w32[fin]), 2.0 ** -14) * 2.0 ** -10
    assert (np.abs(g32[fin] - w32[fin]) <= ulp).all()
    assert (got != want).mean() < 0.01  # the two exp() implementations agree almost everywhere


def stacked_experts(E, K, N, g, seed):
    gen = torch.Generator().manual_seed(seed)
    qw = torch.randint(MIN_INT32, MAX_INT32, (E, K, N // 8), dtype=torch.int32, generator=gen)
    qz = torch.randint(MIN_INT32, MAX_INT32, (E, K // g, N // 8), dtype=torch.int32, generator=gen)
    sc = (torch.rand((E, K // g, N), generator=gen) * 0.02 + 0.005).half()
    return qw, qz, sc


@pytest.mark.parametrize("rows", [8, 16])
@pytest.mark.parametrize("T,E,topk,K,N", [(4, 8, 2, 256, 512), (1, 8, 2, 512, 256), (19, 4, 2, 256, 1024), (40, 8, 2, 128, 256)])
def test_grouped_gemm_vs_oracle_unpinned_in_the_reference(ops, oracle, T, E, topk, K, N, rows):
    """grouped_gemm_forward (moe.py:60-89): every (token, slot) pair against its own expert."""
    g = 128
    qw, qz, sc = stacked_experts(E, K, N, g, seed=T + E + K)
    gen = torch.Generator().manual_seed(7)
    x = torch.randn((T, K), generator=gen).half()
    logits = torch.randn((T, E), generator=gen)
    w, ids = ops.fused_topk(logits.cuda(), topk, True)
    s_ids, e_ids, npad = ops.moe_align_block_size(ids, rows, E)
    y = ops.grouped_gemm_forward(x.cuda().view(T, 1, K), qw.cuda(), sc.cuda(), qz.cuda(), w, s_ids, e_ids, npad, False,
                                 block_rows=rows)
    assert y.shape == (T, topk, N) and ops.last_kernel() == "gemv_mfma_grouped"
    y2 = ops.grouped_gemm_forward(x.cuda().view(T, 1, K), qw.cuda(), sc.cuda(), qz.cuda(), w, s_ids, e_ids, npad, True,
                                  block_rows=rows)
    idc, wc = ids.cpu().numpy(), w.cpu().numpy()
    for t in range(T):
        for j in range(topk):
            e = int(idc[t, j])
            ref32, _ = oracle.linear_gemm(x[t:t + 1].numpy(), qw[e].numpy(), qz[e].numpy(), sc[e].numpy(), g)
            W = oracle.dequant_gemm(qw[e].numpy(), qz[e].numpy(), sc[e].numpy(), g)
            sig = oracle.weight_rounding_sigma(x[t:t + 1].numpy(), W)
            assert_product_close(y[t, j].cpu().numpy().astype(np.float64)[None], ref32, f"pair {t},{j}", wsigma=sig)
            assert_product_close(y2[t, j].cpu().numpy().astype(np.float64)[None], ref32 * wc[t, j], f"weighted {t},{j}",
                                 wsigma=sig * wc[t, j])
    assert ops.workspace_is_clean(y.device)


def test_moe_block_vs_oracle_unpinned_in_the_reference(ops, oracle):
    """FusedSparseMoeBlock / apply_moe_weights (moe.py:12-91) on a Mixtral-shaped toy: E=8, top-2."""
    from autoawq_amd.modules.fused.moe import FusedSparseMoeBlock

    T, E, H, I, g = 4, 8, 256, 384, 128
    w1q, w1z, w1s = stacked_experts(E, H, 2 * I, g, seed=1)
    w2q, w2z, w2s = stacked_experts(E, I, H, g, seed=2)
    gen = torch.Generator().manual_seed(3)
    x = torch.randn((T, H), generator=gen).half()
    gate = torch.nn.Linear(H, E, bias=False).half()
    gate.weight.data = (torch.randn((E, H), generator=gen) * 0.1).half()

    class Stack:  # what fuse_linears(..., operation=torch.stack) returns: an object with the buffers
        pass
    ws, w2 = Stack(), Stack()
    ws.qweight, ws.qzeros, ws.scales = w1q.cuda(), w1z.cuda(), w1s.cuda()
    w2.qweight, w2.qzeros, w2.scales = w2q.cuda(), w2z.cuda(), w2s.cuda()
    blk = FusedSparseMoeBlock(2, gate.cuda(), ws, w2)
    with torch.no_grad():
        out = blk(x.cuda().view(1, T, H))
    assert out.shape == (1, T, H)
    logits = blk.gate(x.cuda()).detach().float().cpu().numpy()  # the router itself is not on the int4 path
    want, ids, wt = oracle.moe_forward(x.numpy(), logits, dict(qweight=w1q.numpy(), qzeros=w1z.numpy(), scales=w1s.numpy()),
                                       dict(qweight=w2q.numpy(), qzeros=w2z.numpy(), scales=w2s.numpy()), 2, g)
    got = out[0].cpu().numpy().astype(np.float64)
    w32 = want.astype(np.float64)
    rms = np.sqrt((w32 ** 2).mean())
    assert (np.abs(got - w32) <= 4e-3 * np.abs(w32) + 4e-3 * rms).all(), np.abs(got - w32).max() / rms


@pytest.mark.parametrize("T", [1, 4, 40])
def test_moe_activation_folded_into_w2_is_bit_identical(ops, T):
    """AWQ_GEMM_FLAG_X_GATED_SILU on the grouped GEMM: silu(gate) * up applied while staging == the separate awq_silu_and_mul
    launch, bit for bit (8- and 16-row blocks)."""
    from autoawq_amd.modules.fused import moe

    E, H, I, g, topk = 8, 512, 768, 128, 2
    w1q, w1z, w1s = stacked_experts(E, H, 2 * I, g, seed=11)
    w2q, w2z, w2s = stacked_experts(E, I, H, g, seed=12)

    class Stack:
        pass
    ws, w2 = Stack(), Stack()
    ws.qweight, ws.qzeros, ws.scales = w1q.cuda(), w1z.cuda(), w1s.cuda()
    w2.qweight, w2.qzeros, w2.scales = w2q.cuda(), w2z.cuda(), w2s.cuda()
    gen = torch.Generator(device="cuda").manual_seed(T)
    x = torch.randn((T, H), device="cuda", generator=gen).half()
    logits = torch.randn((T, E), device="cuda", generator=gen)
    saved = moe.FUSE_ACTIVATION_INTO_W2
    try:
        moe.FUSE_ACTIVATION_INTO_W2 = True
        a = moe.apply_moe_weights(ws, w2, x, logits, topk, True)
        moe.FUSE_ACTIVATION_INTO_W2 = False
        b = moe.apply_moe_weights(ws, w2, x, logits, topk, True)
    finally:
        moe.FUSE_ACTIVATION_INTO_W2 = saved
    assert torch.equal(a, b)


def test_moe_prefill_path_vs_oracle_and_vs_the_block_path_unpinned_in_the_reference(ops, oracle):
    """T = 512 tokens, top-2 of 8 experts (1024 pairs, ~128 rows per expert): apply_moe_weights sorts the pairs by expert on the
    device and runs ONE grouped launch of the register-decoded MFMA GEMM per projection (awq_grouped_gemm_prefill,
    modules/fused/moe.py::_apply_moe_prefill; round 3: one GEMM per expert and a host read-back).  Against the CPU oracle on a
    sample of tokens, against the 16-row-block grouped kernel on all of them, against one awq_gemm_forward per expert for the
    kernel alone (ragged and EMPTY experts included), and captured into a hipGraph (nothing is read back)."""
    from autoawq_amd.modules.fused import moe

    T, E, H, I, g, topk = 512, 8, 512, 768, 128, 2
    w1q, w1z, w1s = stacked_experts(E, H, 2 * I, g, seed=21)
    w2q, w2z, w2s = stacked_experts(E, I, H, g, seed=22)

    class Stack:
        pass
    ws, w2 = Stack(), Stack()
    ws.qweight, ws.qzeros, ws.scales = w1q.cuda(), w1z.cuda(), w1s.cuda()
    w2.qweight, w2.qzeros, w2.scales = w2q.cuda(), w2z.cuda(), w2s.cuda()
    gen = torch.Generator().manual_seed(5)
    x = torch.randn((T, H), generator=gen).half()
    logits = torch.randn((T, E), generator=gen)
    assert T * topk >= moe.PREFILL_MIN_PAIRS
    got = moe.apply_moe_weights(ws, w2, x.cuda(), logits.cuda(), topk, True)
    assert ops.last_kernel() != "none"
    # the grouped kernel alone: rows sorted by expert with a ragged split (one expert empty, one with a single row, one with
    # 257 rows = three 128-row tiles) == one awq_gemm_forward per expert on its rows, bit for bit
    counts = torch.tensor([0, 1, 257, 100, 129, 128, 17, 300], dtype=torch.int32)
    seg = torch.zeros(E + 1, dtype=torch.int32)
    seg[1:] = torch.cumsum(counts, 0)
    P = int(seg[-1])
    xs = torch.randn((P, H), generator=gen).half().cuda()
    for bm in (1, 2):
        yg = ops.grouped_gemm_prefill(xs, ws.qweight, ws.scales, ws.qzeros, seg.cuda(), flags=ops.gemm_flags(nlog=bm))
        assert ops.last_kernel() == "gemm_regb_grouped" and yg.shape == (P, 2 * I)
        for e in range(E):
            lo, hi = int(seg[e]), int(seg[e + 1])
            if hi > lo:
                ref = ops.gemm_forward(xs[lo:hi], ws.qweight[e], ws.scales[e], ws.qzeros[e], flags=ops.gemm_flags(ops.KERNEL_REGB, nlog=bm))
                assert torch.equal(yg[lo:hi], ref), (bm, e)
    s_ = torch.cuda.Stream()
    with torch.cuda.stream(s_):
        xc, lc = x.cuda(), logits.cuda()
        moe.apply_moe_weights(ws, w2, xc, lc, topk, True)
        s_.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s_):
            captured = moe.apply_moe_weights(ws, w2, xc, lc, topk, True)
        gr.replay()
        s_.synchronize()
    assert torch.equal(captured, got), "the captured MoE prefill block must replay to the same result"
    saved = moe.PREFILL_MIN_PAIRS
    try:
        moe.PREFILL_MIN_PAIRS = 1 << 30
        blocks = moe.apply_moe_weights(ws, w2, x.cuda(), logits.cuda(), topk, True)
    finally:
        moe.PREFILL_MIN_PAIRS = saved
    # two valid evaluation orders: the fp16 [gate | up] intermediate is rounded by different kernels, and that rounding noise
    # passes through silu * up and the 768-term w2 sum; the oracle below is the parity check proper
    d = (got.float() - blocks.float()).abs()
    assert bool((d <= 1e-2 * blocks.float().abs() + 1e-2 * blocks.float().abs().mean()).all()), float(d.max())
    sample = torch.arange(0, T, 16)
    want, _, _ = oracle.moe_forward(x[sample].numpy(), logits[sample].numpy(), dict(qweight=w1q.numpy(), qzeros=w1z.numpy(), scales=w1s.numpy()),
                                    dict(qweight=w2q.numpy(), qzeros=w2z.numpy(), scales=w2s.numpy()), topk, g)
    g64, w64 = got[sample.cuda()].cpu().numpy().astype(np.float64), want.astype(np.float64)
    rms = np.sqrt((w64 ** 2).mean())
    assert (np.abs(g64 - w64) <= 4e-3 * np.abs(w64) + 4e-3 * rms).all(), np.abs(g64 - w64).max() / rms


@pytest.mark.parametrize("layout", ["gemm", "gemv"])
def test_quant_fused_mlp_vs_oracle(ops, oracle, layout):
    """QuantFusedMLP (mlp.py:14-70): down(silu(gate(x)) * up(x)), one fused gate|up launch."""
    from autoawq_amd import WQLinear_GEMM, WQLinear_GEMV
    from autoawq_amd.modules.fused.mlp import QuantFusedMLP

    H, I, g, M = 256, 512, 128, 3
    gen = torch.Generator().manual_seed(11)
    x = torch.randn((1, M, H), generator=gen).half()
    mods, Ws = [], []
    for (K, N, seed) in [(H, I, 1), (I, H, 2), (H, I, 3)]:  # gate, down, up
        if layout == "gemm":
            qw, qz, sc, _, _ = fullrange_case(K, N, g, 1, seed=seed, realistic=True)
            m = WQLinear_GEMM(4, g, K, N, False, "cuda")
            W = oracle.dequant_gemm(qw.numpy(), qz.numpy(), sc.numpy(), g)
        else:
            qw, qz, sc, _ = gemv_case(K, N, g, 1, seed=seed)
            m = WQLinear_GEMV(4, g, K, N, False, "cuda")
            W = oracle.dequant_gemv(qw.numpy(), qz.numpy(), sc.numpy(), g)
        m.qweight, m.qzeros, m.scales = qw.cuda(), qz.cuda(), sc.cuda()
        mods.append(m)
        Ws.append(W)
    mlp = QuantFusedMLP(mods[0], mods[1], mods[2])
    out = mlp(x.cuda())
    assert out.shape == (1, M, H)
    x2 = x[0].numpy()
    g32, g16 = oracle.matmul(x2, Ws[0])
    u32, u16 = oracle.matmul(x2, Ws[2])
    h = oracle.silu_and_mul(np.concatenate([g16, u16], axis=1))
    y32, _ = oracle.matmul(h, Ws[1])
    rms = np.sqrt((y32.astype(np.float64) ** 2).mean())
    err = np.abs(out[0].cpu().numpy().astype(np.float64) - y32)
    assert (err <= 4e-3 * np.abs(y32) + 4e-3 * rms).all(), err.max() / rms
    rw = torch.tensor([[0.25], [0.5], [1.0]], dtype=torch.float16, device="cuda")
    assert torch.equal(mlp(x.cuda()[0], routing_weights=rw), rw * mlp(x.cuda()[0]))
    if layout == "gemm":  # the activation fused into the down projection == the separate silu_and_mul launch, bit for bit
        assert mlp.FUSE_ACTIVATION_INTO_DOWN
        mlp.FUSE_ACTIVATION_INTO_DOWN = False
        unfused = mlp(x.cuda())
        mlp.FUSE_ACTIVATION_INTO_DOWN = True
        assert torch.equal(out, unfused)


@pytest.mark.parametrize("M", [1, 2, 8, 9, 16])
@pytest.mark.parametrize("K,N", [(512, 256), (11008, 4096), (1408, 4096)])
def test_gated_silu_staging_equals_separate_kernel(ops, M, K, N):
    """AWQ_GEMM_FLAG_X_GATED_SILU: x = [gate | up]; staging silu(gate) * up inside the decode kernel gives
    the result of awq_silu_and_mul followed by the plain call, bitwise (same fp32 formula, one rounding)."""
    qw, qz, s, _, bias = fullrange_case(K, N, 128, 1, seed=K + M, realistic=True)
    gen = torch.Generator().manual_seed(M)
    gu = (torch.randn((M, 2 * K), generator=gen) * 2).half().cuda()
    dq, ds, dz, db = qw.cuda(), s.cuda(), qz.cuda(), bias.cuda()
    want = ops.gemm_forward(ops.silu_and_mul(gu), dq, ds, dz, db, flags=ops.gemm_flags(ops.KERNEL_MFMA_GEMV))  # the same kernel, plain
    got = ops.gemm_forward(gu, dq, ds, dz, db, flags=ops.X_GATED_SILU)
    assert ops.last_kernel() == "gemv_mfma"
    assert torch.equal(got, want)
    with pytest.raises(Exception):
        ops.gemm_forward(torch.zeros((17, 2 * K), dtype=torch.float16, device="cuda"), dq, ds, dz, flags=ops.X_GATED_SILU)


# ------------------------------------------------------------------ GEMVFast layout (WQLinear_GEMVFast)

def gemvfast_case(K, N, g, M, seed):
    from autoawq_amd.utils.packing import calculate_zeros_width

    gen = torch.Generator().manual_seed(seed)
    gp = calculate_zeros_width(K, g) * 8
    G = K // g
    qw = torch.randint(-32768, 32767, (N // 4, K), dtype=torch.int16, generator=gen)
    sc = torch.zeros((gp, N), dtype=torch.float16)
    sc[:G] = (torch.rand((G, N), generator=gen) * 0.02 + 0.005).half()
    qz = torch.zeros((gp, N), dtype=torch.float16)
    qz[:G] = -(sc[:G].float() * torch.randint(0, 16, (G, N), generator=gen).float()).half()
    x = torch.randn((M, K), generator=gen).half()
    return qw, sc, qz, x


@pytest.mark.parametrize("name", ["packed_K512_N64_g128", "packed_K256_N32_g64", "packed_K128_N32_g32"])
def test_gemvfast_layout_golden(ops, oracle, name):
    g = golden(name)
    gs = int(g["group_size"])
    qw, sc, qz = dev(g["fast_qweight"]), dev(g["fast_scales"]), dev(g["fast_qzeros"])
    W = oracle.dequant_gemvfast(g["fast_qweight"], g["fast_scales"], g["fast_qzeros"], gs)  # [K, N]
    Wt = ops.dequantize_weights_gemv_fast(qw, sc, qz, gs)
    assert np.array_equal(Wt.cpu().numpy().view(np.uint16), np.ascontiguousarray(W.T).view(np.uint16))
    y = ops.gemv_fast_forward(dev(g["x"]), qw, sc, qz, gs).cpu().numpy().astype(np.float64)
    y32, _ = oracle.matmul(g["x"], W)
    assert_product_close(y, y32, f"{name} gemvfast", wsigma=oracle.weight_rounding_sigma(g["x"], W))


@pytest.mark.parametrize("K,N,g", [(4096, 4096, 128), (11008, 4096, 128), (1024, 80, 64), (512, 48, 32), (2048, 208, 2048)])
@pytest.mark.parametrize("M", [1, 3, 8, 16, 20])
def test_gemvfast_layout_vs_oracle_unpinned_in_the_reference(ops, oracle, K, N, g, M):
    qw, sc, qz, x = gemvfast_case(K, N, g, M, seed=K + N + M)
    W = oracle.dequant_gemvfast(qw.numpy(), sc.numpy(), qz.numpy(), g)
    Wt = ops.dequantize_weights_gemv_fast(qw.cuda(), sc.cuda(), qz.cuda(), g)
    assert np.array_equal(Wt.cpu().numpy().view(np.uint16), np.ascontiguousarray(W.T).view(np.uint16))
    y32, _ = oracle.matmul(x.numpy(), W)
    wsig = oracle.weight_rounding_sigma(x.numpy(), W)
    # AUTO: from five rows at group size 128 the batched kernel (round 5); kernel=1 forces the 16-row kernel (csrc/gemv_fast.hip)
    for flags in (0, ops.gemm_flags(kernel=1, waves=4, unit=8), ops.gemm_flags(kernel=1, waves=16, unit=4)):
        y = ops.gemv_fast_forward(x.cuda(), qw.cuda(), sc.cuda(), qz.cuda(), g, flags=flags)
        # (below five rows AUTO takes the batched kernel where one pass of eight waves covers K: profiles/r05_sweep_small_batch.txt)
        batch = flags == 0 and (M >= 5 or 2048 < K <= 4096) and g == 128 and N % 16 == 0
        assert ops.last_kernel() == ("gemv_batch_fast" if batch else "gemv_fast"), ops.last_kernel()
        assert_product_close(y.cpu().numpy().astype(np.float64), y32, f"gemvfast K{K} N{N} g{g} M{M}", wsigma=wsig)
    e = torch.zeros((M, K), dtype=torch.float16, device="cuda")
    ks = (torch.arange(M, device="cuda") * 29 + 11) % K
    e[torch.arange(M, device="cuda"), ks] = 1.0
    assert torch.equal(ops.gemv_fast_forward(e, qw.cuda(), sc.cuda(), qz.cuda(), g), Wt.t()[ks])


@pytest.mark.parametrize("K,N", [(4096, 11008), (4096, 4096), (11008, 4096), (8192, 1280), (1024, 8192), (256, 16), (4224, 208)])
def test_gemvfast_batch_kernel_vs_oracle_unpinned_in_the_reference(ops, oracle, K, N):
    """csrc/gemv_batch.hip in its GEMVFast form (round 5: the batched kernel on qweight int16 [N/4, K], scales / qzeros fp16 [GP, N]; the
    reference runs awq_v2_ext.gemm_forward_cuda_prefill there, gemv_fast.py:203-206): 7B and 70B-shard shapes, one and several passes
    over K, few tiles, every ring form, batches 1 .. 128 (round 6: one launch, 33 .. 128 rows as two / four row parts of a block); against the oracle's W = fp16(w s + qzeros) and fp32 product;
    one-hot rows select rows of the bit-exact W^T; bitwise reproducible."""
    g = 128
    qw, sc, qz, xall = gemvfast_case(K, N, g, 128, seed=K + 3 * N)
    W = oracle.dequant_gemvfast(qw.numpy(), sc.numpy(), qz.numpy(), g)
    qwc, scc, qzc, xc = qw.cuda(), sc.cuda(), qz.cuda(), xall.cuda()
    Wt = ops.dequantize_weights_gemv_fast(qwc, scc, qzc, g)
    y32_all, _ = oracle.matmul(xall.numpy(), W)
    wsig_all = oracle.weight_rounding_sigma(xall.numpy(), W)
    yex_all = oracle.matmul_exact_gemvfast(xall.numpy(), qw.numpy(), sc.numpy(), qz.numpy(), g)  # (unpinned in the reference itself: autoawq-kernels)
    bt = ops.gemm_flags(kernel=GEMV_KERNEL_BATCH)
    for M in ([1, 2, 3, 4, 5, 8, 12, 16, 17, 31, 32, 33, 64, 65, 96, 127, 128] if K * N <= 4096 * 11008 else [1, 4, 5, 16, 32, 64, 128]):
        x = xc[128 - M:]
        y32, wsig = y32_all[128 - M:], wsig_all[128 - M:]
        # beside the 6-sigma-widened bound below: within 1 ulp + 1e-4 rms of the EXACT product w s + qzeros
        assert_close_to_exact(ops.gemv_fast_forward(x, qwc, scc, qzc, g, flags=bt).cpu().numpy(), yex_all[128 - M:], f"batch-fast K{K} N{N} M{M} vs exact")
        # forced forms: ring slots 1 | 2; row parts inside the block (waves=1) or across 2 / 3 / 4 blocks of an XCD (refused where a part
        # would exceed 32 rows)
        forms = [0, ops.gemm_flags(splitk=1), ops.gemm_flags(splitk=2)] if M in (1, 5, 16, 17, 32) else [0]
        if M in (17, 32, 33, 64, 96, 127, 128):
            forms += [ops.gemm_flags(waves=w) for w in (1, 2, 3, 4)]
        for f in forms:
            try:
                y = ops.gemv_fast_forward(x, qwc, scc, qzc, g, flags=bt | f)
            except Exception as e:
                w = (f >> 24) & 0xF
                assert "code -3" in str(e) and (w == 1 and M > 32 or w >= 2 and (M + w - 1) // w > 32), (e, M, f)  # (in-block parts: where their tile lists overflow LDS)
                continue
            assert ops.last_kernel() == "gemv_batch_fast"
            assert_product_close(y.cpu().numpy().astype(np.float64), y32, f"batch-fast K{K} N{N} M{M} f{f:x}", wsigma=wsig)
            assert torch.equal(y, ops.gemv_fast_forward(x, qwc, scc, qzc, g, flags=bt | f)), "not bitwise reproducible"
        ya = ops.gemv_fast_forward(x, qwc, scc, qzc, g)
        if M >= 5 or 2048 < K <= 4096:  # AUTO takes it
            assert ops.last_kernel() == "gemv_batch_fast" and torch.equal(ya, ops.gemv_fast_forward(x, qwc, scc, qzc, g, flags=bt))
        else:
            assert ops.last_kernel() == "gemv_fast"
    for M in (5, 16, 20):
        e = torch.zeros((M, K), dtype=torch.float16, device="cuda")
        ks = (torch.arange(M, device="cuda") * 977 + K - 5) % K
        e[torch.arange(M, device="cuda"), ks] = 1.0
        assert torch.equal(ops.gemv_fast_forward(e, qwc, scc, qzc, g, flags=bt), Wt.t()[ks]), "one-hot rows must select rows of W"
        assert int(ops.gemv_fast_forward(torch.zeros_like(e), qwc, scc, qzc, g, flags=bt).abs().max()) == 0


def test_gemvfast_module_forward(ops, oracle):
    from autoawq_amd import WQLinear_GEMVFast

    g = golden("packed_K512_N64_g128")
    m = WQLinear_GEMVFast(4, 128, 512, 64, True, "cuda")
    m.qweight, m.scales, m.qzeros, m.bias = (dev(g["fast_qweight"]), dev(g["fast_scales"]), dev(g["fast_qzeros"]),
                                             dev(g["bias"]))
    W = oracle.dequant_gemvfast(g["fast_qweight"], g["fast_scales"], g["fast_qzeros"], 128)
    y32, _ = oracle.matmul(g["x"], W, g["bias"])
    out = m(dev(g["x"]).view(4, 1, 512))
    assert out.shape == (4, 1, 64) and out.dtype == torch.float16
    ulp = np.maximum(np.abs(y32), 2.0 ** -14) * 2.0 ** -10
    wsig = oracle.weight_rounding_sigma(g["x"], W)
    assert (np.abs(out[:, 0].cpu().numpy().astype(np.float64) - y32) <= product_tol_(y32) + 6 * wsig + 2 * ulp).all()


@pytest.mark.parametrize("K,N,g", [(320, 64, 64), (192, 32, 32)])
def test_gemv_layout_modules_fall_back_where_the_decode_kernels_refuse(ops, oracle, K, N, g):
    """K % 128 != 0: awq_gemv_forward and
    awq_gemv_fast_forward answer UNSUPPORTED; the modules then dequantise (bit-exact HIP kernels) and multiply, so every
    valid tensor of the two layouts has a forward (the reference's kernels take these shapes, gemv.py:168-180)."""
    from autoawq_amd import WQLinear_GEMV, WQLinear_GEMVFast, _lib

    M = 3
    qw, qz, sc, x = gemv_case(K, N, g, M, seed=K + N)
    with pytest.raises(_lib.AwqHipError) as ei:
        ops.gemv_forward(x.cuda(), qw.cuda(), sc.cuda(), qz.cuda(), g)
    assert ei.value.code == _lib.ERR_UNSUPPORTED
    m = WQLinear_GEMV(4, g, K, N, False, "cuda")
    m.qweight, m.qzeros, m.scales = qw.cuda(), qz.cuda(), sc.cuda()
    W = oracle.dequant_gemv(qw.numpy(), qz.numpy(), sc.numpy(), g)
    y32, _ = oracle.matmul(x.numpy(), W)
    assert_product_close(m(x.cuda()).cpu().numpy().astype(np.float64), y32, f"gemv module fallback K{K} g{g}")
    fq, fs, fz, fx = gemvfast_case(K, N, g, M, seed=K + N + 1)
    with pytest.raises(_lib.AwqHipError) as ei:
        ops.gemv_fast_forward(fx.cuda(), fq.cuda(), fs.cuda(), fz.cuda(), g)
    assert ei.value.code == _lib.ERR_UNSUPPORTED
    mf = WQLinear_GEMVFast(4, g, K, N, False, "cuda")
    mf.qweight, mf.scales, mf.qzeros = fq.cuda(), fs.cuda(), fz.cuda()
    Wf = oracle.dequant_gemvfast(fq.numpy(), fs.numpy(), fz.numpy(), g)
    yf, _ = oracle.matmul(fx.numpy(), Wf)
    assert_product_close(mf(fx.cuda().view(M, 1, K))[:, 0].cpu().numpy().astype(np.float64), yf, f"gemvfast module fallback K{K} g{g}")


@pytest.mark.parametrize("T,E,k,blk", [(4, 8, 2, 8), (1, 8, 2, 16), (37, 8, 2, 16), (64, 16, 4, 16), (200, 64, 8, 8), (5, 3, 1, 4)])
def test_moe_route_kernel_vs_torch_and_oracle(ops, oracle, T, E, k, blk):
    """one-launch routing == softmax/topk (torch, the reference's ROCm branch moe.py:152-156) + the
    oracle's moe_align restatement."""
    logits = torch.randn((T, E), generator=torch.Generator().manual_seed(T + E)) * 2
    w, ids, s_ids, e_ids, npad = ops.moe_route(logits.cuda(), k, True, blk)
    tw, ti = ops.fused_topk(logits.cuda(), k, True)
    assert torch.equal(ids, ti)
    assert torch.allclose(w, tw, rtol=1e-5, atol=1e-7)
    ws, we, wn = oracle.moe_align(ids.cpu().numpy(), E, blk)
    assert int(npad) == wn
    assert np.array_equal(s_ids.cpu().numpy(), ws)
    assert np.array_equal(e_ids.cpu().numpy()[: wn // blk], we[: wn // blk])
    w2, _, _, _, _ = ops.moe_route(logits.cuda(), k, False, blk)
    assert torch.allclose(w2, torch.topk(torch.softmax(logits.float(), -1), k, -1)[0].cuda(), rtol=1e-5, atol=1e-7)


# ------------------------------------------------------------------ tensor-parallel shards on one GPU

@pytest.mark.parametrize("world", [2, 8])
def test_tp_shards_sequentially_on_one_gpu(ops, oracle, world):
    """the sharding math is provable on one device by running the per-rank shards one
    after the other -- column-parallel shards concatenate, row-parallel partial sums (whole
    groups per rank, uneven 86-group split of the Llama-2-7B down_proj) add up to the unsharded
    product; the kernels see the per-rank shard shapes bench.py --gpus N uses."""
    from autoawq_amd import WQLinear_GEMM
    from autoawq_amd.tp import ColumnParallelWQLinear, RowParallelWQLinear, split_even_units

    K, N, g, M = 11008, 4096, 128, 2  # down_proj: 86 groups
    qw, qz, s, x, bias = fullrange_case(K, N, g, M, seed=77, realistic=True)
    full = WQLinear_GEMM(4, g, K, N, True, "cuda")
    full.qweight, full.qzeros, full.scales, full.bias = qw.cuda(), qz.cuda(), s.cuda(), bias.cuda()
    y_full = full(x.cuda())
    parts = [c for _, c in split_even_units(K // g, world)]
    assert sum(parts) == 86 and max(parts) - min(parts) <= 1
    acc = torch.zeros((M, N), dtype=torch.float32, device="cuda")
    for r in range(world):
        rp = RowParallelWQLinear(full, r, world)
        k0, k1 = rp.bounds
        assert k0 % g == 0 and k1 % g == 0 and (rp.shard.bias is not None) == (r == 0)
        acc += rp.shard(x.cuda()[:, k0:k1]).float()
    y32, _ = oracle.linear_gemm(x.numpy(), qw.numpy(), qz.numpy(), s.numpy(), g, bias.numpy())
    W = oracle.dequant_gemm(qw.numpy(), qz.numpy(), s.numpy(), g)
    assert_product_close(acc.cpu().numpy().astype(np.float64), y32, f"row-parallel tp{world}",
                         wsigma=oracle.weight_rounding_sigma(x.numpy(), W) + world * np.abs(y32) * 2.0 ** -11)
    assert (acc.half().float() - y_full.float()).abs().max() <= 4e-3 * y_full.float().abs().max()
    cols = [ColumnParallelWQLinear(full, r, world)(x.cuda()) for r in range(world)]
    # same columns, but a narrower shard may pick another K split: equal up to the fp32 summation order
    yc = torch.cat(cols, dim=-1).float()
    assert (yc - y_full.float()).abs().max() <= 2 * 2.0 ** -10 * y_full.float().abs().max()
