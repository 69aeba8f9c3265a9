"""Parity at the shapes BASELINE.json's configs name (not toy shapes): the prefill GEMM of config 3
at M = 16384, the Mixtral-8x7B MoE block of config 5 built with fuse_linears the way the reference
builds it, all against the CPU oracle.  The decode shapes of config 2 (every Linear bench.py times,
every 1 <= M <= 16) are in test_gpu_parity.py::test_gemm_vs_oracle_all_variants."""
import numpy as np
import pytest
import torch

from conftest import assert_product_close
from test_gpu_parity import MAX_INT32, MIN_INT32, fullrange_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from autoawq_amd import _lib, ops as _ops

    _lib.lib()
    return _ops


# ------------------------------------------------------------------ config 3: prefill GEMM, M = 8 x 2048

@pytest.mark.parametrize("K,N,M,what", [
    (4096, 11008, 16384, "gate/up, 256 x 256 fat tiles"),
    (11008, 4096, 16384, "down, 256 x 256 fat tiles"),
    (4096, 11008, 2048, "gate/up, 128 x 256 fat tiles"),
    (4096, 4096, 16384, "q/k/v/o, 256 x 256 fat tiles"),
])
def test_fused_prefill_gemm_at_config3_shape_vs_oracle(ops, oracle, K, N, M, what):
    """The fused dequant + MFMA kernel at the full config-3 problem; the oracle checks a seeded sample of
    128 whole rows (it needs seconds per 128 rows), the rest of the output is held to the HIP dequant +
    fp32 product of the same weights.  Also the route the module takes at this size (bit-exact HIP dequant
    + vendor fp16 GEMM) on the same sample."""
    qw, qz, s, _, bias = fullrange_case(K, N, 128, 1, seed=K + N + M, realistic=True)
    gen = torch.Generator().manual_seed(M)
    x = torch.randn((M, K), generator=gen).half()
    dq, ds, dz, db, dx = qw.cuda(), s.cuda(), qz.cuda(), bias.cuda(), x.cuda()
    y = ops.gemm_forward(dx, dq, ds, dz, db, flags=ops.gemm_flags(ops.KERNEL_TILED, nlog=2))
    assert ops.last_kernel() == "gemm_tiled"
    rows = torch.randperm(M, generator=gen)[:128].sort().values
    # the tile corners are the ragged places: always include the first and last rows of the problem
    rows[0], rows[-1] = 0, M - 1
    y32, _ = oracle.linear_gemm(x[rows].numpy(), qw.numpy(), qz.numpy(), s.numpy(), 128, bias.numpy())
    assert_product_close(y[rows.cuda()].cpu().numpy().astype(np.float64), y32, f"fused tiled {what}")
    # every row: against the fp32 product of the bit-exact dequantised weights, in row chunks
    W = ops.dequantize_weights(dq, ds, dz).float()
    worst = 0.0
    for m0 in range(0, M, 2048):
        ref = dx[m0:m0 + 2048].float() @ W + db.float()
        err = (y[m0:m0 + 2048].float() - ref).abs()
        tol = 2e-3 * ref.abs() + 2e-3 * ref.abs().mean()
        assert bool((err <= tol).all()), (what, m0, float(err.max()))
        worst = max(worst, float((err / tol).max()))
    assert worst <= 1.0
    del W
    from autoawq_amd import WQLinear_GEMM

    mod = WQLinear_GEMM(4, 128, K, N, True, "cuda")
    mod.qweight, mod.qzeros, mod.scales, mod.bias = dq, dz, ds, db
    ym = mod(dx)
    assert_product_close(ym[rows.cuda()].cpu().numpy().astype(np.float64), y32, f"module route {what}")


# ------------------------------------------------------------------ config 3: the kernel that actually runs there

@pytest.mark.parametrize("K,N", [(4096, 11008), (11008, 4096)])
@pytest.mark.parametrize("bm", [1, 2])
def test_regb_prefill_kernel_directly_at_config3_shape(ops, oracle, K, N, bm):
    """csrc/gemm_regb.hip FORCED (128- and 256-row tiles) at M = 16384: every output element against the fp32 product of
    the bit-exact dequantised weights, 128 sampled rows (first and last included) against the CPU oracle, and AUTO takes
    this kernel at this size (VERDICT r02 item 1 ii)."""
    M = 16384
    qw, qz, s, _, bias = fullrange_case(K, N, 128, 1, seed=K + 7 * N + bm, realistic=True)
    gen = torch.Generator().manual_seed(K + bm)
    x = torch.randn((M, K), generator=gen).half()
    dq, ds, dz, db, dx = qw.cuda(), s.cuda(), qz.cuda(), bias.cuda(), x.cuda()
    y = ops.gemm_forward(dx, dq, ds, dz, db, flags=ops.gemm_flags(ops.KERNEL_REGB, nlog=bm))
    assert ops.last_kernel() == "gemm_regb"
    assert ops.auto_kernel(M, K, N, 128) == ops.KERNEL_REGB
    rows = torch.randperm(M, generator=gen)[:128].sort().values
    rows[0], rows[-1] = 0, M - 1
    y32, _ = oracle.linear_gemm(x[rows].numpy(), qw.numpy(), qz.numpy(), s.numpy(), 128, bias.numpy())
    assert_product_close(y[rows.cuda()].cpu().numpy().astype(np.float64), y32, f"regb bm{bm} {K}x{N} M{M}")
    W = ops.dequantize_weights(dq, ds, dz).float()
    for m0 in range(0, M, 2048):
        ref = dx[m0:m0 + 2048].float() @ W + db.float()
        err = (y[m0:m0 + 2048].float() - ref).abs()
        tol = 2e-3 * ref.abs() + 2e-3 * ref.abs().mean()
        assert bool((err <= tol).all()), (K, N, bm, m0, float(err.max()))
    assert torch.equal(y, ops.gemm_forward(dx, dq, ds, dz, db, flags=ops.gemm_flags(ops.KERNEL_REGB, nlog=bm))), "not reproducible"


@pytest.mark.gpu
@pytest.mark.parametrize("K,N,M", [(4096, 11008, 16384), (11008, 4096, 2048), (4096, 12288, 130), (4096, 4096, 97), (1024, 200, 300),
                                   (512, 64, 70), (3584, 8192, 257), (8192, 1280, 128)])
@pytest.mark.parametrize("bm", [1, 2])
def test_gemvfast_layout_prefill_route_vs_oracle_unpinned_in_the_reference(ops, oracle, K, N, M, bm):
    """Round 6: the prefill route of WQLinear_GEMVFast (the reference runs awq_v2_ext.gemm_forward_cuda_prefill there,
    awq/modules/linear/gemv_fast.py:203-206) as TWO hand-written launches -- csrc/repack.hip (GEMVFast words -> GEMM-layout words,
    bit-exact against utils/convert.py's torch unpack) + csrc/gemm_regb.hip in its FZ form (W = fp16(w s + qzeros), this format's own
    scales / fp16 zero terms).  BASELINE configs[2]'s shape at M = 16384, the transposed shape, ragged M / N, 70B shard shapes: sampled
    rows against the CPU oracle (oracle.dequant_gemvfast: unpinned in the reference itself -- the arithmetic lives in autoawq-kernels),
    every output against the fp32 product of the bit-exact dequantised weights, one-hot rows return rows of W bit for bit, bitwise
    reproducible; the module and the awq_v2_ext shim take this route by default and no torch.matmul is reached."""
    from test_gpu_parity import gemvfast_case
    from autoawq_amd.utils.convert import _unpack_fast, _unpack_rows

    g = 128
    qw, sc, qz, _ = gemvfast_case(K, N, g, 1, seed=K + 7 * N + M)
    gen = torch.Generator().manual_seed(K + M + bm)
    x = torch.randn((M, K), generator=gen).half()
    dq, ds, dz, dx = qw.cuda(), sc.cuda(), qz.cuda(), x.cuda()
    # the repack alone: the integers of the GEMM-layout words == the integers of the GEMVFast words, transposed
    kn = ops.repack_gemvfast_to_gemm(dq)
    assert kn.shape == (K, N // 8) and kn.dtype == torch.int32
    assert torch.equal(_unpack_rows(kn.cpu(), [0, 2, 4, 6, 1, 3, 5, 7]), _unpack_fast(qw).t().contiguous()), "repack is not bit-exact"
    fl = ops.gemm_flags(nlog=bm)
    y = ops.gemv_fast_prefill(dx, dq, ds, dz, g, flags=fl)
    assert ops.last_kernel() == "repack_fast+gemm_regb_fz" and y.shape == (M, N)
    W = oracle.dequant_gemvfast(qw.numpy(), sc.numpy(), qz.numpy(), g)  # [K, N] fp16
    rows = torch.randperm(M, generator=gen)[:min(M, 96)].sort().values
    rows[0], rows[-1] = 0, M - 1
    y32, _ = oracle.matmul(x[rows].numpy(), W)
    assert_product_close(y[rows.cuda()].cpu().numpy().astype(np.float64), y32, f"gemvfast prefill bm{bm} {K}x{N} M{M}")
    Wt = ops.dequantize_weights_gemv_fast(dq, ds, dz, g)
    assert np.array_equal(Wt.cpu().numpy().view(np.uint16), np.ascontiguousarray(W.T).view(np.uint16))
    Wf = Wt.float().t().contiguous()
    for m0 in range(0, M, 2048):
        ref = dx[m0:m0 + 2048].float() @ Wf
        err = (y[m0:m0 + 2048].float() - ref).abs()
        tol = 2e-3 * ref.abs() + 2e-3 * ref.abs().mean()
        assert bool((err <= tol).all()), (K, N, bm, m0, float(err.max()))
    assert torch.equal(y, ops.gemv_fast_prefill(dx, dq, ds, dz, g, flags=fl)), "not reproducible"
    rowsel = min(M, 256)
    e = torch.zeros((rowsel, K), dtype=torch.float16, device="cuda")
    ks = (torch.arange(rowsel, device="cuda") * 61 + 17) % K
    e[torch.arange(rowsel, device="cuda"), ks] = 1.0
    assert torch.equal(ops.gemv_fast_prefill(e, dq, ds, dz, g, flags=fl), Wt.t()[ks]), "one-hot rows must select rows of W"
    if bm == 1 and M <= 4096 and N % 16 == 0:
        import autoawq_amd.modules.linear.gemv as gemv_mod
        from autoawq_amd import WQLinear_GEMVFast, awq_v2_ext

        mod = WQLinear_GEMVFast(4, g, K, N, False, "cuda")
        mod.qweight, mod.qzeros, mod.scales = dq, dz, ds
        assert mod.PREFILL_IMPL == "auto"  # by measurement: gemv.prefill_route
        assert_product_close(mod(dx.view(1, M, K))[0][rows.cuda()].cpu().numpy().astype(np.float64), y32, f"module auto {K}x{N} M{M}")
        mod.PREFILL_IMPL = "fused"
        called = []
        orig = gemv_mod.dequant_matmul_nk
        import autoawq_amd.modules.linear.gemv_fast as fast_mod
        fast_mod.dequant_matmul_nk = lambda *a, **k: (called.append(1), orig(*a, **k))[1]
        try:
            ym = mod(dx.view(1, M, K))[0]
        finally:
            fast_mod.dequant_matmul_nk = orig
        assert not called, "the hand-written prefill route of WQLinear_GEMVFast reached the dequantise + vendor GEMM route"
        assert torch.equal(ym, ops.gemv_fast_prefill(dx, dq, ds, dz, g)) or ops.last_kernel() in ("gemv_batch_fast", "repack_fast+gemm_regb_fz")
        assert_product_close(ym[rows.cuda()].cpu().numpy().astype(np.float64), y32, f"module fused {K}x{N} M{M}")
        ys = awq_v2_ext.gemm_forward_cuda_prefill(dx.view(1, M, K), dq, ds, dz)[0]
        assert_product_close(ys[rows.cuda()].cpu().numpy().astype(np.float64), y32, f"awq_v2_ext shim {K}x{N} M{M}")
        mod.PREFILL_IMPL = "two_pass"
        assert_product_close(mod(dx.view(1, M, K))[0][rows.cuda()].cpu().numpy().astype(np.float64), y32, f"module two_pass {K}x{N} M{M}")


@pytest.mark.parametrize("K,N,M", [(4096, 11008, 16384), (11008, 4096, 4096), (4096, 4096, 17), (4096, 12288, 130), (1024, 200, 300),
                                   (512, 64, 70), (3584, 8192, 257), (8192, 1280, 64)])
@pytest.mark.parametrize("bm", [1, 2])
def test_gemv_layout_prefill_kernel_vs_oracle(ops, oracle, K, N, M, bm):
    """csrc/gemm_regb.hip in its N-MAJOR form (round 4: AWQ_GEMV_KERNEL_PREFILL, explicit; WQLinear_GEMV.PREFILL_IMPL = "fused") on the
    GEMV layout's own buffers: config 3's shape in both orientations, ragged M / N (partial row and column tiles, N % 256 != 0,
    N % 8 != 0 at N = 200 ... N % 4 == 0), the 70B shard shapes; sampled rows against the CPU oracle, every output against the
    fp32 product of the bit-exact dequantised weights (awq_dequantize_weights_gemv); AUTO stays on the 16-row chunks of the
    decode kernels; one-hot rows select rows of W; bitwise reproducible; the module's two routes (default: dequantise + dense
    GEMM; "fused": this kernel) agree with the oracle."""
    from test_gpu_parity import gemv_case

    qw, qz, sc, _ = gemv_case(K, N, 128, 1, seed=K + 5 * N + M)
    gen = torch.Generator().manual_seed(K + M + bm)
    x = torch.randn((M, K), generator=gen).half()
    dq, ds, dz, dx = qw.cuda(), sc.cuda(), qz.cuda(), x.cuda()
    fl = ops.gemm_flags(kernel=ops.GEMV_KERNEL_PREFILL, nlog=bm)
    y = ops.gemv_forward(dx, dq, ds, dz, 128, flags=fl)
    assert ops.last_kernel() == "gemm_regb_nk" and y.shape == (M, N)
    if M <= 300 and bm == 1:
        ya = ops.gemv_forward(dx, dq, ds, dz, 128)      # AUTO: the decode kernels, 16 rows per launch
        assert ops.last_kernel() in ("gemv_nk", "gemv_lds", "gemv_rows", "gemv_batch")
        assert float((ya.float() - y.float()).abs().max()) <= 2e-2 * float(y.float().abs().max())
    W = oracle.dequant_gemv(qw.numpy(), qz.numpy(), sc.numpy(), 128)     # [K, N] fp16, the reference's rounding
    rows = torch.randperm(M, generator=gen)[:min(M, 96)].sort().values
    rows[0], rows[-1] = 0, M - 1
    y32, _ = oracle.matmul(x[rows].numpy(), W)
    assert_product_close(y[rows.cuda()].cpu().numpy().astype(np.float64), y32, f"gemv-layout prefill bm{bm} {K}x{N} M{M}")
    Wt = ops.dequantize_weights_gemv(dq, ds, dz, 128)
    assert np.array_equal(Wt.cpu().numpy().view(np.uint16), np.ascontiguousarray(W.T).view(np.uint16))
    Wf = Wt.float().t().contiguous()
    for m0 in range(0, M, 2048):
        ref = dx[m0:m0 + 2048].float() @ Wf
        err = (y[m0:m0 + 2048].float() - ref).abs()
        tol = 2e-3 * ref.abs() + 2e-3 * ref.abs().mean()
        assert bool((err <= tol).all()), (K, N, bm, m0, float(err.max()))
    assert torch.equal(y, ops.gemv_forward(dx, dq, ds, dz, 128, flags=fl)), "not reproducible"
    rowsel = min(M, 256)
    e = torch.zeros((rowsel, K), dtype=torch.float16, device="cuda")
    ks = (torch.arange(rowsel, device="cuda") * 61 + 17) % K
    e[torch.arange(rowsel, device="cuda"), ks] = 1.0
    if rowsel >= 17:
        assert torch.equal(ops.gemv_forward(e, dq, ds, dz, 128, flags=fl), Wt.t()[ks]), "one-hot rows must select rows of W"
    if bm == 1 and M <= 4096:
        from autoawq_amd import WQLinear_GEMV

        mod = WQLinear_GEMV(4, 128, K, N, False, "cuda")
        mod.qweight, mod.qzeros, mod.scales = dq, dz, ds
        from autoawq_amd.modules.linear.gemv import prefill_min_rows

        PREFILL_MIN_ROWS = prefill_min_rows(K)

        # the module's routes: below PREFILL_MIN_ROWS the batched-decode kernel; from there "repack" (default: csrc/repack.hip + the
        # fused MFMA GEMM on the temporary), "two_pass" (dequantise + dense GEMM) or "fused" (this kernel)
        assert mod.PREFILL_IMPL == "auto"  # by measurement: gemv.prefill_route
        assert_product_close(mod(dx)[rows.cuda()].cpu().numpy().astype(np.float64), y32, f"module auto {K}x{N} M{M}")
        mod.PREFILL_IMPL = "repack"
        assert_product_close(mod(dx)[rows.cuda()].cpu().numpy().astype(np.float64), y32, f"module default {K}x{N} M{M}")
        assert ops.last_kernel() == "gemv_batch" if M < PREFILL_MIN_ROWS else ops.last_kernel() in ("gemm_regb", "gemm_tiled"), ops.last_kernel()
        mod.PREFILL_IMPL = "two_pass"
        assert_product_close(mod(dx)[rows.cuda()].cpu().numpy().astype(np.float64), y32, f"module two-pass {K}x{N} M{M}")
        mod.PREFILL_IMPL = "fused"
        if M >= PREFILL_MIN_ROWS:
            assert torch.equal(mod(dx), y) and ops.last_kernel() == "gemm_regb_nk"


def test_gemv_layout_prefill_kernel_group_sizes_and_refusals(ops, oracle):
    """group sizes 64 and K (one group), and what the kernel refuses (group_size 32, K % 64): the wrapper then runs the 16-row
    decode kernels in chunks, or the module its dequantise + GEMM route -- every valid tensor still has a forward."""
    from autoawq_amd import _lib
    from test_gpu_parity import gemv_case

    pre = ops.gemm_flags(kernel=ops.GEMV_KERNEL_PREFILL)
    for K, N, g, M in [(1024, 256, 64, 40), (2048, 512, 2048, 33), (1024, 72, 64, 20)]:
        qw, qz, sc, x = gemv_case(K, N, g, M, seed=K + N + g)
        y = ops.gemv_forward(x.cuda(), qw.cuda(), sc.cuda(), qz.cuda(), g, flags=pre)
        assert ops.last_kernel() == "gemm_regb_nk", (K, N, g)
        W = oracle.dequant_gemv(qw.numpy(), qz.numpy(), sc.numpy(), g)
        y32, _ = oracle.matmul(x.numpy(), W)
        assert_product_close(y.cpu().numpy().astype(np.float64), y32, f"prefill g{g} {K}x{N} M{M}")
    qw, qz, sc, x = gemv_case(512, 40, 32, 33, seed=9)
    with pytest.raises(_lib.AwqHipError) as ei:
        ops.gemv_forward(x.cuda(), qw.cuda(), sc.cuda(), qz.cuda(), 32, flags=ops.gemm_flags(kernel=ops.GEMV_KERNEL_PREFILL))
    assert ei.value.code == _lib.ERR_UNSUPPORTED
    y = ops.gemv_forward(x.cuda(), qw.cuda(), sc.cuda(), qz.cuda(), 32)     # AUTO: the 16-row chunks
    assert ops.last_kernel() == "gemv_nk"
    W = oracle.dequant_gemv(qw.numpy(), qz.numpy(), sc.numpy(), 32)
    y32, _ = oracle.matmul(x.numpy(), W)
    assert_product_close(y.cpu().numpy().astype(np.float64), y32, "g32 falls through to the tile kernel",
                         wsigma=oracle.weight_rounding_sigma(x.numpy(), W))


# ------------------------------------------------------------------ the batched-decode configurations bench.py times

@pytest.mark.parametrize("K,N", [(4096, 11008), (4096, 22016), (4096, 28672)])
@pytest.mark.parametrize("M", [17, 32, 33, 48, 64])
def test_batched_decode_at_benched_shapes_vs_oracle(ops, oracle, K, N, M):
    """csrc/gemm_skinny.hip at the widths `gemm_bs` times (4096 x 11008) and at the fused gate|up widths (22016, and 28672
    of the 70B model): AUTO and forced K splits against the oracle, the one-hot row check.  Above 32 rows the kernel has four
    reducer blocks per tile which must all be resident: N > 16384 is refused there (344 reducers on 256 CUs spun
    until the give-up) and AUTO takes the LDS-tiled kernel, checked here as well."""
    qw, qz, s, x, bias = fullrange_case(K, N, 128, M, seed=K + N + M, realistic=True)
    y32, _ = oracle.linear_gemm(x.numpy(), qw.numpy(), qz.numpy(), s.numpy(), 128, bias.numpy())
    dq, ds, dz, db, dx = qw.cuda(), s.cuda(), qz.cuda(), bias.cuda(), x.cuda()
    takes = not (M > 32 and N > 16384)
    y = ops.gemm_forward(dx, dq, ds, dz, db)
    assert ops.last_kernel() == ("gemm_skinny" if takes else "gemm_tiled")
    assert_product_close(y.cpu().numpy().astype(np.float64), y32, f"auto {K}x{N} M{M}")
    ran = 0
    for sk in (0, 2, 4, 8):
        fl = ops.gemm_flags(ops.KERNEL_SKINNY, splitk=sk)
        try:
            y = ops.gemm_forward(dx, dq, ds, dz, db, flags=fl)
        except Exception as e:
            assert "code -3" in str(e), e
            assert not takes or sk in (2, 8), f"split {sk} refused at {K}x{N} M{M}"  # (2 < the four reducers above 32 rows; 8 x exchange)
            continue
        assert takes
        ran += 1
        assert_product_close(y.cpu().numpy().astype(np.float64), y32, f"skinny {K}x{N} M{M} s{sk}")
        assert torch.equal(y, ops.gemm_forward(dx, dq, ds, dz, db, flags=fl))
    assert ran >= (2 if takes else 0)
    ops.check_workspaces()
    assert ops.workspace_is_clean(dx.device)
    W = ops.dequantize_weights(dq, ds, dz)
    e = torch.zeros((M, K), dtype=torch.float16, device="cuda")
    ks = (torch.arange(M, device="cuda") * 61 + 17) % K
    e[torch.arange(M, device="cuda"), ks] = 1.0
    assert torch.equal(ops.gemm_forward(e, dq, ds, dz), W[ks]), "one-hot rows must select rows of W"


# ------------------------------------------------------------------ config 4: Llama-3-70B, TP = 8 shards and unsharded

@pytest.mark.parametrize("K,N,what", [(8192, 1280, "qkv shard"), (1024, 8192, "o shard"), (8192, 7168, "gate|up shard"),
                                      (3584, 8192, "down shard"), (8192, 10240, "qkv unsharded"), (28672, 8192, "down unsharded")])
@pytest.mark.parametrize("M", [1, 8])
def test_config4_shapes_vs_oracle(ops, oracle, K, N, M, what):
    """BASELINE configs[3] (Llama-3-70B, TP = 8): the per-rank shard shapes tools/bench_tp_shards.py times and the unsharded
    projections, GEMM layout AUTO + the GEMV layout's two kernels, against the oracle."""
    from autoawq_amd.utils.convert import pack_linear

    qw, qz, s, x, bias = fullrange_case(K, N, 128, M, seed=K + 3 * N + M, realistic=True)
    y32, _ = oracle.linear_gemm(x.numpy(), qw.numpy(), qz.numpy(), s.numpy(), 128, None)
    dq, ds, dz, dx = qw.cuda(), s.cuda(), qz.cuda(), x.cuda()
    W = oracle.dequant_gemm(qw.numpy(), qz.numpy(), s.numpy(), 128)
    wsig = oracle.weight_rounding_sigma(x.numpy(), W)
    y = ops.gemm_forward(dx, dq, ds, dz)
    assert_product_close(y.cpu().numpy().astype(np.float64), y32, f"gemm layout {what} M{M} ({ops.last_kernel()})", wsigma=wsig)
    # the same integers in the GEMV layout (repacked on the device, bit-exact: tests/test_checkpoint.py pins the converter)
    from autoawq_amd import WQLinear_GEMM
    from autoawq_amd.utils.convert import convert_linear

    mg = WQLinear_GEMM(4, 128, K, N, False, "cuda")
    mg.qweight, mg.qzeros, mg.scales = dq, dz, ds
    mv = convert_linear(mg, "gemv")
    Wt = ops.dequantize_weights_gemv(mv.qweight, mv.scales, mv.qzeros, 128)
    assert np.array_equal(Wt.cpu().numpy().view(np.uint16), np.ascontiguousarray(W.T).view(np.uint16))
    for flags, name in ((0, "auto"), (ops.gemm_flags(kernel=1), "tile16"), (ops.gemm_flags(kernel=2), "rows")):
        try:
            yv = ops.gemv_forward(dx, mv.qweight, mv.scales, mv.qzeros, 128, flags=flags)
        except Exception as e:
            assert name == "rows" and M > 4 and "code -3" in str(e), e
            continue
        assert_product_close(yv.cpu().numpy().astype(np.float64), y32, f"gemv layout {name} {what} M{M} ({ops.last_kernel()})", wsigma=wsig)


# ------------------------------------------------------------------ the fused 7B widths on the GEMV / GEMVFast layouts

@pytest.mark.parametrize("N", [12288, 22016])
@pytest.mark.parametrize("M", [1, 8])
def test_gemv_and_gemvfast_layouts_at_fused_widths(ops, oracle, N, M):
    """4096 -> 12288 (q|k|v) and 4096 -> 22016 (gate|up): the widths bench.py --layout gemv / gemvfast times."""
    from test_gpu_parity import gemv_case, gemvfast_case

    K, g = 4096, 128
    qw, qz, sc, x = gemv_case(K, N, g, M, seed=N + M)
    W = oracle.dequant_gemv(qw.numpy(), qz.numpy(), sc.numpy(), g)
    y32, _ = oracle.matmul(x.numpy(), W)
    wsig = oracle.weight_rounding_sigma(x.numpy(), W)
    for flags in (0, ops.gemm_flags(kernel=1)):
        y = ops.gemv_forward(x.cuda(), qw.cuda(), sc.cuda(), qz.cuda(), g, flags=flags)
        assert_product_close(y.cpu().numpy().astype(np.float64), y32, f"gemv {K}x{N} M{M} ({ops.last_kernel()})", wsigma=wsig)
    qf, sf, zf, xf = gemvfast_case(K, N, g, M, seed=N + 2 * M)
    Wf = oracle.dequant_gemvfast(qf.numpy(), sf.numpy(), zf.numpy(), g)
    yf32, _ = oracle.matmul(xf.numpy(), Wf)
    y = ops.gemv_fast_forward(xf.cuda(), qf.cuda(), sf.cuda(), zf.cuda(), g)
    assert_product_close(y.cpu().numpy().astype(np.float64), yf32, f"gemvfast {K}x{N} M{M}", wsigma=oracle.weight_rounding_sigma(xf.numpy(), Wf))


# ------------------------------------------------------------------ config 5: Mixtral-8x7B MoE block, bs = 4, top-2

def _rand_gemm_module(K, N, g, gen):
    from autoawq_amd import WQLinear_GEMM

    m = WQLinear_GEMM(4, g, K, N, False, "cuda")
    m.qweight = torch.randint(MIN_INT32, MAX_INT32, (K, N // 8), dtype=torch.int32, device="cuda", generator=gen)
    m.qzeros = torch.randint(MIN_INT32, MAX_INT32, (K // g, N // 8), dtype=torch.int32, device="cuda", generator=gen)
    m.scales = (torch.rand((K // g, N), device="cuda", generator=gen) * 0.02 + 0.005).half()
    return m


def test_mixtral_shape_moe_block_built_with_fuse_linears_vs_oracle_unpinned_in_the_reference(ops, oracle):
    """BASELINE config 5 at its real shape: E = 8, top-2, hidden 4096, intermediate 14336, 4 tokens.
    The expert stacks are built exactly like awq/models/mixtral.py:131-151 does (fuse_linears([w1, w3]) per
    expert on N, then fuse_linears(..., dim=0, operation=torch.stack) over the experts, the same for w2);
    both grouped GEMMs (K = 4096, N = 28672 and K = 14336, N = 4096) are checked pair by pair against
    the oracle, then the whole block against the oracle's restatement of apply_moe_weights
    (awq/modules/fused/moe.py:45-91)."""
    from autoawq_amd.modules.fused.moe import FusedSparseMoeBlock
    from autoawq_amd.utils.fused_utils import fuse_linears

    E, H, I, g, T, topk = 8, 4096, 14336, 128, 4, 2
    gen = torch.Generator(device="cuda").manual_seed(5)
    experts = [dict(w1=_rand_gemm_module(H, I, g, gen), w3=_rand_gemm_module(H, I, g, gen),
                    w2=_rand_gemm_module(I, H, g, gen)) for _ in range(E)]
    w1_first = experts[0]["w1"].qweight.clone()
    fused_w1w3s = [fuse_linears([e["w1"], e["w3"]], "cuda") for e in experts]
    assert not hasattr(experts[0]["w1"], "qweight")  # the sources give their buffers up (fused_utils.py:159-160)
    assert fused_w1w3s[0].qweight.shape == (H, 2 * I // 8) and torch.equal(fused_w1w3s[0].qweight[:, : I // 8], w1_first)
    ws = fuse_linears(fused_w1w3s, "cuda", dim=0, operation=torch.stack)
    w2s = fuse_linears([e["w2"] for e in experts], "cuda", dim=0, operation=torch.stack)
    assert ws.qweight.shape == (E, H, 2 * I // 8) and ws.qzeros.shape == (E, H // g, 2 * I // 8) and ws.scales.shape == (E, H // g, 2 * I)
    assert w2s.qweight.shape == (E, I, H // 8) and w2s.scales.shape == (E, I // g, H)
    del experts, fused_w1w3s
    torch.cuda.empty_cache()

    cgen = torch.Generator().manual_seed(6)
    x = torch.randn((T, H), generator=cgen).half()
    gate = torch.nn.Linear(H, E, bias=False).half()
    gate.weight.data = (torch.randn((E, H), generator=cgen) * 0.05).half()
    blk = FusedSparseMoeBlock(topk, gate.cuda(), ws, w2s)
    with torch.no_grad():
        out = blk(x.cuda().view(1, T, H))
        logits = blk.gate(x.cuda()).float()
    w, ids, s_ids, e_ids, npad = ops.moe_route(logits, topk, True, 8)

    # ---- the two grouped GEMMs, pair by pair (only the experts that were hit are copied to the host)
    gu = ops.grouped_gemm_forward(x.cuda().view(T, 1, H), ws.qweight, ws.scales, ws.qzeros, w, s_ids, e_ids, npad, False,
                                  block_rows=8)
    assert ops.last_kernel() == "gemv_mfma_grouped" and gu.shape == (T, topk, 2 * I)
    act = ops.silu_and_mul(gu)
    dn = ops.grouped_gemm_forward(act, w2s.qweight, w2s.scales, w2s.qzeros, w, s_ids, e_ids, npad, True, block_rows=8)
    idc, wc = ids.cpu().numpy(), w.cpu().numpy()
    host = {}
    for e in sorted(set(int(v) for v in idc.reshape(-1))):
        host[e] = tuple(t[e].cpu().numpy() for t in (ws.qweight, ws.qzeros, ws.scales, w2s.qweight, w2s.qzeros, w2s.scales))
    want = np.zeros((T, H), np.float32)
    for t in range(T):
        for j in range(topk):
            q1, z1, s1, q2, z2, s2 = host[int(idc[t, j])]
            r32, r16 = oracle.linear_gemm(x[t:t + 1].numpy(), q1, z1, s1, g)
            sig = oracle.weight_rounding_sigma(x[t:t + 1].numpy(), oracle.dequant_gemm(q1, z1, s1, g))
            assert_product_close(gu[t, j].cpu().numpy().astype(np.float64)[None], r32, f"w1|w3 pair {t},{j}", wsigma=sig)
            # second GEMM on the activation the DEVICE produced, so this is a check of that GEMM alone
            a16 = act[t, j].cpu().numpy()[None]
            d32, _ = oracle.linear_gemm(a16, q2, z2, s2, g)
            sig2 = oracle.weight_rounding_sigma(a16, oracle.dequant_gemm(q2, z2, s2, g))
            assert_product_close(dn[t, j].cpu().numpy().astype(np.float64)[None], d32 * wc[t, j], f"w2 pair {t},{j}",
                                 wsigma=sig2 * wc[t, j])
            # and the oracle's own chain for the block-level check below
            o32, _ = oracle.linear_gemm(oracle.silu_and_mul(r16), q2, z2, s2, g)
            want[t] += np.float16(np.float32(wc[t, j]) * o32[0]).astype(np.float32)
    got = out[0].cpu().numpy().astype(np.float64)
    w32 = want.astype(np.float16).astype(np.float64)
    rms = np.sqrt((w32 ** 2).mean())
    assert (np.abs(got - w32) <= 4e-3 * np.abs(w32) + 4e-3 * rms).all(), np.abs(got - w32).max() / rms
    assert ops.workspace_is_clean(out.device)
