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


# ------------------------------------------------------------------ config 5: Mixtral-8x7B MoE block, bs = 4, top-2

def _rand_gemm_module(K, N, g, gen):
    from autoawq_amd import WQLinear_GEMM

    m = WQLinear_GEMM(4, g, K, N, False, "cuda")
    m.qweight = torch.randint(MIN_INT32, MAX_INT32, (K, N // 8), dtype=torch.int32, device="cuda", generator=gen)
    m.qzeros = torch.randint(MIN_INT32, MAX_INT32, (K // g, N // 8), dtype=torch.int32, device="cuda", generator=gen)
    m.scales = (torch.rand((K // g, N), device="cuda", generator=gen) * 0.02 + 0.005).half()
    return m


def test_mixtral_shape_moe_block_built_with_fuse_linears_vs_oracle(ops, oracle):
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
