"""MoE decode on GEMV-layout expert stacks (round 6): awq_grouped_gemv_forward -- every (token, expert) pair one batch-1 call of the
row-streaming kernel, all pairs in one launch -- against the oracle, against the plain per-pair launches of the same kernel
(bitwise), and through apply_moe_weights against the oracle's MoE block and the GEMM-layout grouped path.
Reference call sites: awq/modules/fused/moe.py:60-89 (grouped_gemm_forward x 2, silu_and_mul, mul_routed_weight, top-k sum);
awq/models/mixtral.py:130-158 (the stacks).  "Unpinned in the reference": the grouped kernels live in autoawq-kernels."""
import numpy as np
import pytest
import torch

from conftest import assert_close_to_exact

pytestmark = pytest.mark.gpu

MIN_INT32, MAX_INT32 = -(2 ** 31), 2 ** 31 - 1


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from autoawq_amd import _lib, ops as _ops

    _lib.lib()  # fails loudly if the extension is missing
    return _ops


def gemv_stack(E, K, N, g, seed):
    from autoawq_amd.utils.packing import calculate_zeros_width

    gen = torch.Generator().manual_seed(seed)
    zw, G = calculate_zeros_width(K, g), K // g
    qw = torch.randint(MIN_INT32, MAX_INT32, (E, N, K // 8), dtype=torch.int32, generator=gen)
    zn = torch.randint(0, 16, (E, N, zw * 8), dtype=torch.int32, generator=gen)
    zn[:, :, G:] = 0
    qz = torch.zeros((E, N, zw), dtype=torch.int32)
    for i in range(8):
        qz |= zn[:, :, i::8] << (4 * i)
    sc = torch.zeros((E, N, zw * 8), dtype=torch.float16)
    sc[:, :, :G] = (torch.rand((E, N, G), generator=gen) * 0.02 + 0.005).half()
    return qw, qz, sc


# (K, N): one per slot count of the row-streaming kernel (SL = 1, 2, 3, 4, 6, 8), ragged row counts, the Mixtral K values
SHAPES = [(256, 512), (512, 250), (1024, 96), (4096, 1024), (6144, 256), (8192, 128), (11008, 64), (14336, 256), (4096, 7168)]


@pytest.mark.parametrize("K,N", SHAPES)
@pytest.mark.parametrize("T,topk", [(1, 2), (4, 2), (5, 3), (8, 1)])
def test_grouped_rows_kernel_vs_oracle_and_vs_per_pair_launches_unpinned_in_the_reference(ops, oracle, K, N, T, topk):
    E, g = 8, 128
    qw, qz, sc = gemv_stack(E, K, N, g, seed=K + N + T)
    gen = torch.Generator().manual_seed(T * 31 + topk)
    x = torch.randn((T, K), generator=gen).half()
    ids = torch.stack([torch.randperm(E, generator=gen)[:topk] for _ in range(T)]).int()
    w = torch.rand((T, topk), generator=gen).float() + 0.25
    qwc, qzc, scc, xc, idc, wc = qw.cuda(), qz.cuda(), sc.cuda(), x.cuda(), ids.cuda(), w.cuda()
    y = ops.grouped_gemv_forward(xc, qwc, scc, qzc, idc, g)
    assert ops.last_kernel() == "gemv_rows_grouped" and y.shape == (T, topk, N)
    yw = ops.grouped_gemv_forward(xc, qwc, scc, qzc, idc, g, topk_weights=wc)
    ones = ops.grouped_gemv_forward(xc, qwc, scc, qzc, idc, g, topk_weights=torch.ones_like(wc))
    assert torch.equal(ones, y), "a routing weight of 1 must not change a bit"
    assert torch.equal(y, ops.grouped_gemv_forward(xc, qwc, scc, qzc, idc, g)), "not bitwise reproducible"
    # x with one row per pair (the w2 call's form)
    xp = x.repeat_interleave(topk, dim=0).cuda()
    assert torch.equal(ops.grouped_gemv_forward(xp, qwc, scc, qzc, idc, g), y)
    rows = ops.gemm_flags(kernel=2)  # AWQ_GEMV_KERNEL_ROWS
    for t in range(T):
        for j in range(topk):
            e = int(ids[t, j])
            yex = oracle.matmul_exact_gemv(x[t:t + 1].numpy(), qw[e].numpy(), qz[e].numpy(), sc[e].numpy(), g)
            assert_close_to_exact(y[t, j].cpu().numpy()[None], yex, f"pair ({t},{j}) expert {e} K{K} N{N}")
            assert_close_to_exact(yw[t, j].cpu().numpy()[None], yex * float(w[t, j]), f"weighted pair ({t},{j}) K{K} N{N}")
            one = ops.gemv_forward(xc[t:t + 1], qwc[e], scc[e], qzc[e], g, flags=rows)
            assert ops.last_kernel() == "gemv_rows"
            assert torch.equal(one[0], y[t, j]), f"pair ({t},{j}): the grouped launch differs from the plain launch of the same kernel"
    # forced numbers of parts (ragged against the XCD map, more parts than super-units)
    for parts in (1, 8, 24, 100, 1000):
        try:
            yp = ops.grouped_gemv_forward(xc, qwc, scc, qzc, idc, g, parts=parts)
        except Exception as e:  # too few parts for the block's LDS (every row a block produces parks its partial sums there)
            assert "code -3" in str(e) and parts < 24, e
            continue
        assert torch.equal(yp, y), f"parts={parts}"


@pytest.mark.parametrize("K,N", [(256, 512), (4096, 2048), (6144, 256), (14336, 128)])
def test_grouped_rows_silu_pairs_equals_the_plain_kernels_epilogue(ops, K, N):
    E, g, T, topk = 8, 128, 4, 2
    qw, qz, sc = gemv_stack(E, K, N, g, seed=3 * K + N)
    gen = torch.Generator().manual_seed(11)
    x = (torch.randn((T, K), generator=gen) * 0.5).half().cuda()
    ids = torch.stack([torch.randperm(E, generator=gen)[:topk] for _ in range(T)]).int().cuda()
    qwc, qzc, scc = qw.cuda(), qz.cuda(), sc.cuda()
    act = ops.grouped_gemv_forward(x, qwc, scc, qzc, ids, g, silu_pairs=True)
    assert act.shape == (T, topk, N // 2)
    full = ops.grouped_gemv_forward(x, qwc, scc, qzc, ids, g)
    # == awq_silu_and_mul on the unfused rows, with (gate, up) de-interleaved
    gate_up = torch.cat([full[..., 0::2], full[..., 1::2]], dim=-1).contiguous()
    assert torch.equal(act, ops.silu_and_mul(gate_up.view(-1, N)).view(T, topk, N // 2))
    for t in range(T):
        for j in range(topk):
            e = int(ids[t, j])
            one = ops.gemv_forward_ex(x[t:t + 1], qwc[e], scc[e], qzc[e], g, silu_pairs=True)
            assert torch.equal(one[0], act[t, j])


def test_grouped_rows_skips_pairs_of_foreign_experts_and_rejects_bad_calls(ops):
    E, g, K, N, T, topk = 4, 128, 512, 256, 3, 2
    qw, qz, sc = gemv_stack(E, K, N, g, seed=5)
    x = torch.randn((T, K)).half().cuda()
    ids = torch.tensor([[0, 3], [-1, 2], [7, 1]], dtype=torch.int32).cuda()  # -1 and 7: not experts of this stack
    y = ops.grouped_gemv_forward(x, qw.cuda(), sc.cuda(), qz.cuda(), ids, g, zero_init=True)
    assert int(y[1, 0].abs().max()) == 0 and int(y[2, 0].abs().max()) == 0
    assert float(y[0, 0].abs().max()) > 0 and float(y[1, 1].abs().max()) > 0 and float(y[2, 1].abs().max()) > 0
    with pytest.raises(Exception):
        ops.grouped_gemv_forward(x[:, :256].contiguous(), qw.cuda(), sc.cuda(), qz.cuda(), ids, g)
    with pytest.raises(Exception):  # silu pairs and a routing weight do not go together (moe.py:73-89: w1 has none)
        ops.grouped_gemv_forward(x, qw.cuda(), sc.cuda(), qz.cuda(), ids, g, topk_weights=torch.ones((T, topk), device="cuda"),
                                 silu_pairs=True)


class Stack:
    pass


def gemm_stacks(E, H, I, g, seed):
    gen = torch.Generator().manual_seed(seed)

    def one(K, N):
        s = Stack()
        s.qweight = torch.randint(MIN_INT32, MAX_INT32, (E, K, N // 8), dtype=torch.int32, generator=gen)
        s.qzeros = torch.randint(MIN_INT32, MAX_INT32, (E, K // g, N // 8), dtype=torch.int32, generator=gen)
        s.scales = (torch.rand((E, K // g, N), generator=gen) * 0.02 + 0.005).half()
        return s
    return one(H, 2 * I), one(I, H)


def test_gemm_stack_to_gemv_is_the_same_weights(ops, oracle):
    """the twin's dequantised rows == the GEMM-layout experts' dequantised columns, bit for bit; w1's rows interleaved (gate_j, up_j)"""
    from autoawq_amd.utils.convert import gemm_stack_to_gemv

    E, H, I, g = 3, 256, 384, 128
    w1, w2 = gemm_stacks(E, H, I, g, seed=9)
    t1 = gemm_stack_to_gemv(w1.qweight.cuda(), w1.qzeros.cuda(), w1.scales.cuda(), interleave_halves=True)
    t2 = gemm_stack_to_gemv(w2.qweight.cuda(), w2.qzeros.cuda(), w2.scales.cuda())
    for e in range(E):
        W1 = oracle.dequant_gemm(w1.qweight[e].numpy(), w1.qzeros[e].numpy(), w1.scales[e].numpy(), g)  # [K, N]
        got = ops.dequantize_weights_gemv(t1.qweight[e], t1.scales[e], t1.qzeros[e], g).cpu().numpy()    # [N, K]
        want = np.stack([W1[:, :I].T, W1[:, I:].T], axis=1).reshape(2 * I, H)
        assert np.array_equal(got.view(np.uint16), np.ascontiguousarray(want).view(np.uint16))
        W2 = oracle.dequant_gemm(w2.qweight[e].numpy(), w2.qzeros[e].numpy(), w2.scales[e].numpy(), g)
        got2 = ops.dequantize_weights_gemv(t2.qweight[e], t2.scales[e], t2.qzeros[e], g).cpu().numpy()
        assert np.array_equal(got2.view(np.uint16), np.ascontiguousarray(W2.T).view(np.uint16))


@pytest.mark.parametrize("T", [1, 4, 6])
def test_moe_block_on_decode_twins_vs_oracle_and_vs_the_gemm_layout_path_unpinned_in_the_reference(ops, oracle, T):
    from autoawq_amd.modules.fused import moe

    E, H, I, g, topk = 8, 512, 768, 128, 2
    w1, w2 = gemm_stacks(E, H, I, g, seed=31)
    a, b = Stack(), Stack()
    for dst, src in ((a, w1), (b, w2)):
        dst.qweight, dst.qzeros, dst.scales = src.qweight.cuda(), src.qzeros.cuda(), src.scales.cuda()
    gen = torch.Generator().manual_seed(T)
    x = torch.randn((T, H), generator=gen).half()
    logits = torch.randn((T, E), generator=gen)
    base = moe.apply_moe_weights(a, b, x.cuda(), logits.cuda(), topk, True)
    assert ops.last_kernel() == "gemv_mfma_grouped"
    moe.build_decode_twins(a, b)
    got = moe.apply_moe_weights(a, b, x.cuda(), logits.cuda(), topk, True)
    assert ops.last_kernel() == "gemv_rows_grouped"
    want, _, _ = oracle.moe_forward(x.numpy(), logits.numpy(), dict(qweight=w1.qweight.numpy(), qzeros=w1.qzeros.numpy(), scales=w1.scales.numpy()),
                                    dict(qweight=w2.qweight.numpy(), qzeros=w2.qzeros.numpy(), scales=w2.scales.numpy()), topk, g)
    w64 = want.astype(np.float64)
    rms = np.sqrt((w64 ** 2).mean())
    for name, y in (("twins", got), ("gemm layout", base)):
        err = np.abs(y.cpu().numpy().astype(np.float64) - w64)
        assert (err <= 4e-3 * np.abs(w64) + 4e-3 * rms).all(), (name, err.max() / rms)
    # captured: nothing about the routing is read back
    xs, ls = x.cuda(), logits.cuda()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        moe.apply_moe_weights(a, b, xs, ls, topk, True)
        st.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            out = moe.apply_moe_weights(a, b, xs, ls, topk, True)
        gr.replay()
        st.synchronize()
    assert torch.equal(out, got)
    # beyond ROWS_MAX_PAIRS the GEMM-layout grouped kernel keeps serving
    saved = moe.ROWS_MAX_PAIRS
    try:
        moe.ROWS_MAX_PAIRS = 0
        moe.apply_moe_weights(a, b, x.cuda(), logits.cuda(), topk, True)
        assert ops.last_kernel() == "gemv_mfma_grouped"
    finally:
        moe.ROWS_MAX_PAIRS = saved


# ---------------------------------------------------------------- prefill-sized token counts: the sort as an index list (round 6)

@pytest.mark.parametrize("P,E", [(5, 8), (1024, 8), (3000, 8), (2049, 64), (64, 3)])
def test_moe_sort_pairs_equals_stable_argsort(ops, P, E):
    gen = torch.Generator().manual_seed(P + E)
    ids = torch.randint(0, E, (P,), generator=gen).int()
    if P > 100:
        ids[::37] = -1      # pairs of foreign experts (expert-parallel shards) are not placed
        ids[5::91] = E + 3
    order, seg = ops.moe_sort_pairs(ids.cuda().view(-1, 1), E)
    valid = (ids >= 0) & (ids < E)
    key = torch.where(valid, ids, torch.full_like(ids, E)).long()
    want = torch.argsort(key, stable=True)[: int(valid.sum())].int()
    counts = torch.bincount(ids[valid].long(), minlength=E)
    wseg = torch.zeros(E + 1, dtype=torch.int32)
    wseg[1:] = torch.cumsum(counts, 0).int()
    assert torch.equal(seg.cpu(), wseg)
    assert torch.equal(order.cpu()[: int(valid.sum())], want)


def test_moe_route_routing_only_many_tokens(ops):
    T, E, k = 1537, 8, 2
    logits = torch.randn((T, E), generator=torch.Generator().manual_seed(3)) * 2
    w, ids, a, b, c = ops.moe_route(logits.cuda(), k, True, 0)
    assert a is None and b is None and c is None
    tw, ti = ops.fused_topk(logits.cuda(), k, True)
    assert torch.equal(ids, ti) and torch.allclose(w, tw, rtol=1e-5, atol=1e-7)


def test_grouped_prefill_gather_scatter_and_weights(ops):
    """awq_grouped_gemm_prefill_ex against awq_grouped_gemm_prefill on explicitly gathered / scattered tensors: the same launch,
    the sort kept as an index list -- bit for bit; routing weights within one fp16 rounding of fp16(product) * w."""
    T, E, topk, K, N, g = 300, 8, 2, 512, 768, 128
    gen = torch.Generator().manual_seed(17)
    qw = torch.randint(MIN_INT32, MAX_INT32, (E, K, N // 8), dtype=torch.int32, generator=gen).cuda()
    qz = torch.randint(MIN_INT32, MAX_INT32, (E, K // g, N // 8), dtype=torch.int32, generator=gen).cuda()
    sc = (torch.rand((E, K // g, N), generator=gen) * 0.02 + 0.005).half().cuda()
    x = torch.randn((T, K), generator=gen).half().cuda()
    ids = torch.stack([torch.randperm(E, generator=gen)[:topk] for _ in range(T)]).int()
    ids[:40, 0] = 5  # one crowded expert (several row tiles), and expert 6 emptied
    ids[ids == 6] = 7
    for t in range(T):
        if ids[t, 0] == ids[t, 1]:
            ids[t, 1] = (int(ids[t, 0]) + 1) % 6
    ids = ids.cuda()
    w = (torch.rand((T, topk), generator=gen) + 0.25).float().cuda()
    order, seg = ops.moe_sort_pairs(ids, E)
    P = T * topk
    xs = x.index_select(0, (order // topk).long())
    base = ops.grouped_gemm_prefill(xs, qw, sc, qz, seg)
    got = ops.grouped_gemm_prefill_ex(x, qw, sc, qz, seg, order, x_div=topk, gather=True)
    assert ops.last_kernel() == "gemm_regb_grouped" and torch.equal(got, base)
    sorted_in = ops.grouped_gemm_prefill_ex(xs, qw, sc, qz, seg, order)
    assert torch.equal(sorted_in, base)
    scat = ops.grouped_gemm_prefill_ex(xs, qw, sc, qz, seg, order, scatter=True)
    want = torch.empty_like(base)
    want.index_copy_(0, order.long(), base)
    assert torch.equal(scat, want)
    sw = ops.grouped_gemm_prefill_ex(xs, qw, sc, qz, seg, order, scatter=True, pair_weights=w)
    ref = want.float() * w.view(-1)[:, None]
    assert ((sw.float() - ref).abs() <= 2.0 ** -10 * ref.abs() + 2.0 ** -24).all()
    ones = ops.grouped_gemm_prefill_ex(xs, qw, sc, qz, seg, order, scatter=True, pair_weights=torch.ones_like(w))
    assert torch.equal(ones, want)


def test_moe_prefill_fused_glue_matches_the_torch_glue(ops):
    from autoawq_amd.modules.fused import moe

    T, E, H, I, g, topk = 512, 8, 512, 768, 128, 2
    w1, w2 = gemm_stacks(E, H, I, g, seed=41)
    a, b = Stack(), Stack()
    for dst, src in ((a, w1), (b, w2)):
        dst.qweight, dst.qzeros, dst.scales = src.qweight.cuda(), src.qzeros.cuda(), src.scales.cuda()
    gen = torch.Generator().manual_seed(8)
    x = torch.randn((T, H), generator=gen).half().cuda()
    logits = torch.randn((T, E), generator=gen).cuda()
    saved = moe.FUSED_PREFILL_GLUE
    try:
        moe.FUSED_PREFILL_GLUE = True
        new = moe.apply_moe_weights(a, b, x, logits, topk, True)
        moe.FUSED_PREFILL_GLUE = False
        old = moe.apply_moe_weights(a, b, x, logits, topk, True)
    finally:
        moe.FUSED_PREFILL_GLUE = saved
    # the same launches on the same rows; the routing weight now multiplies the fp32 product (one rounding) instead of its fp16
    d = (new.float() - old.float()).abs()
    # (each of a token's topk rows may move by one fp16 ulp of ITS magnitude before the sum; the strong checks are the bitwise
    # kernel tests above and the oracle test of the module path)
    assert float(d.max()) <= 4e-3 * float(old.float().abs().max()), float(d.max())
