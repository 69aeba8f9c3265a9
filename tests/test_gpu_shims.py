"""Every `awq_ext` / `awq_v2_ext` shim entry point on the GPU kernels, called with the reference's positional
forms (the call sites are cited; the reference's own modules running over these shims are in
tests/test_route_a.py, which needs /root/reference and therefore runs in the build container only)."""
import numpy as np
import pytest
import torch

from conftest import assert_product_close, golden
from test_gpu_parity import dev, gemv_case, gemvfast_case, stacked_experts

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ext():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from autoawq_amd import _lib, awq_ext, awq_v2_ext

    _lib.lib()
    return awq_ext, awq_v2_ext


def test_gemv_entry_points(ext, oracle):
    """awq/modules/linear/gemv.py:168-180: gemmv2_forward_cuda(inputs, qweight, scales, qzeros, group_size, split_k_iters)
    above 8 rows, gemv_forward_cuda(inputs, qweight, scales, qzeros, group_size) up to 8."""
    awq_ext, _ = ext
    K, N, g = 1024, 72, 64
    for M, fn in ((3, lambda *a: awq_ext.gemv_forward_cuda(*a, g)), (8, lambda *a: awq_ext.gemv_forward_cuda(*a, g)),
                  (12, lambda *a: awq_ext.gemmv2_forward_cuda(*a, g, 8)), (40, lambda *a: awq_ext.gemmv2_forward_cuda(*a, g, 8))):
        qw, qz, sc, x = gemv_case(K, N, g, M, seed=M)
        W = oracle.dequant_gemv(qw.numpy(), qz.numpy(), sc.numpy(), g)
        y32, _ = oracle.matmul(x.numpy(), W)
        out = fn(x.cuda(), qw.cuda(), sc.cuda(), qz.cuda())
        assert out.shape == (M, N) and out.dtype == torch.float16
        assert_product_close(out.cpu().numpy().astype(np.float64), y32, f"gemv shim M={M}", wsigma=oracle.weight_rounding_sigma(x.numpy(), W))


def test_silu_and_mul_and_layernorm_write_the_callers_tensor(ext, oracle):
    """awq/modules/fused/moe.py:73-76 silu_and_mul(out, gate_up); awq/modules/fused/norm.py:33-36
    layernorm_forward_cuda(x, weight, out, eps): both fill a caller-allocated output."""
    awq_ext, _ = ext
    gen = torch.Generator().manual_seed(1)
    gu = (torch.randn((5, 2, 2 * 384), generator=gen) * 2).half()
    out = torch.full((5, 2, 384), 7.0, dtype=torch.float16, device="cuda")
    assert awq_ext.silu_and_mul(out, gu.cuda()) is None
    want = oracle.silu_and_mul(gu.numpy()).astype(np.float32)
    ulp = np.maximum(np.abs(want), 2.0 ** -14) * 2.0 ** -10
    assert (np.abs(out.cpu().numpy().astype(np.float32) - want) <= ulp).all()
    x = torch.randn((1, 6, 512), generator=gen).half()
    w = (torch.rand(512, generator=gen) + 0.5).half()
    o = torch.empty_like(x, device="cuda")
    assert awq_ext.layernorm_forward_cuda(x.cuda(), w.cuda(), o, 1e-5) is None
    xf = x.float()
    ref = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5) * w.float()
    assert torch.allclose(o.cpu().float(), ref, rtol=2e-3, atol=2e-3)


def test_moe_entry_points(ext, oracle):
    """awq/modules/fused/moe.py:121-133 moe_alig_block_size(topk_ids, num_experts, block_size, sorted_ids, expert_ids,
    num_tokens_post_pad) fills the reference's pre-allocated tensors; :60-89 grouped_gemm_forward(x, qweight, scales,
    qzeros, topk_weights, sorted_token_ids, expert_ids, num_tokens_post_padded, mul_weights, 8) with 16-row blocks."""
    awq_ext, _ = ext
    from autoawq_amd import ops

    T, E, topk, K, N, g = 6, 8, 2, 256, 512, 128
    qw, qz, sc = stacked_experts(E, K, N, g, seed=3)
    gen = torch.Generator().manual_seed(4)
    x = torch.randn((T, K), generator=gen).half()
    w, ids = ops.fused_topk(torch.randn((T, E), generator=gen).cuda(), topk, True)
    # the reference allocates exactly these (moe.py:121-128)
    sorted_ids = torch.empty((ids.numel() + E * (16 - 1),), dtype=torch.int32, device="cuda").fill_(ids.numel())
    expert_ids = torch.empty((ids.numel() + E,), dtype=torch.int32, device="cuda")
    npad = torch.empty((1,), dtype=torch.int32, device="cuda")
    assert awq_ext.moe_alig_block_size(ids, E, 16, sorted_ids, expert_ids, npad) is None
    ws, we, wn = oracle.moe_align(ids.cpu().numpy(), E, 16)
    assert int(npad) == wn and np.array_equal(sorted_ids.cpu().numpy(), ws)
    assert np.array_equal(expert_ids.cpu().numpy()[: wn // 16], we[: wn // 16])
    for mul in (False, True):
        y = awq_ext.grouped_gemm_forward(x.cuda().view(T, 1, K), qw.cuda(), sc.cuda(), qz.cuda(), w, sorted_ids, expert_ids, npad, mul, 8)
        assert y.shape == (T, topk, N)
        idc, wc = ids.cpu().numpy(), w.cpu().numpy()
        for t in range(T):
            for j in range(topk):
                e = int(idc[t, j])
                ref32, _ = oracle.linear_gemm(x[t:t + 1].numpy(), qw[e].numpy(), qz[e].numpy(), sc[e].numpy(), g)
                sig = oracle.weight_rounding_sigma(x[t:t + 1].numpy(), oracle.dequant_gemm(qw[e].numpy(), qz[e].numpy(), sc[e].numpy(), g))
                k = wc[t, j] if mul else 1.0
                assert_product_close(y[t, j].cpu().numpy().astype(np.float64)[None], ref32 * k, f"pair {t},{j} mul={mul}", wsigma=sig * k)


@pytest.mark.parametrize("K,N,g", [(512, 64, 128), (1024, 80, 64), (512, 48, 32)])
def test_awq_v2_ext_entry_points(ext, oracle, K, N, g):
    """awq/modules/linear/gemv_fast.py:191-206: gemv_forward_cuda_decode(inputs, qweight, scales, qzeros, m, n, k,
    group_size) for batch < 8, one token; gemm_forward_cuda_prefill(inputs, qweight, scales, qzeros) otherwise -- which
    is NOT told the group size: g = 64 / 32 modules must come out right too."""
    _, awq_v2_ext = ext
    qw, sc, qz, _ = gemvfast_case(K, N, g, 1, seed=K + g)
    W = oracle.dequant_gemvfast(qw.numpy(), sc.numpy(), qz.numpy(), g)
    gen = torch.Generator().manual_seed(g)
    xd = torch.randn((5, 1, K), generator=gen).half()
    out = awq_v2_ext.gemv_forward_cuda_decode(xd.cuda(), qw.cuda(), sc.cuda(), qz.cuda(), 5, N, K, g)
    y32, _ = oracle.matmul(xd.reshape(5, K).numpy(), W)
    assert out.shape == (5, 1, N)
    assert_product_close(out[:, 0].cpu().numpy().astype(np.float64), y32, "v2 decode", wsigma=oracle.weight_rounding_sigma(xd.reshape(5, K).numpy(), W))
    for shape in ((1, 20, K), (2, 100, K)):  # <= 64 rows: the decode kernel in chunks; above: dequant + fp16 GEMM
        xp = torch.randn(shape, generator=gen).half()
        op = awq_v2_ext.gemm_forward_cuda_prefill(xp.cuda(), qw.cuda(), sc.cuda(), qz.cuda())
        yp, _ = oracle.matmul(xp.reshape(-1, K).numpy(), W)
        assert op.shape == shape[:-1] + (N,)
        assert_product_close(op.reshape(-1, N).cpu().numpy().astype(np.float64), yp, f"v2 prefill {shape}",
                             wsigma=oracle.weight_rounding_sigma(xp.reshape(-1, K).numpy(), W) + np.abs(yp) * 2.0 ** -10)
    with pytest.raises(ValueError):
        awq_v2_ext.gemm_forward_cuda_prefill(xd.cuda(), qw.cuda(), sc[:-4].cuda().contiguous(), qz[:-4].cuda().contiguous())
