"""Route A: the UNMODIFIED reference modules over the `awq_ext` / `awq_v2_ext` shims.

Needs /root/reference, which exists only in the build container -- where there is no GPU.  So the shims'
back end (`autoawq_amd.ops`) is replaced here by the CPU oracle (test infrastructure may use it; the product
never does): what runs is the reference's own `WQLinear_GEMM / _GEMV / _GEMVFast`, `QuantFusedMLP`,
`apply_moe_weights` and `FasterTransformerRMSNorm` code, calling our shim functions with ITS positional
argument orders and return conventions -- every call site of SURVEY.md section 2.2.  The same shim functions are
exercised on the GPU kernels, with the same call forms, by tests/test_gpu_shims.py."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import golden

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "awq")), reason="the reference tree is not on this machine")


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


# ---- recording mode (tests/golden/make_route_a_calls.py sets RECORD to a list): the positional call forms of the reference, with
# their inputs and the oracle-backed results, for tests/test_gpu_route_a_replay.py -- the reference tree does not exist on the GPU box
RECORD = None


def _recording(mname, fname, fn):
    def wrapped(*args, **kwargs):
        before = [a.detach().clone() if torch.is_tensor(a) else a for a in args]
        ret = fn(*args, **kwargs)
        after = {i: a.detach().clone() for i, a in enumerate(args)
                 if torch.is_tensor(a) and (a.shape != before[i].shape or not torch.equal(a, before[i]))}
        RECORD.append(dict(mod=mname, fn=fname, args=before, kwargs=dict(kwargs), ret=ret.detach().clone() if torch.is_tensor(ret) else ret,
                           mutated=after))
        return ret
    return wrapped


@pytest.fixture()
def route_a(oracle, monkeypatch):
    """shims registered as `awq_ext` / `awq_v2_ext`, ops -> oracle, a fresh import of the reference package"""
    from autoawq_amd import awq_ext, awq_v2_ext, ops

    calls = []

    def dequantize_weights(qweight, scales, qzeros):
        calls.append("dequantize_weights")
        G = qzeros.shape[0]
        return _t(oracle.dequant_gemm(qweight.numpy(), qzeros.numpy(), scales.numpy(), qweight.shape[0] // G))

    def gemm_forward(x2d, qweight, scales, qzeros, bias=None, flags=0):
        calls.append("gemm_forward")
        g = qweight.shape[0] // qzeros.shape[0]
        _, y16 = oracle.linear_gemm(x2d.numpy(), qweight.numpy(), qzeros.numpy(), scales.numpy(), g)
        return _t(y16)

    def gemv_forward(x2d, qweight, scales, qzeros, group_size, flags=0):
        calls.append("gemv_forward")
        W = oracle.dequant_gemv(qweight.numpy(), qzeros.numpy(), scales.numpy(), group_size)
        return _t(oracle.matmul(x2d.numpy(), W)[1])

    def gemv_fast_forward(x2d, qweight, scales, qzeros, group_size, flags=0):
        calls.append("gemv_fast_forward")
        W = oracle.dequant_gemvfast(qweight.numpy(), scales.numpy(), qzeros.numpy(), group_size)
        return _t(oracle.matmul(x2d.numpy(), W)[1])

    def dequantize_weights_gemv_fast(qweight, scales, qzeros, group_size):
        calls.append("dequantize_weights_gemv_fast")
        return _t(np.ascontiguousarray(oracle.dequant_gemvfast(qweight.numpy(), scales.numpy(), qzeros.numpy(), group_size).T))

    def silu_and_mul(gate_up, out=None):
        calls.append("silu_and_mul")
        r = _t(oracle.silu_and_mul(gate_up.numpy()))
        if out is None:
            return r
        out.copy_(r)
        return out

    def rmsnorm(x, weight, eps, residual=None, out=None):
        calls.append("rmsnorm")
        xf = x.float()
        r = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps) * weight.float()).to(x.dtype)
        if out is None:
            return r
        out.copy_(r)
        return out

    def grouped_gemm_forward(x, qweight, scales, qzeros, topk_weights, sorted_token_ids, expert_ids, num_tokens_post_padded,
                             mul_weights, split_k_iters=8, block_rows=16):
        calls.append("grouped_gemm_forward")
        T, topk = topk_weights.shape
        E, K, NW = qweight.shape
        g = K // qzeros.shape[1]
        x2 = x.reshape(-1, K)
        x_div = topk if x.shape[1] == 1 else 1
        y = torch.zeros((T * topk, NW * 8), dtype=torch.float16)
        npad = int(num_tokens_post_padded.reshape(-1)[0])
        for blk in range(npad // block_rows):
            e = int(expert_ids[blk])
            for r in range(block_rows):
                pid = int(sorted_token_ids[blk * block_rows + r])
                if pid >= T * topk:
                    continue
                y32, _ = oracle.linear_gemm(x2[pid // x_div: pid // x_div + 1].numpy(), qweight[e].numpy(), qzeros[e].numpy(),
                                            scales[e].numpy(), g)
                row = _t(y32)[0]
                if mul_weights:
                    row = row * float(topk_weights.reshape(-1)[pid])
                y[pid] = row.half()
        return y.reshape(T, topk, NW * 8)

    for name, fn in dict(dequantize_weights=dequantize_weights, gemm_forward=gemm_forward, gemv_forward=gemv_forward,
                         gemv_fast_forward=gemv_fast_forward, dequantize_weights_gemv_fast=dequantize_weights_gemv_fast,
                         silu_and_mul=silu_and_mul, rmsnorm=rmsnorm, grouped_gemm_forward=grouped_gemm_forward).items():
        monkeypatch.setattr(ops, name, fn)
    saved = {k: v for k, v in sys.modules.items() if k == "awq" or k.startswith("awq.") or k in ("awq_ext", "awq_v2_ext")}
    for k in saved:
        del sys.modules[k]
    if RECORD is not None:  # tests/golden/make_route_a_calls.py: every shim call, as the UNMODIFIED reference issues it, goes into the fixture file
        for mod, mname in ((awq_ext, "awq_ext"), (awq_v2_ext, "awq_v2_ext")):
            for fname in [n for n in dir(mod) if callable(getattr(mod, n)) and not n.startswith("_") and getattr(getattr(mod, n), "__module__", "") == mod.__name__]:
                if fname in ("infer_group_size",):
                    continue
                monkeypatch.setattr(mod, fname, _recording(mname, fname, getattr(mod, fname)))
    monkeypatch.setitem(sys.modules, "awq_ext", awq_ext)
    monkeypatch.setitem(sys.modules, "awq_v2_ext", awq_v2_ext)
    monkeypatch.syspath_prepend(REF)
    monkeypatch.setattr(sys, "dont_write_bytecode", True)
    import awq.modules.linear.gemm as rg
    import awq.modules.linear.gemv as rv
    import awq.modules.linear.gemv_fast as rf

    assert rg.awq_ext is awq_ext and rv.awq_ext is awq_ext and rf.awq_v2_ext is awq_v2_ext  # what try_import found
    yield dict(gemm=rg, gemv=rv, fast=rf, calls=calls)
    for k in [k for k in sys.modules if k == "awq" or k.startswith("awq.")]:
        del sys.modules[k]
    sys.modules.update(saved)


def _close(y, ref32, ulps=4):
    y, ref32 = np.asarray(y, np.float64), np.asarray(ref32, np.float64)
    rms = float(np.sqrt(np.mean(ref32 ** 2)))
    tol = ulps * np.maximum(np.abs(ref32), 2.0 ** -14) * 2.0 ** -10 + 2e-3 * np.abs(ref32) + 2e-3 * rms
    return bool((np.abs(y - ref32) <= tol).all())


def test_reference_linear_modules_run_on_the_shims(route_a, oracle):
    g = golden("packed_K512_N64_g128")
    K, N, gs = 512, 64, 128
    x = _t(g["x"])                                   # [4, 512] fp16
    W = g["W"]
    bias = _t(g["bias"])
    y32, _ = oracle.matmul(g["x"], W, g["bias"])
    # ---- WQLinear_GEMM: decode branch (gemm_forward_cuda), >= 1024 tokens (dequantize_weights_cuda + matmul), backward
    m = route_a["gemm"].WQLinear_GEMM(4, gs, K, N, True, "cpu")
    m.qweight, m.qzeros, m.scales, m.bias = _t(g["gemm_qweight"]), _t(g["gemm_qzeros"]), _t(g["gemm_scales"]), bias
    out = m(x.view(1, 4, K))
    assert out.shape == (1, 4, N) and _close(out[0].numpy(), y32) and route_a["calls"][-1] == "gemm_forward"
    big = torch.randn((2, 600, K), generator=torch.Generator().manual_seed(0)).half()
    ob = m(big)
    yb, _ = oracle.matmul(big.reshape(-1, K).numpy(), W, g["bias"])
    assert route_a["calls"][-1] == "dequantize_weights" and _close(ob.reshape(-1, N).numpy(), yb)
    mt = route_a["gemm"].WQLinear_GEMM(4, gs, K, N, False, "cpu", training=True)
    mt.qweight, mt.qzeros, mt.scales = m.qweight, m.qzeros, m.scales
    xin = x.view(1, 4, K).clone().requires_grad_(True)
    mt(xin).float().sum().backward()                # backward: dequantize_weights_cuda(qweight, scales, qzeros, 1, 0, 0, False)
    assert _close(xin.grad[0].float().numpy(), np.broadcast_to(W.astype(np.float32).sum(1), (4, K)), ulps=16)
    # ---- WQLinear_GEMV: <= 8 rows (gemv_forward_cuda), > 8 rows (gemmv2_forward_cuda)
    v = route_a["gemv"].WQLinear_GEMV(4, gs, K, N, True, "cpu")
    v.qweight, v.qzeros, v.scales, v.bias = _t(g["gemv_qweight"]), _t(g["gemv_qzeros"]), _t(g["gemv_scales"]), bias
    assert _close(v(x).numpy(), y32) and route_a["calls"][-1] == "gemv_forward"
    x12 = torch.randn((12, K), generator=torch.Generator().manual_seed(1)).half()
    y12, _ = oracle.matmul(x12.numpy(), W, g["bias"])
    assert _close(v(x12).numpy(), y12)
    # ---- WQLinear_GEMVFast: decode (gemv_forward_cuda_decode), prefill (gemm_forward_cuda_prefill: g inferred)
    f = route_a["fast"].WQLinear_GEMVFast(4, gs, K, N, True, "cpu")
    f.qweight, f.qzeros, f.scales, f.bias = _t(g["fast_qweight"]), _t(g["fast_qzeros"]), _t(g["fast_scales"]), bias
    Wf = oracle.dequant_gemvfast(g["fast_qweight"], g["fast_scales"], g["fast_qzeros"], gs)
    yf, _ = oracle.matmul(g["x"], Wf, g["bias"])
    od = f(x.view(4, 1, K))
    assert od.shape == (4, 1, N) and _close(od[:, 0].numpy(), yf) and route_a["calls"][-1] == "gemv_fast_forward"
    op = f(x.view(1, 4, K))
    assert op.shape == (1, 4, N) and _close(op[0].numpy(), yf)
    g64 = golden("packed_K256_N32_g64")              # a g = 64 GEMVFast module: the prefill shim must not assume 128
    f64 = route_a["fast"].WQLinear_GEMVFast(4, 64, 256, 32, False, "cpu")
    f64.qweight, f64.qzeros, f64.scales = _t(g64["fast_qweight"]), _t(g64["fast_qzeros"]), _t(g64["fast_scales"])
    W64 = oracle.dequant_gemvfast(g64["fast_qweight"], g64["fast_scales"], g64["fast_qzeros"], 64)
    y64, _ = oracle.matmul(g64["x"], W64)
    assert _close(f64(_t(g64["x"]).unsqueeze(0))[0].numpy(), y64)


def test_reference_fused_mlp_moe_and_norm_run_on_the_shims(route_a, oracle):
    import awq.modules.fused.mlp as rmlp
    import awq.modules.fused.moe as rmoe
    import awq.modules.fused.norm as rnorm
    from autoawq_amd import awq_ext

    assert rmlp.AWQ_INSTALLED and rmoe.AWQ_INSTALLED and rmoe.awq_ext is awq_ext
    gen = torch.Generator().manual_seed(3)
    H, I, gs, M = 256, 384, 128, 3

    def rand_mod(K, N):
        m = route_a["gemm"].WQLinear_GEMM(4, gs, K, N, False, "cpu")
        m.qweight = torch.randint(-2 ** 31, 2 ** 31 - 1, (K, N // 8), dtype=torch.int32, generator=gen)
        m.qzeros = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // gs, N // 8), dtype=torch.int32, generator=gen)
        m.scales = (torch.rand((K // gs, N), generator=gen) * 0.02 + 0.005).half()
        return m, oracle.dequant_gemm(m.qweight.numpy(), m.qzeros.numpy(), m.scales.numpy(), gs)

    (gate, Wg), (down, Wd), (up, Wu) = rand_mod(H, I), rand_mod(I, H), rand_mod(H, I)
    mlp = rmlp.QuantFusedMLP(gate, down, up)         # self.linear = awq_ext.gemm_forward_cuda, 5th positional arg = 8
    x = torch.randn((1, M, H), generator=gen).half()
    out = mlp(x)
    g16, u16 = oracle.matmul(x[0].numpy(), Wg)[1], oracle.matmul(x[0].numpy(), Wu)[1]
    act = torch.nn.functional.silu(_t(g16)) * _t(u16)
    want, _ = oracle.matmul(act.numpy(), Wd)
    assert out.shape == (1, M, H) and _close(out[0].numpy(), want, ulps=8)
    # ---- apply_moe_weights: moe_alig_block_size fills the caller's tensors, grouped_gemm_forward x2, silu_and_mul(out, gate_up)
    E, T, topk = 4, 5, 2

    class Stack:
        pass
    w1, w2 = Stack(), Stack()
    exps = [(rand_mod(H, 2 * I), rand_mod(I, H)) for _ in range(E)]
    for dst, idx in ((w1, 0), (w2, 1)):
        dst.qweight = torch.stack([e[idx][0].qweight for e in exps])
        dst.qzeros = torch.stack([e[idx][0].qzeros for e in exps])
        dst.scales = torch.stack([e[idx][0].scales for e in exps])
    xt = torch.randn((T, H), generator=gen).half()
    logits = torch.randn((T, E), generator=gen)
    got = rmoe.apply_moe_weights(w1, w2, xt, logits, topk, renormalize=True)
    want, _, _ = oracle.moe_forward(xt.numpy(), logits.numpy(), dict(qweight=w1.qweight.numpy(), qzeros=w1.qzeros.numpy(), scales=w1.scales.numpy()),
                                    dict(qweight=w2.qweight.numpy(), qzeros=w2.qzeros.numpy(), scales=w2.scales.numpy()), topk, gs)
    assert got.shape == (T, H) and _close(got.numpy(), want.astype(np.float32), ulps=8)
    assert {"grouped_gemm_forward", "silu_and_mul"} <= set(route_a["calls"])
    # ---- FasterTransformerRMSNorm.forward: layernorm_forward_cuda(x, weight, out, eps) writes the caller's tensor
    w = (torch.rand(H, generator=gen) + 0.5).half()
    n = rnorm.FasterTransformerRMSNorm(w, 1e-5)
    if not rnorm.IPEX_INSTALLED:
        y = n(x)
        xf = x.float()
        ref = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5) * w.float())
        assert torch.allclose(y.float(), ref, rtol=2e-3, atol=2e-3) and route_a["calls"][-1] == "rmsnorm"
