"""Python face of the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
autoawq_amd/ never does (tests/test_boundary.py greps for that).

Two things live here:
  * ctypes bindings to oracle/libawq_oracle.so (the C restatement, numpy in / numpy out);
  * `torch_*` functions: a torch-CPU restatement that walks the same op sequence as the
    reference's own CPU path (awq/utils/packing_utils.py:87-102 then awq/modules/linear/
    gemm.py:76-79) -- broadcast shift, column gather, mask, repeat_interleave, multiply,
    matmul -- so that timing it on the GPU box's host cores is a fair "AutoAWQ CPU path"
    number (bench.py cpu_baseline.kind == "port").
Parity status: pinned against tests/golden/*.npz (reference outputs) by tests/test_oracle.py.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libawq_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "awq_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libawq_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.awq_oracle_zeros_width.restype = ctypes.c_int64
        _lib.awq_oracle_zeros_width.argtypes = [ctypes.c_int64, ctypes.c_int64]
        _lib.awq_oracle_moe_align.restype = ctypes.c_int32
        _lib.awq_oracle_h2f.restype = ctypes.c_float
        _lib.awq_oracle_h2f.argtypes = [ctypes.c_uint16]
        _lib.awq_oracle_f2h.restype = ctypes.c_uint16
        _lib.awq_oracle_f2h.argtypes = [ctypes.c_float]
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _i64(v):
    return ctypes.c_int64(int(v))


def unpack_gemm(q):
    """[rows, words] int32 -> [rows, 8*words] uint8 nibbles in logical column order."""
    q = _c(q, np.int32)
    out = np.empty((q.shape[0], q.shape[1] * 8), np.uint8)
    lib().awq_oracle_unpack_gemm(_p(q), _i64(q.shape[0]), _i64(q.shape[1]), _p(out))
    return out


def dequant_gemm(qweight, qzeros, scales, g):
    qweight, qzeros = _c(qweight, np.int32), _c(qzeros, np.int32)
    scales = _c(scales, np.float16)
    K, N = qweight.shape[0], qweight.shape[1] * 8
    W = np.empty((K, N), np.float16)
    lib().awq_oracle_dequant_gemm(_p(qweight), _p(qzeros), _p(scales), _i64(K), _i64(N), _i64(g), _p(W))
    return W


def zeros_width(K, g):
    return int(lib().awq_oracle_zeros_width(K, g))


def dequant_gemv(qweight, qzeros, scales, g):
    """GEMV layout ([N, K/8] etc.) -> W in [K, N] orientation."""
    qweight, qzeros = _c(qweight, np.int32), _c(qzeros, np.int32)
    scales = _c(scales, np.float16)
    N, K = qweight.shape[0], qweight.shape[1] * 8
    W = np.empty((K, N), np.float16)
    lib().awq_oracle_dequant_gemv(_p(qweight), _p(qzeros), _p(scales), _i64(K), _i64(N), _i64(g), _p(W))
    return W


def unpack_gemvfast(qweight):
    qweight = _c(qweight, np.int16)
    N, K = qweight.shape[0] * 4, qweight.shape[1]
    w = np.empty((K, N), np.uint8)
    lib().awq_oracle_unpack_gemvfast(_p(qweight), _i64(K), _i64(N), _p(w))
    return w


def dequant_gemvfast(qweight, scales, qzeros, g):
    qweight = _c(qweight, np.int16)
    scales, qzeros = _c(scales, np.float16), _c(qzeros, np.float16)
    N, K = qweight.shape[0] * 4, qweight.shape[1]
    W = np.empty((K, N), np.float16)
    lib().awq_oracle_dequant_gemvfast(_p(qweight), _p(scales), _p(qzeros), _i64(K), _i64(N), _i64(g), _p(W))
    return W


def matmul(x, W, bias=None):
    """x [M,K] fp16, W [K,N] fp16 -> (y32 exact-ish fp32, y16 rounded fp16)."""
    x, W = _c(x, np.float16), _c(W, np.float16)
    bias = _c(bias, np.float16) if bias is not None else None
    M, K = x.shape
    N = W.shape[1]
    y32 = np.empty((M, N), np.float32)
    y16 = np.empty((M, N), np.float16)
    lib().awq_oracle_matmul(_p(x), _p(W), _p(bias), _i64(M), _i64(K), _i64(N), _p(y32), _p(y16))
    return y32, y16


def linear_gemm(x, qweight, qzeros, scales, g, bias=None):
    x = _c(x, np.float16)
    qweight, qzeros = _c(qweight, np.int32), _c(qzeros, np.int32)
    scales = _c(scales, np.float16)
    bias = _c(bias, np.float16) if bias is not None else None
    M, K = x.shape
    N = qweight.shape[1] * 8
    y32 = np.empty((M, N), np.float32)
    y16 = np.empty((M, N), np.float16)
    lib().awq_oracle_linear_gemm(_p(x), _p(qweight), _p(qzeros), _p(scales), _p(bias), _i64(M), _i64(K),
                                 _i64(N), _i64(g), _p(y32), _p(y16))
    return y32, y16


def linear_gemm_exact(x, qweight, qzeros, scales, g, bias=None, chunk=2048):
    """Exact-arithmetic product (float64): y = x @ ((w - z) * s) with NO fp16 rounding of the
    dequantised weight.  The reference rounds W to fp16 first (packing_utils.py:98-100); a kernel
    that applies the scale after the integer dot product lands between the two.  Checker only."""
    x = _c(x, np.float16).astype(np.float64)
    w = unpack_gemm(qweight).astype(np.int16)
    z = unpack_gemm(qzeros).astype(np.int16)
    sc = _c(scales, np.float16).astype(np.float32)
    K, N = w.shape
    y = np.empty((x.shape[0], N), np.float64)
    for n0 in range(0, N, chunk):
        n1 = min(N, n0 + chunk)
        d = (w[:, n0:n1] - np.repeat(z[:, n0:n1], g, axis=0)).astype(np.float32)
        W = d * np.repeat(sc[:, n0:n1], g, axis=0)  # exact in fp32: 5-bit x 11-bit significands
        y[:, n0:n1] = x @ W.astype(np.float64)
    if bias is not None:
        y += _c(bias, np.float16).astype(np.float64)[None, :]
    return y


def matmul_exact_gemv(x, qweight, qzeros, scales, g, chunk=2048):
    """Exact-arithmetic product on the GEMV layout (float64, no fp16 rounding of the dequantised weight): qweight [N, K/8] (nibble i
    of word c = w[n, 8c + i]), qzeros [N, ZW] (nibble i of word c = z[n, group 8c + i]), scales [N, 8 ZW] -- the packer
    awq/modules/linear/gemv.py:94-153.  Checker only (the default-path kernels are held to 1 ulp + 1e-4 rms of THIS,
    not only to the 6-sigma-widened bound against the fp16-rounded weights)."""
    x = _c(x, np.float16).astype(np.float64)
    qweight, qzeros = _c(qweight, np.int32), _c(qzeros, np.int32)
    sc = _c(scales, np.float16).astype(np.float32)
    N, K = qweight.shape[0], qweight.shape[1] * 8
    G = K // g
    sh = (4 * np.arange(8, dtype=np.int32))[None, None, :]
    y = np.empty((x.shape[0], N), np.float64)
    for n0 in range(0, N, chunk):
        n1 = min(N, n0 + chunk)
        w = ((qweight[n0:n1, :, None] >> sh) & 15).reshape(n1 - n0, K).astype(np.int16)                 # [n, K]
        z = ((qzeros[n0:n1, :, None] >> sh) & 15).reshape(n1 - n0, -1)[:, :G].astype(np.int16)          # [n, G]
        d = (w - np.repeat(z, g, axis=1)).astype(np.float32)
        W = d * np.repeat(sc[n0:n1, :G], g, axis=1)  # exact in fp32: 5-bit x 11-bit significands
        y[:, n0:n1] = x @ W.astype(np.float64).T
    return y


def matmul_exact_gemvfast(x, qweight, scales, qzeros, g, chunk=2048):
    """Exact-arithmetic product on the GEMVFast layout (float64): W = w * s + qzeros with NO rounding of the sum (the format stores
    qzeros = fp16(-(s z)), awq/modules/linear/gemv_fast.py:175-181; that stored value is the exact operand).  Checker only."""
    x = _c(x, np.float16).astype(np.float64)
    w = unpack_gemvfast(qweight)  # [K, N] uint8
    sc = _c(scales, np.float16).astype(np.float64)
    qz = _c(qzeros, np.float16).astype(np.float64)
    K, N = w.shape
    G = K // g
    y = np.empty((x.shape[0], N), np.float64)
    for n0 in range(0, N, chunk):
        n1 = min(N, n0 + chunk)
        W = w[:, n0:n1].astype(np.float64) * np.repeat(sc[:G, n0:n1], g, axis=0) + np.repeat(qz[:G, n0:n1], g, axis=0)
        y[:, n0:n1] = x @ W
    return y


def weight_rounding_sigma(x, W):
    """Std-dev bound of the product noise caused by the reference's own fp16 rounding of W:
    each W[k,n] carries an error <= ulp/2, ulp <= 2^-10 |W| (uniform: sigma = ulp/sqrt(12)), so
    sigma_y[m,n] <= 2^-10/sqrt(12) * sqrt(sum_k x[m,k]^2 W[k,n]^2)."""
    x2 = _c(x, np.float16).astype(np.float64) ** 2
    W2 = np.asarray(W, np.float32).astype(np.float64) ** 2
    return (2.0 ** -10 / np.sqrt(12.0)) * np.sqrt(x2 @ W2)


def silu_and_mul(gate_up):
    gate_up = _c(gate_up, np.float16)
    d = gate_up.shape[-1] // 2
    rows = gate_up.size // (2 * d)
    out = np.empty(gate_up.shape[:-1] + (d,), np.float16)
    lib().awq_oracle_silu_and_mul(_p(gate_up), _i64(rows), _i64(d), _p(out))
    return out


def moe_align(topk_ids, num_experts, block):
    ids = _c(topk_ids, np.int32).reshape(-1)
    numel = ids.size
    sorted_ids = np.empty(numel + num_experts * (block - 1), np.int32)
    expert_ids = np.full(numel + num_experts, -1, np.int32)
    n = lib().awq_oracle_moe_align(_p(ids), _i64(numel), ctypes.c_int32(num_experts), ctypes.c_int32(block),
                                   _p(sorted_ids), _p(expert_ids))
    return sorted_ids, expert_ids, int(n)


def moe_forward(x, gating, w1, w2, top_k, g):
    """Appendix A.5 restatement of awq/modules/fused/moe.py:45-91,137-171 in numpy + the C oracle.
    x [T,H] fp16; gating [T,E]; w1/w2 = dicts(qweight [E,K,N/8], qzeros, scales) GEMM layout.
    Intermediates are rounded to fp16 where the reference kernels return fp16 tensors."""
    x = _c(x, np.float16)
    T = x.shape[0]
    logits = np.asarray(gating, np.float32)
    p = np.exp(logits - logits.max(-1, keepdims=True))
    p = p / p.sum(-1, keepdims=True)
    ids = np.argsort(-p, axis=-1, kind="stable")[:, :top_k]
    wt = np.take_along_axis(p, ids, -1)
    wt = wt / wt.sum(-1, keepdims=True)
    H2 = w2["qweight"].shape[2] * 8
    y = np.zeros((T, H2), np.float32)
    for t in range(T):
        for j in range(top_k):
            e = int(ids[t, j])
            _, gu = linear_gemm(x[t:t + 1], w1["qweight"][e], w1["qzeros"][e], w1["scales"][e], g)
            act = silu_and_mul(gu)
            h32, _ = linear_gemm(act, w2["qweight"][e], w2["qzeros"][e], w2["scales"][e], g)
            y[t] += np.float16(np.float32(wt[t, j]) * h32[0]).astype(np.float32)
    return y.astype(np.float16), ids.astype(np.int32), wt.astype(np.float32)


# ------------------------------------------------------------------ torch-CPU "port" baseline

_REV = (0, 4, 1, 5, 2, 6, 3, 7)


def torch_dequantize_gemm(qweight, qzeros, scales, group_size):
    """Same op sequence as awq/utils/packing_utils.py:87-102 (unpack_awq :8-26, reverse_awq_order
    :29-43, mask :94-95, repeat_interleave + multiply :98-100), written from that description."""
    import torch

    def explode(q):
        sh = torch.arange(0, 32, 4, device=q.device)
        nib = torch.bitwise_right_shift(q.unsqueeze(-1), sh.view(1, 1, 8)).to(torch.int8)
        return nib.reshape(q.shape[0], -1)

    iw, iz = explode(qweight), explode(qzeros)
    cols = torch.arange(iw.shape[-1], dtype=torch.int32, device=iw.device).view(-1, 8)[:, list(_REV)].reshape(-1)
    iw = torch.bitwise_and(iw[:, cols], 15)
    iz = torch.bitwise_and(iz[:, cols], 15)
    s = scales.repeat_interleave(group_size, dim=0)
    z = iz.repeat_interleave(group_size, dim=0)
    return (iw - z) * s


def torch_linear_gemm(x, qweight, qzeros, scales, group_size, bias=None):
    """awq/modules/linear/gemm.py:71-79 naive branch: dequantize, matmul, + bias."""
    import torch

    W = torch_dequantize_gemm(qweight, qzeros, scales, group_size)
    out = torch.matmul(x.to(torch.float16), W)
    return out + bias if bias is not None else out
