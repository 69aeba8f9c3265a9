#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by RUNNING the reference.

This script is the only file in the repo that imports /root/reference. It is run by hand in
the build container (where /root/reference exists), never by tests, smoke() or bench.py:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What it pins:
  * awq/utils/packing_utils.py:87-102  dequantize_gemm  (GEMM-layout unpack + reorder + dequant)
  * awq/modules/linear/gemm.py:71-79   naive branch  x @ W (+bias)  of WQLinear_GEMM.forward
  * awq/modules/linear/gemm.py:171-251 / gemv.py:77-154 / gemv_fast.py:26-65,127-183
    the three from_linear packers (=> the GEMV / GEMVFast bit layouts of Appendix A.3 / A.4)
  * awq/quantize/quantizer.py:74-109   pseudo_quantize_tensor (how realistic w/z/s are made)
  * awq/utils/fused_utils.py:45-142    fuse_qkv on GEMM-layout modules
Every array is written as a small .npz next to this script.
"""
import os
import sys

sys.dont_write_bytecode = True
REF = "/root/reference"
sys.path.insert(0, REF)

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))

import awq.modules.linear.gemm as ref_gemm  # noqa: E402
from awq.modules.linear.gemm import WQLinear_GEMM  # noqa: E402
from awq.modules.linear.gemv import WQLinear_GEMV  # noqa: E402
from awq.modules.linear.gemv_fast import WQLinear_GEMVFast  # noqa: E402
from awq.utils.packing_utils import dequantize_gemm, unpack_awq, reverse_awq_order  # noqa: E402
from awq.utils.fused_utils import fuse_qkv  # noqa: E402

# the CPU naive branch is only reachable with the Triton probe off
ref_gemm.TRITON_AVAILABLE = False
ref_gemm.user_has_been_warned = True

MAX_INT32 = 0x7FFFFFFF
MIN_INT32 = -MAX_INT32 - 1


def npy(t):
    return t.detach().cpu().contiguous().numpy()


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path, {k: (v.shape, str(v.dtype)) for k, v in arrs.items()})


def ref_forward(mod, x):
    with torch.no_grad():
        return mod(x)


def kat_a6():
    qweight = torch.tensor([[0x76543210], [0xFEDCBA98 - (1 << 32)]], dtype=torch.int32)
    qzeros = torch.tensor([[0x11111111]], dtype=torch.int32)
    scales = torch.tensor([[0.5, 1, 2, 0.25, 1, 1, 1, -1]], dtype=torch.float16)
    W = dequantize_gemm(qweight, qzeros, scales, 4, 2)
    iw, iz = unpack_awq(qweight, qzeros, 4)
    iw, iz = reverse_awq_order(iw, iz, 4)
    x = torch.tensor([[1.0, 2.0]], dtype=torch.float16)
    y = torch.matmul(x, W)
    save("kat_a6", qweight=npy(qweight), qzeros=npy(qzeros), scales=npy(scales),
         w_int=npy(iw & 15).astype(np.uint8), z_int=npy(iz & 15).astype(np.uint8),
         W=npy(W), x=npy(x), y=npy(y))


def fullrange(K, N, g, M, seed, name):
    """Recipe of tests/test_dequantization.py:15-38 (full-range int32 incl. sign bit), on CPU."""
    torch.manual_seed(seed)
    qweight = torch.randint(MIN_INT32, MAX_INT32, (K, N // 8), dtype=torch.int32)
    qzeros = torch.randint(MIN_INT32, MAX_INT32, (K // g, N // 8), dtype=torch.int32)
    scales = torch.randn((K // g, N), dtype=torch.float16)
    bias = torch.randn((N,), dtype=torch.float16)
    x = torch.randn((M, K), dtype=torch.float16)
    W = dequantize_gemm(qweight, qzeros, scales, 4, g)
    iw, iz = unpack_awq(qweight, qzeros, 4)
    iw, iz = reverse_awq_order(iw, iz, 4)
    mod = WQLinear_GEMM(4, g, K, N, True, "cpu")
    mod.qweight, mod.qzeros, mod.scales, mod.bias = qweight, qzeros, scales, bias
    y = ref_forward(mod, x)
    mod.bias = None
    y_nobias = ref_forward(mod, x)
    save(name, qweight=npy(qweight), qzeros=npy(qzeros), scales=npy(scales), bias=npy(bias),
         w_int=npy(iw & 15).astype(np.uint8), z_int=npy(iz & 15).astype(np.uint8),
         W=npy(W), x=npy(x), y=npy(y), y_nobias=npy(y_nobias),
         group_size=np.int32(g))


def pseudo_quant(w, g):
    """awq/quantize/quantizer.py:74-109 with zero_point=True, w_bit=4 (standalone restatement
    is not needed here: we only need its outputs, so call the same math inline)."""
    shape = w.shape
    wg = w.reshape(-1, g)
    mx = wg.amax(dim=1, keepdim=True)
    mn = wg.amin(dim=1, keepdim=True)
    scales = (mx - mn).clamp(min=1e-5) / 15
    zeros = (-torch.round(mn / scales)).clamp_(0, 15)
    wq = (torch.clamp(torch.round(wg / scales) + zeros, 0, 15) - zeros) * scales
    return wq.reshape(shape), scales.view(shape[0], -1), zeros.view(shape[0], -1)


def packed(K, N, g, M, seed, name):
    torch.manual_seed(seed)
    lin = torch.nn.Linear(K, N, bias=True).half()
    lin.weight.data = (torch.randn(N, K) * 0.05).half()
    lin.bias.data = torch.randn(N).half()
    wq, scales, zeros = pseudo_quant(lin.weight.data.float(), g)
    lin.weight.data = wq.half()
    scales = scales.half()
    # GEMM layout wants [G, N] (quantizer.py:236-240); GEMV / GEMVFast want [N, G]
    m_gemm = WQLinear_GEMM.from_linear(lin, 4, g, False, scales.t().contiguous(), zeros.t().contiguous())
    m_gemv = WQLinear_GEMV.from_linear(lin, 4, g, False, scales, zeros)
    m_fast = WQLinear_GEMVFast.from_linear(lin, 4, g, False, scales, zeros)
    W = dequantize_gemm(m_gemm.qweight, m_gemm.qzeros, m_gemm.scales, 4, g)
    iw, iz = unpack_awq(m_gemm.qweight, m_gemm.qzeros, 4)
    iw, iz = reverse_awq_order(iw, iz, 4)
    x = torch.randn((M, K), dtype=torch.float16)
    y = ref_forward(m_gemm, x)
    save(name,
         gemm_qweight=npy(m_gemm.qweight), gemm_qzeros=npy(m_gemm.qzeros), gemm_scales=npy(m_gemm.scales),
         gemv_qweight=npy(m_gemv.qweight), gemv_qzeros=npy(m_gemv.qzeros), gemv_scales=npy(m_gemv.scales),
         fast_qweight=npy(m_fast.qweight), fast_qzeros=npy(m_fast.qzeros), fast_scales=npy(m_fast.scales),
         bias=npy(m_gemm.bias), w_int=npy(iw & 15).astype(np.uint8), z_int=npy(iz & 15).astype(np.uint8),
         W=npy(W), x=npy(x), y=npy(y), lin_weight=npy(lin.weight.data),
         group_size=np.int32(g))


def fused_qkv(K, Nq, Nkv, g, seed, name):
    torch.manual_seed(seed)
    mods = []
    for n in (Nq, Nkv, Nkv):
        m = WQLinear_GEMM(4, g, K, n, True, "cpu")
        m.qweight = torch.randint(MIN_INT32, MAX_INT32, (K, n // 8), dtype=torch.int32)
        m.qzeros = torch.randint(MIN_INT32, MAX_INT32, (K // g, n // 8), dtype=torch.int32)
        m.scales = (torch.rand((K // g, n)) * 0.02 + 0.005).half()
        m.bias = torch.randn(n).half()
        mods.append(m)
    x = torch.randn((2, 3, K), dtype=torch.float16)
    ys = [ref_forward(m, x) for m in mods]
    parts = {}
    for tag, m in zip("qkv", mods):
        parts[tag + "_qweight"] = npy(m.qweight)
        parts[tag + "_qzeros"] = npy(m.qzeros)
        parts[tag + "_scales"] = npy(m.scales)
        parts[tag + "_bias"] = npy(m.bias)
    holder = torch.nn.Module()
    holder.register_buffer("dummy", torch.zeros(1))
    fused = fuse_qkv(holder, *mods)
    y = ref_forward(fused, x)
    assert torch.equal(y, torch.cat(ys, dim=-1))
    save(name, x=npy(x), y=npy(y), fused_qweight=npy(fused.qweight), fused_qzeros=npy(fused.qzeros),
         fused_scales=npy(fused.scales), fused_bias=npy(fused.bias), group_size=np.int32(g), **parts)


if __name__ == "__main__":
    kat_a6()
    fullrange(256, 64, 128, 3, 0, "fullrange_K256_N64_g128")
    fullrange(128, 32, 32, 2, 1, "fullrange_K128_N32_g32")
    packed(512, 64, 128, 4, 2, "packed_K512_N64_g128")
    packed(256, 32, 64, 2, 3, "packed_K256_N32_g64")
    packed(128, 32, 32, 1, 4, "packed_K128_N32_g32")
    fused_qkv(256, 64, 32, 128, 5, "fused_qkv_K256_g128")
