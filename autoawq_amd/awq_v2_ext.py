"""`awq_v2_ext`-compatible shim (GEMVFast kernels) over the C ABI.

The reference probes `try_import("awq_v2_ext")` (awq/modules/linear/gemv_fast.py:5) and calls the
two functions below (gemv_fast.py:191-206).  `sys.modules["awq_v2_ext"] = autoawq_amd.awq_v2_ext`
before importing `awq` makes the unmodified reference WQLinear_GEMVFast run on the gfx950 kernels.
"""
import torch

from . import ops


def gemv_forward_cuda_decode(x, qweight, scales, qzeros, m, n, k, group_size):
    """gemv_fast.py:191-201: x [batch, 1, K] (or [m, K]); returns x.shape[:-1] + (n,)."""
    out = ops.gemv_fast_forward(x.reshape(-1, x.shape[-1]), qweight, scales, qzeros, group_size)
    return out.reshape(x.shape[:-1] + (n,))


def gemm_forward_cuda_prefill(x, qweight, scales, qzeros):
    """gemv_fast.py:203-206: group size is implied by the shapes (K / number of used group rows is
    not recoverable from padded tensors, so the layout's default 128 is assumed like the kernel)."""
    K, N = x.shape[-1], qweight.shape[0] * 4
    x2 = x.reshape(-1, K)
    g = 128
    if x2.shape[0] <= 64 and N % 16 == 0:
        out = ops.gemv_fast_forward(x2, qweight, scales, qzeros, g)
    else:
        out = torch.matmul(x2, ops.dequantize_weights_gemv_fast(qweight, scales, qzeros, g).t())
    return out.reshape(x.shape[:-1] + (N,))
