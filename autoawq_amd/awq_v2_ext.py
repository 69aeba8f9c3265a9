"""`awq_v2_ext`-compatible shim (GEMVFast kernels) over the C ABI.

The reference probes `try_import("awq_v2_ext")` (awq/modules/linear/gemv_fast.py:5) and calls the
two functions below (gemv_fast.py:191-206).  `sys.modules["awq_v2_ext"] = autoawq_amd.awq_v2_ext`
before importing `awq` makes the unmodified reference WQLinear_GEMVFast run on the gfx950 kernels.
"""
import torch

from . import _lib, ops
from .utils.packing import calculate_zeros_width


def gemv_forward_cuda_decode(x, qweight, scales, qzeros, m, n, k, group_size):
    """gemv_fast.py:191-201: x [batch, 1, K] (or [m, K]); returns x.shape[:-1] + (n,)."""
    out = ops.gemv_fast_forward(x.reshape(-1, x.shape[-1]), qweight, scales, qzeros, group_size)
    return out.reshape(x.shape[:-1] + (n,))


def infer_group_size(K, group_rows):
    """The prefill entry point is not told the group size (gemv_fast.py:203-206); the padded row count of
    scales / qzeros determines it: rows == calculate_zeros_width(K, g) * 8 (gemv_fast.py:86-104).  The
    largest candidate wins when the padding makes several fit (it is the layout's default)."""
    fits = [g for g in (128, 64, 32) if K % g == 0 and calculate_zeros_width(K, g) * 8 == group_rows]
    if len(fits) > 1:  # e.g. K = 512: 8 padded rows fit g = 128 and g = 64 -- said out loud, not resolved silently
        import warnings

        warnings.warn(f"awq_v2_ext.gemm_forward_cuda_prefill: {group_rows} scale rows at K={K} fit group sizes {fits}; taking {fits[0]} "
                      f"(use WQLinear_GEMVFast.forward, which knows its group size, to be exact)", RuntimeWarning, stacklevel=3)
    if fits:
        return fits[0]
    raise ValueError(f"awq_v2_ext.gemm_forward_cuda_prefill: cannot infer the group size from K={K} and "
                     f"{group_rows} scale rows")


def gemm_forward_cuda_prefill(x, qweight, scales, qzeros):
    """gemv_fast.py:203-206."""
    K, N = x.shape[-1], qweight.shape[0] * 4
    x2 = x.reshape(-1, K)
    g = infer_group_size(K, scales.shape[0])
    from .modules.linear.gemv import prefill_min_rows, prefill_route

    # (the batched-decode kernel in launches of <= 128 rows while it measures ahead of the prefill routes: gemv.prefill_min_rows)
    if (x2.shape[0] <= 64 or (x2.shape[0] < prefill_min_rows(K) and g == 128 and K % 128 == 0)) and N % 16 == 0:
        return ops.gemv_fast_forward(x2, qweight, scales, qzeros, g).reshape(x.shape[:-1] + (N,))

    out = None
    if prefill_route(x2.shape[0], K, N) == "hand":
        try:  # the words transposed into a temporary + the fused MFMA GEMM with this layout's arithmetic (two hand-written launches)
            out = ops.gemv_fast_prefill(x2.half(), qweight, scales, qzeros, g)
        except _lib.AwqHipError as e:  # shapes the fused kernel refuses (K % 64, group sizes below 64, N % 8)
            if e.code != _lib.ERR_UNSUPPORTED:
                raise
    if out is None:  # dequantise into a temporary + a dense fp16 GEMM (the reference's two-pass route, gemm.py:48-54)
        out = torch.matmul(x2.half(), ops.dequantize_weights_gemv_fast(qweight, scales, qzeros, g).t())
    return out.reshape(x.shape[:-1] + (N,))
