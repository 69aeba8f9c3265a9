"""`awq_ext`-compatible shim over the C ABI.

The reference probes `import awq_ext` (awq/modules/linear/gemm.py:11, gemv.py:6,
awq/modules/fused/mlp.py:7, moe.py:5) and calls the functions below with these positional
argument orders.  Exposing the same names lets the *unmodified* reference modules run on the
gfx950 kernels: `sys.modules["awq_ext"] = autoawq_amd.awq_ext` before importing `awq`
(see INTEGRATION.md).  Every function runs on torch's current HIP stream and returns a newly
allocated tensor unless the reference passes the output in.
"""
from . import ops


def dequantize_weights_cuda(qweight, scales, qzeros, split_k_iters=0, thx=0, thy=0, dbg=False):
    """awq/modules/linear/gemm.py:51-53,100-102 ; tests/test_dequantization.py:41-49."""
    return ops.dequantize_weights(qweight, scales, qzeros)


def gemm_forward_cuda(x, qweight, scales, qzeros, split_k_iters=8):
    """awq/modules/linear/gemm.py:56-58 ; awq/modules/fused/mlp.py:41,49-62 (5th arg unused here:
    split-K is chosen by the library for the gfx950 grid)."""
    return ops.gemm_forward(x.reshape(-1, x.shape[-1]), qweight, scales, qzeros)
