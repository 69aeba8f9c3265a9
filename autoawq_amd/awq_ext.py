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


def gemv_forward_cuda(x, qweight, scales, qzeros, group_size):
    """awq/modules/linear/gemv.py:177-180 ; awq/modules/fused/mlp.py:37-39 (GEMV layout, M <= 8)."""
    return ops.gemv_forward(x.reshape(-1, x.shape[-1]), qweight, scales, qzeros, group_size)


def gemmv2_forward_cuda(x, qweight, scales, qzeros, group_size, split_k_iters=8):
    """awq/modules/linear/gemv.py:168-176 (GEMV layout, M > 8)."""
    return ops.gemv_forward(x.reshape(-1, x.shape[-1]), qweight, scales, qzeros, group_size)


def silu_and_mul(out, gate_up):
    """awq/modules/fused/moe.py:73-76: writes into the caller's `out`."""
    ops.silu_and_mul(gate_up, out)


def grouped_gemm_forward(x, qweight, scales, qzeros, topk_weights, sorted_token_ids, expert_ids,
                         num_tokens_post_padded, mul_weights, split_k_iters=8):
    """awq/modules/fused/moe.py:60-89."""
    return ops.grouped_gemm_forward(x, qweight, scales, qzeros, topk_weights, sorted_token_ids, expert_ids,
                                    num_tokens_post_padded, mul_weights, split_k_iters)


def moe_alig_block_size(topk_ids, num_experts, block_size, sorted_ids, expert_ids, num_tokens_post_pad):
    """awq/modules/fused/moe.py:129-133 (sic: the reference spells it `alig`): fills the caller's tensors."""
    s, e, n = ops.moe_align_block_size(topk_ids, block_size, num_experts)
    sorted_ids.copy_(s)
    expert_ids.copy_(e)
    num_tokens_post_pad.copy_(n)


def layernorm_forward_cuda(x, weight, out, eps):
    """awq/modules/fused/norm.py:33-36: RMSNorm into the caller-allocated `out`."""
    ops.rmsnorm(x, weight, eps, out=out)
