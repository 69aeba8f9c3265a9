"""Buffer-level fusion helpers (reference: awq/utils/fused_utils.py:45-162).

`fuse_qkv` / `fuse_linears` concatenate the packed buffers of sibling WQLinear modules so one
kernel launch serves them: GEMM layout concatenates along N (dim 1 of qweight / qzeros / scales),
GEMV layout along the output rows (dim 0).  With `operation=torch.stack` the experts of an MoE layer
are stacked on a new leading dim (awq/models/mixtral.py:130-158)."""
import torch

from ..modules.linear.gemm import WQLinear_GEMM
from ..modules.linear.gemv import WQLinear_GEMV
from ..modules.linear.gemv_fast import WQLinear_GEMVFast


def fuse_linears(linears, device=None, dim=1, operation=torch.cat):
    """Same call forms as awq/utils/fused_utils.py:145-162 (`dim=1, operation=torch.cat` joins N of
    GEMM-layout modules; `dim=0, operation=torch.stack` stacks experts)."""
    first = linears[0]
    device = device if device is not None else first.qweight.device
    total_out = sum(l.out_features for l in linears) if operation is torch.cat else first.out_features
    fused = type(first)(first.w_bit, first.group_size, first.in_features, total_out, bias=first.bias is not None,
                        dev=device)
    fused.qweight = operation([l.qweight for l in linears], dim=dim).to(device)
    fused.qzeros = operation([l.qzeros for l in linears], dim=dim).to(device)
    fused.scales = operation([l.scales for l in linears], dim=dim).to(device)
    if first.bias is not None:
        fused.bias = operation([l.bias for l in linears], dim=0).to(device)
    for l in linears:
        del l.qweight, l.qzeros, l.scales
    return fused


def fuse_qkv(module, q_proj, k_proj, v_proj):
    """awq/utils/fused_utils.py:45-142 for the three layouts served here: GEMM concatenates N on
    dim 1 of every buffer (`:87-96`), GEMV on dim 0 (`:76-86`), GEMVFast qweight on dim 0 and
    scales / zeros on dim 1 (`:125-135`); the source modules give their buffers up (`:139-140`)."""
    first = q_proj
    bias = torch.cat([q_proj.bias, k_proj.bias, v_proj.bias], dim=0) if q_proj.bias is not None else None
    if isinstance(first, WQLinear_GEMV):
        cls, dw, dzs = WQLinear_GEMV, 0, 0
    elif isinstance(first, WQLinear_GEMVFast):
        cls, dw, dzs = WQLinear_GEMVFast, 0, 1
    elif isinstance(first, WQLinear_GEMM):
        cls, dw, dzs = WQLinear_GEMM, 1, 1
    else:
        raise TypeError(f"fuse_qkv: unsupported linear type {type(first).__name__}")
    qkv = cls(first.w_bit, first.group_size, first.in_features,
              q_proj.out_features + k_proj.out_features + v_proj.out_features, bias is not None,
              first.qweight.device)
    qkv.qweight = torch.cat([q_proj.qweight, k_proj.qweight, v_proj.qweight], dim=dw)
    qkv.qzeros = torch.cat([q_proj.qzeros, k_proj.qzeros, v_proj.qzeros], dim=dzs).contiguous()
    qkv.scales = torch.cat([q_proj.scales, k_proj.scales, v_proj.scales], dim=dzs).contiguous()
    if hasattr(first, "split_k_iters"):
        qkv.split_k_iters = first.split_k_iters
    qkv.bias = bias
    for m in (q_proj, k_proj, v_proj):
        del m.qweight, m.qzeros, m.scales
    return qkv


def get_attention_shapes(attention_shapes, n_heads, n_kv_heads, head_dim):
    """awq/utils/fused_utils.py:165-201: how QuantAttentionFused views one fused qkv row and slices q / k / v out of it.  A
    caller-supplied dict passes through; n_kv_heads == 0 is the multi-head form `[3, n_heads, head_dim]` (q, k, v planes);
    otherwise the grouped-query form `[n_heads + 2 n_kv_heads, head_dim]` with the q heads first, then k, then v.  Both are the
    q | k | v head order in memory -- the order the gfx950 kernels read."""
    if attention_shapes is not None:
        return attention_shapes
    if n_kv_heads == 0:
        kv, view = n_heads, (-1, n_heads, head_dim)
        pick = {name: (lambda xqkv, i=i: xqkv[:, :, i]) for i, name in enumerate(("xq_slice", "xk_slice", "xv_slice"))}
    else:
        kv, view = n_kv_heads, (n_heads + 2 * n_kv_heads, head_dim)
        bounds = {"xq_slice": (0, n_heads), "xk_slice": (n_heads, n_heads + kv), "xv_slice": (n_heads + kv, n_heads + 2 * kv)}
        pick = {name: (lambda xqkv, a=a, b=b: xqkv[:, :, a:b]) for name, (a, b) in bounds.items()}
    shapes = {"xqkv_view": view, **pick}
    for prefix in ("", "single_"):
        shapes[prefix + "xq_view"] = (n_heads, head_dim)
        shapes[prefix + "xk_view"] = (kv, head_dim)
        shapes[prefix + "xv_view"] = (kv, head_dim)
    shapes["xk_reshape"] = (kv, head_dim // 8, 8)
    return shapes
