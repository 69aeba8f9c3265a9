"""QuantFusedMLP for MI355X (reference: awq/modules/fused/mlp.py:14-70).

Same constructor `(gate_proj, down_proj, up_proj, activation=F.silu)`, same registered buffers
(`gate_proj_qweight/_scales/_qzeros`, `up_proj_...`), same forward `down(act(gate(x)) * up(x))`
with the optional `routing_weights` factor.  MI355X-first difference: for GEMM-layout projections
the gate and up weights are concatenated ONCE along N (exactly what awq/utils/fused_utils.py:145-162
`fuse_linears` does for Mixtral) so a decode step issues ONE fused int4 GEMV for both, then one
`silu_and_mul` kernel, then the down projection: 3 launches instead of 2 + 2 elementwise + 1.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ..linear.gemm import WQLinear_GEMM
from ..linear.gemv import WQLinear_GEMV


class QuantFusedMLP(nn.Module):
    FUSE_ACTIVATION_INTO_DOWN = True  # decode-sized batches, GEMM layout

    def __init__(self, gate_proj, down_proj, up_proj, activation=F.silu):
        super().__init__()
        self.register_buffer("gate_proj_qweight", gate_proj.qweight)
        self.register_buffer("gate_proj_scales", gate_proj.scales)
        self.register_buffer("gate_proj_qzeros", gate_proj.qzeros)
        self.register_buffer("up_proj_qweight", up_proj.qweight)
        self.register_buffer("up_proj_scales", up_proj.scales)
        self.register_buffer("up_proj_qzeros", up_proj.qzeros)
        self.in_features = gate_proj.in_features
        self.intermediate_size = gate_proj.out_features
        self.out_features = down_proj.out_features
        self.w_bit = gate_proj.w_bit
        self.down_proj = down_proj
        self.gemv_layout = isinstance(down_proj, WQLinear_GEMV)
        self.group_size = down_proj.group_size
        self.activation = activation
        self._fused = None  # (key, qweight, scales, qzeros) of the gate|up concatenation

    def _gate_up_fused(self):
        """Concatenated [gate | up] buffers, rebuilt if a caller re-assigned the registered ones."""
        key = (self.gate_proj_qweight.data_ptr(), self.up_proj_qweight.data_ptr(), self.gate_proj_qweight._version,
               self.up_proj_qweight._version)
        if self._fused is None or self._fused[0] != key:
            dim = 0 if self.gemv_layout else 1  # GEMV layout stacks output rows, GEMM layout columns
            self._fused = (key,
                           torch.cat([self.gate_proj_qweight, self.up_proj_qweight], dim=dim).contiguous(),
                           torch.cat([self.gate_proj_scales, self.up_proj_scales], dim=dim).contiguous(),
                           torch.cat([self.gate_proj_qzeros, self.up_proj_qzeros], dim=dim).contiguous())
        return self._fused[1:]

    def forward(self, x, routing_weights=None, gate_up=None):
        """`gate_up` [rows, 2 * intermediate]: the fused gate|up projection already computed by the caller
        (LlamaLikeBlock folds the preceding norm into it); `x` then only provides shape and dtype."""
        out_shape = x.shape[:-1] + (self.intermediate_size,)
        x = x.reshape(-1, x.shape[-1])
        in_dtype = x.dtype
        if in_dtype != torch.float16:
            x = x.half()
        if gate_up is None:
            qw, sc, qz = self._gate_up_fused()
            if self.gemv_layout:
                gate_up = ops.gemv_forward(x, qw, sc, qz, self.group_size)
            else:
                gate_up = ops.gemm_forward(x, qw, sc, qz)
        if (self.activation is F.silu and not self.gemv_layout and gate_up.shape[0] <= 16 and in_dtype == torch.float16
                and isinstance(self.down_proj, WQLinear_GEMM) and self.FUSE_ACTIVATION_INTO_DOWN):
            # decode: silu(gate) * up is applied by the down projection while it stages its activations
            # (AWQ_GEMM_FLAG_X_GATED_SILU): one launch less, bit-identical to the separate kernel
            d = self.down_proj
            out = ops.gemm_forward(gate_up, d.qweight, d.scales, d.qzeros, d.bias, flags=ops.X_GATED_SILU)
            out = out.reshape(out_shape[:-1] + (self.out_features,))
            if routing_weights is not None:
                out = routing_weights * out
            return out
        if self.activation is F.silu:
            h = ops.silu_and_mul(gate_up)
        else:
            I = self.intermediate_size
            h = self.activation(gate_up[:, :I]) * gate_up[:, I:]
        h = h.reshape(out_shape)
        if in_dtype != torch.float16:
            h = h.to(in_dtype)
        out = self.down_proj(h)
        if routing_weights is not None:
            out = routing_weights * out
        return out


class QuantLlamaMLP(QuantFusedMLP):
    """Kept for backward compatibility like the reference (mlp.py:73-87): (gate, down, up) order."""

    def __init__(self, gate_proj, down_proj, up_proj):
        super().__init__(gate_proj, down_proj, up_proj)
