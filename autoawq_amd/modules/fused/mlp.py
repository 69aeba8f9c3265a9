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
from ..linear.gemv import WQLinear_GEMV, tensor_key


class QuantFusedMLP(nn.Module):
    FUSE_ACTIVATION_INTO_DOWN = True  # decode-sized batches, GEMM layout

    def __init__(self, gate_proj, down_proj, up_proj, activation=F.silu):
        super().__init__()
        self.in_features = gate_proj.in_features
        self.intermediate_size = gate_proj.out_features
        self.out_features = down_proj.out_features
        self.w_bit = gate_proj.w_bit
        self.down_proj = down_proj
        self.gemv_layout = isinstance(down_proj, WQLinear_GEMV)
        self.group_size = down_proj.group_size
        self.activation = activation
        self._fused = None
        self._pairs = None
        self._adopt(gate_proj.qweight, gate_proj.scales, gate_proj.qzeros, up_proj.qweight, up_proj.scales, up_proj.qzeros)

    def _adopt(self, gq, gs, gz, uq, us, uz):
        """ONE resident copy of the gate / up weights: the [gate | up] concatenation the fused projection reads; the six
        buffers the reference registers (mlp.py:25-32: `gate_proj_qweight` ... `up_proj_qzeros`, the names checkpoints and
        `state_dict()` use) are views into it, so loading into them, or reading them, touches the same memory."""
        dim = 0 if self.gemv_layout else 1  # GEMV layout stacks output rows, GEMM layout columns
        self._pairs = None
        with torch.inference_mode(False):  # (the registered buffers become views of these: never inference tensors, so that a
            # later load_state_dict -- an in-place copy from an ordinary context -- works even if a forward under inference mode re-fused them)
            fused = tuple(torch.cat([g, u], dim=dim).contiguous() for g, u in ((gq, uq), (gs, us), (gz, uz)))
        self._fused = fused
        for name, f, g in (("qweight", fused[0], gq), ("scales", fused[1], gs), ("qzeros", fused[2], gz)):
            n = g.shape[dim]
            for proj, view in (("gate_proj_", f.narrow(dim, 0, n)), ("up_proj_", f.narrow(dim, n, f.shape[dim] - n))):
                if proj + name in self._buffers:
                    self._buffers[proj + name] = view
                else:
                    self.register_buffer(proj + name, view)

    def _apply(self, fn, recurse=True):
        # .to() / .cuda() / .half() give every buffer a storage of its own: fuse the moved tensors again -- unless nothing
        # moved (a no-op .to(): the views still sit in the fused storage)
        super()._apply(fn, recurse)
        base = self._fused[0].untyped_storage().data_ptr() if self._fused is not None else None
        if base is None or self.gate_proj_qweight.untyped_storage().data_ptr() != base:
            self._adopt(self.gate_proj_qweight, self.gate_proj_scales, self.gate_proj_qzeros,
                        self.up_proj_qweight, self.up_proj_scales, self.up_proj_qzeros)
        return self

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        # the six registered buffers are views into the fused [gate | up] tensors: hand out tensors with storages of their
        # own (safetensors' save_file rejects tensors that share memory)
        super()._save_to_state_dict(destination, prefix, keep_vars)
        for proj in ("gate_proj_", "up_proj_"):
            for name in ("qweight", "scales", "qzeros"):
                key = prefix + proj + name
                if key in destination:
                    destination[key] = destination[key].clone()

    def _gate_up_fused(self):
        """Concatenated [gate | up] buffers (re-fused if a caller re-assigned one of the registered views)."""
        base = self._fused[0].untyped_storage().data_ptr()
        if (self.gate_proj_qweight.untyped_storage().data_ptr() != base or self.up_proj_qweight.untyped_storage().data_ptr() != base
                or self.gate_proj_scales.untyped_storage().data_ptr() != self._fused[1].untyped_storage().data_ptr()
                or self.up_proj_scales.untyped_storage().data_ptr() != self._fused[1].untyped_storage().data_ptr()
                or self.gate_proj_qzeros.untyped_storage().data_ptr() != self._fused[2].untyped_storage().data_ptr()
                or self.up_proj_qzeros.untyped_storage().data_ptr() != self._fused[2].untyped_storage().data_ptr()):
            self._adopt(self.gate_proj_qweight, self.gate_proj_scales, self.gate_proj_qzeros,
                        self.up_proj_qweight, self.up_proj_scales, self.up_proj_qzeros)
        return self._fused

    def gate_up_pairs(self):
        """GEMV layout, decode: the gate and up rows INTERLEAVED (row 2 i = gate_i, row 2 i + 1 = up_i), the form
        `ops.gemv_forward_ex(..., silu_pairs=True)` reads -- the wave that finishes a (gate, up) pair writes
        silu(gate) * up itself, so the [1, 2 I] intermediate and the awq_silu_and_mul launch (mlp.py:64-66) disappear.
        A second resident copy of the two projections (2 x 23 MB per 7B layer), built at the first decode step from the
        registered buffers; rebuilt when they are re-assigned, moved or written in place (`load_state_dict`)."""
        qw, sc, qz = self._gate_up_fused()
        key = tensor_key(qw, sc, qz)  # re-assigned views re-fuse (new pointers); an in-place load bumps the versions
        if self._pairs is None or self._pairs[0] != key:
            I = self.intermediate_size
            self._pairs = (key, tuple(torch.stack([t[:I], t[I:]], dim=1).reshape(t.shape).contiguous() for t in (qw, sc, qz)))
        return self._pairs[1]

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
