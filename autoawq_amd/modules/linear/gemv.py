"""WQLinear_GEMV for MI355X: the nn.Module surface of awq/modules/linear/gemv.py on gfx950 kernels.

Drop-in contract kept (reference awq/modules/linear/gemv.py:27-197):
  * ctor (w_bit, group_size, in_features, out_features, bias, dev); attribute split_k_iters = 8;
  * registered BUFFERS qweight [N, K/8] i32 (ordinal nibbles), qzeros [N, ZW] i32,
    scales [N, 8*ZW] f16 zero-padded, bias [N] f16 | None -- the keys of an AWQ "gemv" checkpoint;
    ZW = calculate_zeros_width(K, g) (:12-24);
  * from_linear(linear, w_bit, group_size, init_only=False, scales=None, zeros=None) (:77-154);
  * forward (:156-186): any leading dims, any float dtype (computed in fp16, cast back), bias added
    AFTER the cast back, in the input dtype (:183-185).
What differs by design: the arithmetic runs in libawq_hip.so (csrc/gemv_nk.hip for M <= 16 per
launch; bit-exact dequant + fp16 GEMM above 64 rows); there is no CPU path -- a non-HIP tensor
raises.
"""
import torch
import torch.nn as nn

from ... import _lib, ops
from ...utils.packing import (GEMV_ORDER, calculate_zeros_width, pack_rows_int4, pack_zeros_nk,
                              quantize_int_weights_nk)

# from this many rows a call runs the fused MFMA GEMM kernels (gemm_skinny / gemm_tiled / gemm_regb by row count) on a cached
# GEMM-layout repack of the same integers instead of ceil(M/16) passes of the 16-row decode kernel (4096 x 11008, M = 32:
# 14 us vs 2 x 24; the reference switches to its batched kernel at 8 rows, gemv.py:168)
PREFILL_MIN_ROWS = 17


def tensor_key(*tensors):
    """Identity AND content version of tensors a derived cache was built from: `data_ptr` changes when a buffer is re-assigned,
    `_version` when it is written in place (`load_state_dict`, `.copy_()`) -- ADVICE r03: a key of pointers alone went stale."""
    return tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in tensors)


def _gemm_layout_copy(m):
    """The same integers repacked (bit-exactly, utils/convert.py) into the GEMM layout, built on first use and kept: batches the
    decode kernels of this layout do not take (prefill) run on the fused MFMA kernels of awq_gemm_forward instead of
    dequantise + vendor GEMM.  Costs one more int4 copy of the weights (HBM is 288 GB); rebuilt if the buffers are re-assigned or written in place."""
    key = tensor_key(m.qweight, m.scales, m.qzeros)
    c = m.__dict__.get("_gemm_copy")
    if c is None or c[0] != key:
        from ...utils.convert import pack_linear, unpack_linear

        w, z, s, _ = unpack_linear(m)
        g = pack_linear("gemm", w, z, s, None, m.in_features, m.out_features, m.group_size)
        c = (key, g.qweight, g.scales, g.qzeros)
        m.__dict__["_gemm_copy"] = c
    return c[1], c[2], c[3]


class WQLinear_GEMV(nn.Module):
    def __init__(self, w_bit, group_size, in_features, out_features, bias, dev):
        super().__init__()
        if w_bit not in [4]:
            raise NotImplementedError("Only 4-bit are supported for now.")
        self.in_features = in_features
        self.out_features = out_features
        self.w_bit = w_bit
        self.group_size = group_size if group_size != -1 else in_features
        self.split_k_iters = 8

        assert self.in_features % self.group_size == 0
        assert out_features % (32 // self.w_bit) == 0
        pack = 32 // self.w_bit
        zw = calculate_zeros_width(in_features, self.group_size)
        self.register_buffer("qweight", torch.zeros((out_features, in_features // pack), dtype=torch.int32, device=dev))
        self.register_buffer("qzeros", torch.zeros((out_features, zw), dtype=torch.int32, device=dev))
        self.register_buffer("scales", torch.zeros((out_features, zw * pack), dtype=torch.float16, device=dev))
        if bias:
            self.register_buffer("bias", torch.zeros((out_features), dtype=torch.float16, device=dev))
        else:
            self.bias = None

    @classmethod
    def from_linear(cls, linear, w_bit, group_size, init_only=False, scales=None, zeros=None):
        awq_linear = cls(w_bit, group_size, linear.in_features, linear.out_features, linear.bias is not None,
                         linear.weight.device)
        if init_only:  # buffers are filled later by load_state_dict
            return awq_linear
        assert scales is not None and zeros is not None  # both [N, G]
        zw = calculate_zeros_width(linear.in_features, group_size)
        qscales = torch.zeros((scales.shape[0], zw * 8), dtype=torch.float16, device=scales.device)
        qscales[:, : scales.shape[1]] = scales
        awq_linear.scales = qscales
        if linear.bias is not None:
            awq_linear.bias = linear.bias.clone().half()
        intweight = quantize_int_weights_nk(linear.weight.data, scales, zeros, qscales, group_size)  # [N, K]
        awq_linear.qweight = pack_rows_int4(intweight, GEMV_ORDER)
        awq_linear.qzeros = pack_zeros_nk(zeros, zw)
        return awq_linear

    @torch.no_grad()
    def forward(self, x):
        out_shape = x.shape[:-1] + (self.out_features,)
        inputs = x.reshape(-1, x.shape[-1])
        input_dtype = inputs.dtype
        if input_dtype != torch.float16:
            inputs = inputs.half()
        if inputs.shape[0] >= PREFILL_MIN_ROWS:
            out = ops.gemm_forward(inputs, *_gemm_layout_copy(self))
        else:
            try:
                out = ops.gemv_forward(inputs, self.qweight, self.scales, self.qzeros, self.group_size)
            except _lib.AwqHipError as e:  # a shape the decode kernels do not take (K % 128, unusual group sizes): the
                if e.code != _lib.ERR_UNSUPPORTED:  # GEMM-layout kernels handle every valid tensor
                    raise
                out = ops.gemm_forward(inputs, *_gemm_layout_copy(self))
        if input_dtype != torch.float16:
            out = out.to(dtype=input_dtype)
        out = out + self.bias if self.bias is not None else out
        return out.reshape(out_shape)

    def extra_repr(self) -> str:
        return "in_features={}, out_features={}, bias={}, w_bit={}, group_size={}".format(
            self.in_features, self.out_features, self.bias is not None, self.w_bit, self.group_size)
