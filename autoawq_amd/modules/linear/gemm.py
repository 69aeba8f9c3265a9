"""WQLinear_GEMM for MI355X: the nn.Module surface of awq/modules/linear/gemm.py on gfx950 kernels.

Drop-in contract kept (reference awq/modules/linear/gemm.py:116-298):
  * ctor (w_bit, group_size, in_features, out_features, bias, dev, training=False);
  * registered BUFFERS qweight [K, N/8] i32, qzeros [K/g, N/8] i32, scales [K/g, N] f16,
    bias [N] f16 | None  -- the safetensors keys of an AWQ "gemm" checkpoint;
  * from_linear(linear, w_bit, group_size, init_only=False, scales=None, zeros=None);
  * forward: any leading dims, any float dtype (computed in fp16, cast back), bias added before
    the cast back (:79 then :284-285), empty batch -> zeros (:44-45), buffers read at call time
    so callers may re-assign concatenated tensors (awq/utils/fused_utils.py:77-96).
What differs by design: the arithmetic runs in libawq_hip.so (fused int4 dequant + GEMV/GEMM
kernels for gfx950); there is no Triton and no CPU path -- a non-HIP tensor raises.
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from ... import ops
from ...utils.packing import pack_rows_int4, quantize_int_weights_kn

# Dispatch by token count M (awq_gemm_forward AUTO; measured on MI355X, profiles/r02_regb_by_m.txt, r01_small_m.txt):
# M <= 8 decode kernel (csrc/gemv_mfma.hip);  9..64 the batched register-decoded kernel (csrc/gemm_skinny.hip);  above that the
# fused dequant + MFMA GEMM through LDS with split-K (csrc/gemm_tiled.hip) until 128 x 256 tiles give every CU a block, from
# there the register-decoded prefill kernel (csrc/gemm_regb.hip).  Every forward is a hand-written kernel: the reference's own
# large-batch route -- dequantise, then a vendor fp16 GEMM (awq/modules/linear/gemm.py:48-54) -- is 5-20 % ahead of the fused
# kernels between ~256 and ~1024 tokens on some shapes (r02_regb_by_m.txt) and stays available as PREFILL_IMPL = "two_pass"
# (A/B, and backward uses it), but nothing selects it by default.
PREFILL_IMPL = "fused"  # "fused" | "two_pass"


class WQLinearMMFunction(Function):
    """Forward/backward of the int4 linear (reference: awq/modules/linear/gemm.py:24-114)."""

    @staticmethod
    def forward(ctx, x, qweight, qzeros, scales, w_bit=4, group_size=128, bias=None, out_features=0):
        ctx.save_for_backward(x, qweight, qzeros, scales, bias)
        ctx.out_features = out_features
        out_shape = x.shape[:-1] + (out_features,)
        x = x.to(torch.float16)
        if x.shape[0] == 0:
            return torch.zeros(out_shape, dtype=x.dtype, device=x.device)

        x2d = x.reshape(-1, x.shape[-1])
        out = _linear_forward(x, x2d, qweight, scales, qzeros, bias)
        out = out.reshape(out_shape)
        if out.dim() == 2:  # reference always hands back a 3-D tensor (:83-84)
            out = out.unsqueeze(0)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        x, qweight, qzeros, scales, bias = ctx.saved_tensors
        # dequantise once, then a plain fp16 GEMM against W^T (reference :97-112)
        weights = ops.dequantize_weights(qweight, scales, qzeros).to(grad_output.dtype)
        grad_input = None
        if ctx.needs_input_grad[0]:
            grad_input = torch.matmul(grad_output, weights.t())
        return grad_input, None, None, None, None, None, None, None


def _linear_forward(x, x2d, qweight, scales, qzeros, bias):
    if PREFILL_IMPL == "two_pass" and x2d.shape[0] > 16:  # opt-in A/B route: bit-exact HIP dequant + vendor fp16 GEMM
        out = torch.mm(x2d, ops.dequantize_weights(qweight, scales, qzeros))
        return out + bias if bias is not None else out
    return ops.gemm_forward(x2d, qweight, scales, qzeros, bias)


class WQLinear_GEMM(nn.Module):
    def __init__(self, w_bit, group_size, in_features, out_features, bias, dev, training=False):
        super().__init__()
        if w_bit not in [4]:
            raise NotImplementedError("Only 4-bit are supported for now.")
        self.in_features = in_features
        self.out_features = out_features
        self.w_bit = w_bit
        self.group_size = group_size if group_size != -1 else in_features
        self.training = training

        assert self.in_features % self.group_size == 0
        assert out_features % (32 // self.w_bit) == 0
        pack = 32 // self.w_bit
        groups = in_features // self.group_size
        self.register_buffer("qweight", torch.zeros((in_features, out_features // pack), dtype=torch.int32, device=dev))
        self.register_buffer("qzeros", torch.zeros((groups, out_features // pack), dtype=torch.int32, device=dev))
        self.register_buffer("scales", torch.zeros((groups, out_features), dtype=torch.float16, device=dev))
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
        assert scales is not None and zeros is not None
        g = awq_linear.group_size
        awq_linear.scales = scales.clone().half()
        if linear.bias is not None:
            awq_linear.bias = linear.bias.clone().half()
        intweight = quantize_int_weights_kn(linear.weight.data, scales, zeros, g)  # [K, N]
        awq_linear.qweight = pack_rows_int4(intweight)
        awq_linear.qzeros = pack_rows_int4(zeros.to(torch.int32))
        return awq_linear

    def forward(self, x):
        out_shape = x.shape[:-1] + (self.out_features,)
        input_dtype = x.dtype
        if input_dtype != torch.float16:
            x = x.half()
        args = (x, self.qweight, self.qzeros, self.scales, self.w_bit, self.group_size, self.bias, self.out_features)
        if self.training:
            out = WQLinearMMFunction.apply(*args)
        else:
            with torch.no_grad():
                out = WQLinearMMFunction.apply(*args)
        if input_dtype != torch.float16:
            out = out.to(dtype=input_dtype)
        return out.reshape(out_shape)

    def extra_repr(self) -> str:
        return "in_features={}, out_features={}, bias={}, w_bit={}, group_size={}".format(
            self.in_features, self.out_features, self.bias is not None, self.w_bit, self.group_size)
