"""WQLinear_GEMVFast for MI355X: the nn.Module surface of awq/modules/linear/gemv_fast.py.

Drop-in contract kept (reference gemv_fast.py:68-208):
  * ctor (w_bit, group_size, in_features, out_features, bias, dev); attributes split_k_iters = 8,
    interleave = 4;
  * BUFFERS qweight [N/4, K] int16 (4-row interleaved), scales [8*ZW, N] fp16 (group major),
    qzeros [8*ZW, N] fp16 = -(scale*zero), bias [N] fp16 | None;
  * from_linear(linear, w_bit, group_size, init_only=False, scales=None, zeros=None);
  * forward needs a 3-D input [batch, tokens, K] (gemv_fast.py:190), bias added afterwards.
The reference picks its decode kernel for batch < 8 and one token, its prefill GEMM otherwise (gemv_fast.py:191-206); here up to 128
rows (64 for K > 4096) run the batched decode kernel on the layout's own buffers (csrc/gemv_batch.hip, GEMVFast form; csrc/gemv_fast.hip for group
sizes other than 128), and prefill-sized inputs the hand-written pair `awq_gemv_fast_prefill` (round 6): the packed words transposed
into a temporary of the call (csrc/repack.hip) + the register-decoded MFMA GEMM with this format's own scales / fp16 zero terms
(csrc/gemm_regb.hip, FZ form) -- `PREFILL_IMPL = "fused"`.  The default, "auto", takes that pair where it measures ahead of or
level with dequantise + dense fp16 GEMM (the reference's gemm.py:48-54 route, `"two_pass"`) and the two-pass route elsewhere
(`gemv.prefill_route`, profiles/r06_prefill_routes.txt).
"""
import torch

from ... import _lib, ops
from ...utils.packing import calculate_zeros_width, pack_intweight_fast, quantize_int_weights_nk
from .gemv import dequant_matmul_nk, prefill_min_rows, prefill_route

# below this many rows: the decode / batched-decode kernels (csrc/gemv_batch.hip in its GEMVFast form, launches of <= 128 rows, group
# size 128; other group sizes: 16 rows, csrc/gemv_fast.hip); from it: PREFILL_IMPL
PREFILL_MIN_ROWS = 257  # (gemv.prefill_min_rows: 257 while K <= 4096, 193 beyond)


class WQLinear_GEMVFast(torch.nn.Module):
    # "auto" (default): `gemv.prefill_route` -- the hand-written pair where it measures ahead of / level with dequantise + dense fp16 GEMM
    # (from 3072 rows on matrices at least as wide as tall), the two-pass route elsewhere; "fused" / "two_pass" force one
    PREFILL_IMPL = "auto"

    def __init__(self, w_bit, group_size, in_features, out_features, bias, dev):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.w_bit = w_bit
        self.group_size = group_size if group_size != -1 else in_features
        self.split_k_iters = 8
        self.interleave = 4

        assert self.in_features % self.group_size == 0
        assert out_features % (32 // self.w_bit) == 0
        pack_num = 32 // self.w_bit
        int16_pack_num = 16 // self.w_bit
        assert out_features % self.interleave == 0
        zw = calculate_zeros_width(in_features, self.group_size)
        self.register_buffer("qweight", torch.zeros((out_features // self.interleave,
                                                     in_features // int16_pack_num * self.interleave),
                                                    dtype=torch.int16, device=dev))
        self.register_buffer("scales", torch.zeros((zw * pack_num, out_features), dtype=torch.float16, device=dev))
        self.register_buffer("qzeros", torch.zeros((zw * pack_num, out_features), dtype=torch.float16, device=dev))
        if bias:
            self.register_buffer("bias", torch.zeros((out_features), dtype=torch.float16, device=dev))
        else:
            self.bias = None

    @classmethod
    def from_linear(cls, linear, w_bit, group_size, init_only=False, scales=None, zeros=None):
        awq_linear = cls(w_bit, group_size, linear.in_features, linear.out_features, linear.bias is not None,
                         linear.weight.device)
        if init_only:
            return awq_linear
        assert scales is not None and zeros is not None  # both [N, G]
        zw = calculate_zeros_width(linear.in_features, group_size)
        qscales = torch.zeros((scales.shape[0], zw * 8), dtype=torch.float16, device=scales.device)
        qscales[:, : scales.shape[1]] = scales
        awq_linear.scales = qscales.transpose(1, 0).contiguous()
        if linear.bias is not None:
            awq_linear.bias = linear.bias.clone().half()
        intweight = quantize_int_weights_nk(linear.weight.data, scales, zeros, qscales, group_size)
        awq_linear.qweight = pack_intweight_fast(intweight.contiguous())
        qzeros = torch.zeros_like(qscales)
        G = scales.shape[1]
        qzeros[:, :G] = -(qscales[:, :G] * zeros.to(torch.int32).to(torch.float32)).to(torch.float16)
        awq_linear.qzeros = qzeros.transpose(1, 0).contiguous()
        return awq_linear

    @torch.no_grad()
    def forward(self, x):
        batch_size, n_tokens, _ = x.shape  # 3-D input required, like the reference
        inputs = x.reshape(-1, x.shape[-1])
        in_dtype = inputs.dtype
        if in_dtype != torch.float16:
            inputs = inputs.half()
        out = None
        rows = inputs.shape[0]
        if (rows <= 16 or (rows < prefill_min_rows(self.in_features) and self.group_size == 128 and self.in_features % 128 == 0)) and self.out_features % 16 == 0:
            try:
                out = ops.gemv_fast_forward(inputs, self.qweight, self.scales, self.qzeros, self.group_size)
            except _lib.AwqHipError as e:  # a shape the decode kernel does not take (K % 128, unusual group sizes)
                if e.code != _lib.ERR_UNSUPPORTED:
                    raise
        impl = self.PREFILL_IMPL
        if impl == "auto":
            impl = "fused" if prefill_route(rows, self.in_features, self.out_features) == "hand" else "two_pass"
        if out is None and impl == "fused":
            # prefill-sized batches: two hand-written launches on THIS layout's buffers (round 6) -- the words transposed into a
            # temporary of the call, then the fused MFMA GEMM with the layout's w * s + qzeros arithmetic (the same effective weights
            # the decode kernel uses).  Nothing resident, no vendor GEMM.
            try:
                out = ops.gemv_fast_prefill(inputs, self.qweight, self.scales, self.qzeros, self.group_size)
            except _lib.AwqHipError as e:  # K % 64, group sizes below 64, N % 8: the two-pass route below
                if e.code != _lib.ERR_UNSUPPORTED:
                    raise
        if out is None:
            # the reference's own two-pass route (awq/modules/linear/gemm.py:48-54): dequantise (hand-written kernel) into a temporary +
            # a dense fp16 GEMM; explicit (`PREFILL_IMPL = "two_pass"`) or for the shapes the fused kernel refuses
            wt = ops.dequantize_weights_gemv_fast(self.qweight, self.scales, self.qzeros, self.group_size)
            out = dequant_matmul_nk(inputs, wt)
        if in_dtype != torch.float16:
            out = out.to(in_dtype)
        out = out.reshape(batch_size, n_tokens, self.out_features)
        return out + self.bias if self.bias is not None else out

    def extra_repr(self) -> str:
        return "in_features={}, out_features={}, bias={}, w_bit={}, group_size={}".format(
            self.in_features, self.out_features, self.bias is not None, self.w_bit, self.group_size)
