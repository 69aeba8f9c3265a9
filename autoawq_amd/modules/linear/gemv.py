"""WQLinear_GEMV for MI355X: the nn.Module surface of awq/modules/linear/gemv.py on gfx950 kernels.

Drop-in contract kept (reference awq/modules/linear/gemv.py:27-197):
  * ctor (w_bit, group_size, in_features, out_features, bias, dev); attribute split_k_iters = 8;
  * registered BUFFERS qweight [N, K/8] i32 (ordinal nibbles), qzeros [N, ZW] i32,
    scales [N, 8*ZW] f16 zero-padded, bias [N] f16 | None -- the keys of an AWQ "gemv" checkpoint;
    ZW = calculate_zeros_width(K, g) (:12-24);
  * from_linear(linear, w_bit, group_size, init_only=False, scales=None, zeros=None) (:77-154);
  * forward (:156-186): any leading dims, any float dtype (computed in fp16, cast back), bias added
    AFTER the cast back, in the input dtype (:183-185).
What differs by design: the arithmetic runs in libawq_hip.so on the module's own buffers at every batch
size (csrc/gemv_rows.hip up to 2 - 4 rows, csrc/gemv_batch.hip from there (5 rows; 4 while K > 2048; 3 while K > 6144) in launches of <= 128 rows -- gemv_lds.hip / gemv_nk.hip for the
group sizes it does not take; from PREFILL_MIN_ROWS rows PREFILL_IMPL); there is no CPU path -- a non-HIP tensor raises.
"""
import torch
import torch.nn as nn

from ... import _lib, ops
from ...utils.packing import (GEMV_ORDER, calculate_zeros_width, pack_rows_int4, pack_zeros_nk,
                              quantize_int_weights_nk)

def tensor_key(*tensors):
    """Identity AND content version of tensors a derived cache was built from: `data_ptr` changes when a buffer is re-assigned,
    `_version` when it is written in place (`load_state_dict`, `.copy_()`) -- ADVICE r03: a key of pointers alone went stale."""
    return tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in tensors)


def dequant_matmul_nk(x2d, wt):
    """The reference's own prefill route (dequantise, then a dense fp16 GEMM: awq/modules/linear/gemm.py:48-54) for the few
    shapes no hand-written kernel of the GEMV / GEMVFast layouts takes (K % 64, group sizes below 64 at prefill sizes):
    `wt` = the bit-exact fp16 W^T [N, K] from awq_dequantize_weights_gemv[_fast], a temporary of this call."""
    return torch.matmul(x2d, wt.t())


# Below this many rows the call runs the decode / batched-decode kernels on the layout's own buffers (csrc/gemv_batch.hip takes any M in
# launches of <= 128 rows, each streaming the matrix once: see prefill_min_rows; the reference switches to its batched kernel at 8
# rows: gemv.py:168).  From it: PREFILL_IMPL.
PREFILL_MIN_ROWS = 257


def prefill_min_rows(in_features):
    """Rows from which a call leaves the batched-decode kernel (csrc/gemv_batch.hip: one launch per <= 128 rows, the row parts across
    blocks of one XCD; more rows in balanced launches): 257 while one pass of a block's waves covers K (K <= 4096: at 4096 x 11008
    128 / 192 / 256 rows cost 31 / 52 / 63 us against 52 / 58 / 66 for dequantise + dense GEMM, 384 rows 89 against 67), 193 beyond
    (11008 x 4096: 128 / 192 rows 42 / 78 us against 62 / 83 for the best prefill route, 256 rows 82 against 70) --
    profiles/r06_prefill_routes.txt; ADVICE r05."""
    return PREFILL_MIN_ROWS if in_features <= 4096 else 193


def prefill_route(rows, in_features, out_features):
    """The "auto" prefill route, by measurement on MI355X (profiles/r06_prefill_routes.txt; ADVICE r05: the hand-written route as an
    unconditional default cost up to 47 % at the down-projection shape): the hand-written pair (packed words transposed into a
    temporary + the fused MFMA GEMM) where it is ahead of or within ~10 % of dequantise + dense fp16 GEMM -- from 3072 rows on
    matrices at least as wide as they are tall (4096 x 11008: 379 vs 435 us at 4096 rows, 1325 vs 1201 at 16384) -- and the
    reference's own two-pass route (awq/modules/linear/gemm.py:48-54) elsewhere (fewer rows: the fused kernel's K walk is latency-
    bound, 104 us at 128 ... 512 rows against 51 ... 72; tall matrices: the vendor's GEMM runs at 0.60 of the MFMA peak there)."""
    return "hand" if rows >= 3072 and out_features >= in_features else "two_pass"


class WQLinear_GEMV(nn.Module):
    PREFILL_IMPL = "auto"  # | "repack" | "two_pass" | "fused" (see forward)

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
        # Every batch size on this layout's OWN buffers (round 4: the second, GEMM-layout copy of every matrix that rounds 2-3
        # kept resident for prefill is gone).  Below PREFILL_MIN_ROWS rows: the decode / batched-decode kernels.  From there: PREFILL_IMPL --
        #   "auto" (default, round 6)  `prefill_route`: "repack" where it measures ahead of / level with "two_pass", else "two_pass";
        #   "repack" (round 5)    transpose the packed nibbles into a TEMPORARY of the call (csrc/repack.hip, K N / 2 bytes) and run
        #                         the fused MFMA GEMM on it (csrc/gemm_regb.hip): two hand-written launches, no vendor GEMM, no fp16
        #                         copy of the weights, bit-identical to what a GEMM-format checkpoint of the same weights computes;
        #   "two_pass"            dequantise (hand-written kernel, bit-exact) into an fp16 temporary + a dense fp16 GEMM: the reference's
        #                         own prefill route (gemm.py:48-54);
        #   "fused"               the register-decoded MFMA GEMM in its N-major form (AWQ_GEMV_KERNEL_PREFILL): no temporary, but
        #                         0.29 of the peak at M = 16384 and latency-bound below ~2000 rows (profiles/r04_bench_*.json).
        out = None
        rows = inputs.shape[0]
        decode = rows <= 16 or (rows < prefill_min_rows(self.in_features) and
                                ops.gemv_auto_kernel(rows, self.in_features, self.out_features, self.group_size)
                                == ops.GEMV_KERNEL_BATCH)  # (the older decode kernels serve 16 rows per launch: not worth chunking)
        impl = self.PREFILL_IMPL
        if impl == "auto":
            impl = "repack" if prefill_route(rows, self.in_features, self.out_features) == "hand" else "two_pass"
        if not decode and impl == "repack" and self.out_features % 8 == 0:  # (the GEMM layout packs eight columns per word)
            try:
                out = ops.gemv_prefill_repack(inputs, self.qweight, self.scales, self.qzeros, self.group_size)
            except _lib.AwqHipError as e:  # a shape the GEMM kernels refuse
                if e.code != _lib.ERR_UNSUPPORTED:
                    raise
        if not decode and impl == "fused":
            try:
                out = ops.gemv_forward(inputs, self.qweight, self.scales, self.qzeros, self.group_size,
                                       flags=ops.gemm_flags(kernel=ops.GEMV_KERNEL_PREFILL))
            except _lib.AwqHipError as e:  # K % 64, group sizes below 64
                if e.code != _lib.ERR_UNSUPPORTED:
                    raise
        if out is None and decode:
            try:
                out = ops.gemv_forward(inputs, self.qweight, self.scales, self.qzeros, self.group_size)
            except _lib.AwqHipError as e:  # a shape no decode kernel of this layout takes (K % 128 with odd group sizes)
                if e.code != _lib.ERR_UNSUPPORTED:
                    raise
        if out is None:
            out = dequant_matmul_nk(inputs, ops.dequantize_weights_gemv(self.qweight, self.scales, self.qzeros, self.group_size))
        if input_dtype != torch.float16:
            out = out.to(dtype=input_dtype)
        out = out + self.bias if self.bias is not None else out
        return out.reshape(out_shape)

    def extra_repr(self) -> str:
        return "in_features={}, out_features={}, bias={}, w_bit={}, group_size={}".format(
            self.in_features, self.out_features, self.bias is not None, self.w_bit, self.group_size)
