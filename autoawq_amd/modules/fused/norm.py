"""FasterTransformerRMSNorm for MI355X (reference: awq/modules/fused/norm.py:19-38).

Same constructor `(weight, eps)` and forward `norm(x)`; the arithmetic is `awq_rmsnorm_forward`
(csrc/decoder.hip) instead of `awq_ext.layernorm_forward_cuda`.  `forward(x, residual=r)` is the
MI355X-first extra: `r += x` (in place) and the norm of the sum in ONE launch, which is what a
decoder block does between attention and MLP (awq/modules/fused/block.py:108-119)."""
import torch
from torch import nn

from ... import ops


class FasterTransformerRMSNorm(nn.Module):
    def __init__(self, weight, eps=1e-6):
        super().__init__()
        self.weight = weight
        self.variance_epsilon = eps

    def forward(self, x, residual=None):
        in_dtype = x.dtype
        if in_dtype != torch.float16:
            x = x.half()
        w = self.weight if self.weight.dtype == torch.float16 else self.weight.half()
        out = ops.rmsnorm(x, w, self.variance_epsilon, residual=residual)
        return out if in_dtype == torch.float16 else out.to(in_dtype)
