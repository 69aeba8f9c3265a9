"""Fused sparse-MoE block for MI355X (reference: awq/modules/fused/moe.py:12-171).

`FusedSparseMoeBlock(top_k, gate, ws, w2s)` with `ws` / `w2s` the stacked expert modules built by
`fuse_linears` (awq/utils/fused_utils.py:145-162; Mixtral: w1|w3 concatenated on N, experts stacked
on dim 0 -- awq/models/mixtral.py:130-158).  `apply_moe_weights`, `moe_align_block_size`,
`fused_topk` keep the reference signatures.  Everything stays on the device (no routing data is
read back), so a decode step is hipGraph-capturable: softmax/top-k (torch), block alignment
(torch, sort based), ONE grouped int4 GEMM for gate|up over the experts that were hit, ONE
silu_and_mul, ONE grouped GEMM for the down projection with the routing weights folded into
its epilogue, and the top-k sum.
"""
from typing import Dict

import torch

from ... import ops

BLOCK_ROWS = 16        # token rows per grouped-GEMM block (moe.py:55: moe_align_block_size(topk_ids, 16, E))
DECODE_BLOCK_ROWS = 8  # decode-sized batches: 8-row blocks run the selector-row kernel (one MFMA per fragment)
DECODE_MAX_PAIRS = 64


class FusedSparseMoeBlock(torch.nn.Module):
    def __init__(self, top_k, gate, ws, w2s):
        super().__init__()
        self.gate = gate
        self.top_k = top_k
        self.ws = ws
        self.w2s = w2s

    def forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
        batch_size, sequence_length, hidden_dim = hidden_states.shape
        hidden_states = hidden_states.view(-1, hidden_dim)
        router_logits = self.gate(hidden_states)
        out = apply_moe_weights(self.ws, self.w2s, hidden_states, router_logits, self.top_k, renormalize=True)
        return out.view(batch_size, sequence_length, hidden_dim)


def apply_moe_weights(w1: Dict[str, torch.Tensor], w2: Dict[str, torch.Tensor], x: torch.Tensor,
                      gating_output: torch.Tensor, topk: int, renormalize: bool) -> torch.Tensor:
    num_experts = w1.qweight.shape[0]
    rows = DECODE_BLOCK_ROWS if x.shape[0] * topk <= DECODE_MAX_PAIRS else BLOCK_ROWS
    if num_experts <= 64 and topk <= 8 and x.shape[0] <= 1024:  # one-launch routing
        topk_weights, topk_ids, sorted_token_ids, expert_ids, num_tokens_post_padded = ops.moe_route(
            gating_output, topk, renormalize, rows)
    else:
        topk_weights, topk_ids = fused_topk(gating_output, topk, renormalize)
        sorted_token_ids, expert_ids, num_tokens_post_padded = moe_align_block_size(topk_ids, rows, num_experts)
    in_dtype = x.dtype
    xh = x.half() if in_dtype != torch.float16 else x
    xh = xh.view(xh.shape[0], 1, *xh.shape[1:])
    gate_up = ops.grouped_gemm_forward(xh, w1.qweight, w1.scales, w1.qzeros, topk_weights, sorted_token_ids, expert_ids,
                                       num_tokens_post_padded, False, 8, block_rows=rows)
    out = torch.empty((gate_up.shape[:-1] + (gate_up.shape[-1] // 2,)), dtype=torch.float16, device=x.device)
    ops.silu_and_mul(gate_up, out)
    out = ops.grouped_gemm_forward(out, w2.qweight, w2.scales, w2.qzeros, topk_weights, sorted_token_ids, expert_ids,
                                   num_tokens_post_padded, True, 8, block_rows=rows)
    out = torch.sum(out, dim=1)
    return out.to(in_dtype) if in_dtype != torch.float16 else out


def moe_align_block_size(topk_ids: torch.Tensor, block_size: int, num_experts: int):
    return ops.moe_align_block_size(topk_ids, block_size, num_experts)


def fused_topk(gating_output: torch.Tensor, topk: int, renormalize: bool):
    return ops.fused_topk(gating_output, topk, renormalize)
