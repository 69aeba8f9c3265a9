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
# silu(gate) * up applied by the w2 grouped GEMM while it stages its activations (awq_grouped_gemm_forward_ex,
# AWQ_GEMM_FLAG_X_GATED_SILU): one launch and one [pairs, I] round trip less, bit-identical -- but every one of the 16 column
# tiles of w2 re-evaluates the activation for its K slice, and at Mixtral's bs = 4 that costs more than the launch it saves
# (144 us per block against 133 us: tools/dbg_moe_ab.py, profiles/r03_moe_silu_fold_ab.txt).  Off by default.
FUSE_ACTIVATION_INTO_W2 = False


# Decode on GEMV-layout twins of the expert stacks (round 6): up to this many (token, expert) pairs every pair runs as one
# batch-1 call of the row-streaming kernel, all pairs in one launch per projection (ops.grouped_gemv_forward); beyond, or
# without twins, the GEMM-layout grouped kernel.  Mixtral shape, top-2, us per block rows / GEMM-layout grouped kernel
# (tools/bench_moe.py --sweep, profiles/r06_moe_rows.txt): 1 token 47.6 / 69.9, 2: 79.2 / 110.5, 4: 113.7 / 131.8,
# 8 tokens (16 pairs, all 8 experts) 205 / 193 -- every pair decodes its expert's weights again (HBM traffic follows the
# DISTINCT experts, the VALU work the pairs), so the hand-over sits between 8 and 16 pairs.
ROWS_MAX_PAIRS = 12
ROWS_PARTS = (0, 0)  # blocks one expert matrix is dealt over in the w1|w3 / w2 launch (0 = the kernel's default)


def build_decode_twins(ws, w2s):
    """GEMV-layout copies of the stacked experts for the decode path: `ws` (w1|w3 concatenated on N) with its gate / up columns
    interleaved as row pairs, `w2s` as it is.  A SECOND resident copy of the experts (the GEMM-layout stacks stay: prefill-
    sized token counts read those), built once at load -- `fuse_mixtral(decode_layout="auto")` -- and attached as
    `ws.decode_twin` / `w2s.decode_twin`; rebuild after re-assigning or writing the stacks' buffers."""
    from ...utils.convert import gemm_stack_to_gemv

    K1, K2 = ws.qweight.shape[1], w2s.qweight.shape[1]
    for K, G in ((K1, ws.qzeros.shape[1]), (K2, w2s.qzeros.shape[1])):
        if (K // G) % 128 or K > 16384:  # shapes the row-streaming kernel does not take: no twins, the GEMM-layout kernel serves
            ws.decode_twin = w2s.decode_twin = None
            return None, None
    ws.decode_twin = gemm_stack_to_gemv(ws.qweight, ws.qzeros, ws.scales, interleave_halves=True)
    w2s.decode_twin = gemm_stack_to_gemv(w2s.qweight, w2s.qzeros, w2s.scales)
    return ws.decode_twin, w2s.decode_twin


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


# From this many (token, expert) pairs an expert sees GEMM-sized batches: one fused GEMM per expert.  Mixtral shape, top-2
# (tools/dbg_moe_sizes.py): 256 pairs 755 us (blocks) vs 812 us (per expert), 512 pairs 1265 vs 1003 us.
PREFILL_MIN_PAIRS = 384
FUSED_PREFILL_GLUE = True  # round 6: routing, sort, gather and scatter inside the launches (False: the torch glue of rounds 4-5)


def _apply_moe_prefill(w1, w2, x, gating_output, topk, renormalize):
    """Prefill-sized token counts (round 4): the pairs are sorted by expert ON THE DEVICE, and each projection is ONE launch of
    the register-decoded MFMA GEMM whose row tiles are dealt over the experts from the device-side row offsets
    (awq_grouped_gemm_prefill) -- w1|w3 over all experts, silu * mul, w2 over all experts -- then the routing weight is
    applied per pair and the top-k slots are summed exactly like the decode path (`[T, topk, H]` fp16, sum over dim 1).
    Nothing is read back to the host (round 3 read the per-expert row counts and launched one GEMM per expert and
    projection: 16 launches of ~112 blocks each for Mixtral, and no hipGraph capture)."""
    T, H = x.shape
    E = w1.qweight.shape[0]
    if FUSED_PREFILL_GLUE and E <= 64 and topk <= 8:
        # round 6: the sort stays an index list -- routing (one launch, a wave per token), a counting sort of the pairs (one
        # launch), w1|w3 reading the TOKENS' rows through the list, silu * mul, w2 writing each pair's row in pair order with
        # its routing weight folded into the one rounding (moe.py:84-88), the top-k sum: 6 launches for ~22
        topk_weights, topk_ids = ops.moe_route(gating_output, topk, renormalize, 0)[:2]
        order, seg = ops.moe_sort_pairs(topk_ids, E)
        try:
            gate_up = ops.grouped_gemm_prefill_ex(x, w1.qweight, w1.scales, w1.qzeros, seg, order, x_div=topk, gather=True)
            out = ops.grouped_gemm_prefill_ex(ops.silu_and_mul(gate_up), w2.qweight, w2.scales, w2.qzeros, seg, order, scatter=True,
                                              pair_weights=topk_weights)
            return out.view(T, topk, H).sum(dim=1)
        except ops._lib.AwqHipError as e:  # shapes the register-decoded kernel does not take: the older path below
            if getattr(e, "code", 0) != -3:
                raise
    topk_weights, topk_ids = fused_topk(gating_output, topk, renormalize)
    flat_e = topk_ids.reshape(-1).long()
    order = torch.argsort(flat_e, stable=True)
    counts = torch.zeros(E, dtype=torch.int32, device=x.device).scatter_add_(0, flat_e, torch.ones_like(flat_e, dtype=torch.int32))
    seg = torch.zeros(E + 1, dtype=torch.int32, device=x.device)
    seg[1:] = torch.cumsum(counts, 0)
    xs = x.index_select(0, order // topk)                      # [T * topk, H] rows grouped by expert
    try:
        gate_up = ops.grouped_gemm_prefill(xs, w1.qweight, w1.scales, w1.qzeros, seg)
        ys = ops.grouped_gemm_prefill(ops.silu_and_mul(gate_up), w2.qweight, w2.scales, w2.qzeros, seg)
    except ops._lib.AwqHipError as e:  # shapes the register-decoded kernel does not take (K % 64, group sizes below 64)
        if getattr(e, "code", 0) != -3 or torch.cuda.is_current_stream_capturing():
            raise
        ys = _per_expert_gemms(w1, w2, xs, counts)
    w_sorted = topk_weights.reshape(-1).index_select(0, order).to(torch.float32)
    out = torch.empty((T * topk, H), dtype=torch.float16, device=x.device)
    out.index_copy_(0, order, (ys.float() * w_sorted[:, None]).half())
    return out.view(T, topk, H).sum(dim=1)


def _apply_moe_rows(t1, t2, x, gating_output, topk, renormalize, first_expert=0):
    """Decode on the GEMV-layout twins: softmax / top-k, ONE launch of the row-streaming kernel for w1|w3 over all pairs with
    silu(gate) * up written by the launch itself, ONE for w2 with the routing weight in its epilogue, the top-k sum.  No
    alignment pass (no blocks: every pair is a batch-1 call), nothing read back.  None: a shape the kernel does not take.
    first_expert (expert-parallel shards, autoawq_amd/ep.py): the twins hold experts [first_expert, first_expert + E) of the
    router's global ids; the rows of the other pairs stay zero, i.e. the result is this rank's partial sum."""
    in_dtype = x.dtype
    xh = x.half() if in_dtype != torch.float16 else x
    num_experts = gating_output.shape[1]
    local = first_expert != 0 or t1.qweight.shape[0] != num_experts
    if num_experts <= 64 and topk <= 8 and x.shape[0] <= 1024:
        topk_weights, topk_ids = ops.moe_route(gating_output, topk, renormalize, 0)[:2]  # routing only: no alignment pass
    else:
        topk_weights, topk_ids = fused_topk(gating_output, topk, renormalize)
    try:
        act = ops.grouped_gemv_forward(xh, t1.qweight, t1.scales, t1.qzeros, topk_ids, t1.group_size, silu_pairs=True, parts=ROWS_PARTS[0],
                                       first_expert=first_expert, zero_init=local)
        out = ops.grouped_gemv_forward(act.view(-1, act.shape[-1]), t2.qweight, t2.scales, t2.qzeros, topk_ids, t2.group_size,
                                       topk_weights=topk_weights, parts=ROWS_PARTS[1], first_expert=first_expert, zero_init=local)
    except ops._lib.AwqHipError as e:
        if getattr(e, "code", 0) != -3:
            raise
        return None
    out = torch.sum(out, dim=1)
    return out.to(in_dtype) if in_dtype != torch.float16 else out


def _per_expert_gemms(w1, w2, xs, counts):
    """Fallback of the prefill path for odd shapes: one awq_gemm_forward per expert and projection (reads the counts back)."""
    ys = torch.empty((xs.shape[0], w2.qweight.shape[2] * 8), dtype=torch.float16, device=xs.device)
    off = 0
    for e, n in enumerate(counts.cpu().tolist()):
        if n == 0:
            continue
        gate_up = ops.gemm_forward(xs[off:off + n], w1.qweight[e], w1.scales[e], w1.qzeros[e])
        ys[off:off + n] = ops.gemm_forward(ops.silu_and_mul(gate_up), w2.qweight[e], w2.scales[e], w2.qzeros[e])
        off += n
    return ys


def apply_moe_weights(w1: Dict[str, torch.Tensor], w2: Dict[str, torch.Tensor], x: torch.Tensor,
                      gating_output: torch.Tensor, topk: int, renormalize: bool) -> torch.Tensor:
    num_experts = w1.qweight.shape[0]
    if x.shape[0] * topk >= PREFILL_MIN_PAIRS and x.is_cuda:
        in_dtype = x.dtype
        out = _apply_moe_prefill(w1, w2, x.half() if in_dtype != torch.float16 else x, gating_output, topk, renormalize)
        return out.to(in_dtype) if in_dtype != torch.float16 else out
    t1, t2 = getattr(w1, "decode_twin", None), getattr(w2, "decode_twin", None)
    if t1 is not None and t2 is not None and x.shape[0] * topk <= ROWS_MAX_PAIRS and x.is_cuda:
        out = _apply_moe_rows(t1, t2, x, gating_output, topk, renormalize)
        if out is not None:
            return out
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
    if FUSE_ACTIVATION_INTO_W2:
        # silu(gate) * up is applied by the w2 grouped GEMM while it stages its activations (bit-identical to the separate
        # awq_silu_and_mul launch of moe.py:73-76): one launch and one [pairs, I] round trip less
        out = ops.grouped_gemm_forward(gate_up, w2.qweight, w2.scales, w2.qzeros, topk_weights, sorted_token_ids, expert_ids,
                                       num_tokens_post_padded, True, 8, block_rows=rows, x_gated=True)
    else:
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
