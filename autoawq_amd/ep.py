"""Expert-parallel sparse MoE over the stacked experts (new component: the reference stacks all experts of a
layer on ONE device -- awq/models/mixtral.py:130-158 -- and has no distributed code; SURVEY.md 8 f4.  The oracle
for it is the single-device result of awq/modules/fused/moe.py:45-91).

One process per GPU.  Rank r keeps a contiguous slice [e0, e1) of the stacked expert tensors (`ws` = w1|w3, `w2s`:
qweight [E, K, N/8], qzeros, scales -- what `fuse_linears(..., operation=torch.stack)` builds), i.e. 1/W of the MoE
weights, which is all of Mixtral's bulk (8 x 3 x 4096 x 14336 int4 per layer).  The activations and the router are
replicated (they are after a tensor-parallel attention block: its row-parallel all-reduce leaves every rank with the
same hidden states), so every rank computes the SAME top-k routing and then only the (token, expert) pairs whose
expert it owns:

    softmax + top-k over ALL experts and block alignment of the OWNED pairs in one launch (awq_moe_route_local;
    nothing is read back: a decode step stays hipGraph-capturable)
    grouped int4 GEMM (gate|up)  ->  silu_and_mul  ->  grouped GEMM (down, routing weights in the epilogue) into a
    zeroed output (rows of foreign pairs are never written)  ->  top-k sum
    (larger problems than the routing kernel takes: torch top-k, then the foreign pairs are parked in one extra
    bucket that sorts last and is cut off -- local_routing)

and ONE all-reduce of the [T, H] result (torch.distributed; backend "nccl" is RCCL over xGMI) gives every rank the
block output -- the same collective, at the same place, as a row-parallel dense MLP, so a decoder layer keeps two
all-reduces whatever the mix of dense and MoE blocks.  No all-to-all: with replicated activations there is nothing
to dispatch, and a bs=4 top-2 decode step moves 4 x 4096 fp16 = 32 KiB per rank.

Per-rank HBM traffic at decode is the weights of the OWNED experts that were hit, so the time of a step is set by
the busiest rank (tools/bench_moe.py --ep W reports it for the Mixtral shape).
"""
import torch
import torch.nn as nn

from . import ops
from .modules.fused import moe as _moe
from .tp import split_even_units


class ExpertShard(nn.Module):
    """Experts [e0, e1) of a stacked expert module: the same attribute names (`qweight`, `qzeros`, `scales`,
    `group_size`) the kernels read from the module `fuse_linears(..., operation=torch.stack)` returns."""

    def __init__(self, stacked, e0, e1):
        super().__init__()
        self.e0, self.e1 = e0, e1
        self.group_size = stacked.group_size
        self.register_buffer("qweight", stacked.qweight[e0:e1].contiguous())
        self.register_buffer("qzeros", stacked.qzeros[e0:e1].contiguous())
        self.register_buffer("scales", stacked.scales[e0:e1].contiguous())


def expert_bounds(num_experts, rank, world):
    """Contiguous, as even as possible: sizes differ by at most one expert."""
    s, c = split_even_units(num_experts, world)[rank]
    return s, s + c


def local_routing(topk_ids, e0, num_local, block_rows):
    """Block-aligned routing tensors for the pairs whose expert is in [e0, e0 + num_local), expert ids relative to
    e0; the other pairs are parked in one extra bucket that sorts last and is cut off.  Device-side only.
    Returns (sorted_token_ids, expert_ids, num_tokens_post_padded, owned mask [T, topk])."""
    owned = (topk_ids >= e0) & (topk_ids < e0 + num_local)
    ids = torch.where(owned, topk_ids - e0, torch.full_like(topk_ids, num_local))
    sorted_ids, expert_ids, n_post = ops.moe_align_block_size(ids, block_rows, num_local + 1)
    foreign = (~owned).sum()
    foreign_padded = (foreign + block_rows - 1) // block_rows * block_rows
    n_local = (n_post.to(torch.int64) - foreign_padded).to(torch.int32)
    # blocks past n_local never run; keep their expert index inside the shard all the same
    return sorted_ids, expert_ids.clamp(max=num_local - 1), n_local, owned


def apply_moe_weights_local(w1, w2, x, gating_output, topk, renormalize, e0):
    """This rank's part of apply_moe_weights (awq/modules/fused/moe.py:45-91): [T, H] partial sums over the pairs
    routed to experts [e0, e0 + w1.qweight.shape[0]); the sum over ranks is the reference result."""
    num_local = w1.qweight.shape[0]
    t1, t2 = getattr(w1, "decode_twin", None), getattr(w2, "decode_twin", None)
    if t1 is not None and t2 is not None and x.shape[0] * topk <= _moe.ROWS_MAX_PAIRS and x.is_cuda:
        # decode on GEMV-layout twins of the OWNED experts (ExpertShard.build_decode_twins): the router's global ids go to the
        # launch as they are, pairs of foreign experts are skipped there and their rows stay zero
        out = _moe._apply_moe_rows(t1, t2, x, gating_output, topk, renormalize, first_expert=e0)
        if out is not None:
            return out
    rows = _moe.DECODE_BLOCK_ROWS if x.shape[0] * topk <= _moe.DECODE_MAX_PAIRS else _moe.BLOCK_ROWS
    one_launch = gating_output.is_cuda and gating_output.shape[1] <= 64 and topk <= 8 and x.shape[0] <= 1024
    if one_launch:  # softmax + top-k over ALL experts (identical on every rank) + placement of the owned pairs, one kernel
        topk_weights, topk_ids, sorted_ids, expert_ids, n_local = ops.moe_route(gating_output, topk, renormalize, rows,
                                                                                first_expert=e0, num_local=num_local)
        owned = None
    else:
        topk_weights, topk_ids = ops.fused_topk(gating_output, topk, renormalize)
        sorted_ids, expert_ids, n_local, owned = local_routing(topk_ids, e0, num_local, rows)
    in_dtype = x.dtype
    xh = x.half() if in_dtype != torch.float16 else x
    xh = xh.view(xh.shape[0], 1, *xh.shape[1:])
    gate_up = ops.grouped_gemm_forward(xh, w1.qweight, w1.scales, w1.qzeros, topk_weights, sorted_ids, expert_ids, n_local,
                                       False, 8, block_rows=rows)
    act = torch.empty((gate_up.shape[:-1] + (gate_up.shape[-1] // 2,)), dtype=torch.float16, device=x.device)
    ops.silu_and_mul(gate_up, act)
    out = ops.grouped_gemm_forward(act, w2.qweight, w2.scales, w2.qzeros, topk_weights, sorted_ids, expert_ids, n_local,
                                   True, 8, block_rows=rows, zero_init=owned is None)  # rows of foreign pairs: never written
    if owned is not None:
        out = torch.where(owned.unsqueeze(-1), out, torch.zeros((), dtype=out.dtype, device=out.device))
    out = torch.sum(out, dim=1)
    return out.to(in_dtype) if in_dtype != torch.float16 else out


class ExpertParallelSparseMoeBlock(nn.Module):
    """Drop-in for FusedSparseMoeBlock(top_k, gate, ws, w2s) (awq/modules/fused/moe.py:12-42) on `world` ranks:
    keeps experts expert_bounds(E, rank, world) of the stacked modules and all-reduces the block output."""

    def __init__(self, top_k, gate, ws, w2s, rank, world, group=None, collective=None):
        super().__init__()
        self.collective = collective  # e.g. autoawq_amd.comm.OneShotAllReduce for decode-sized outputs; None: RCCL
        self.top_k, self.gate = top_k, gate
        self.rank, self.world, self.group = rank, world, group
        self.num_experts = ws.qweight.shape[0]
        if world > self.num_experts:
            raise ValueError(f"{world} ranks for {self.num_experts} experts: an expert is the indivisible unit here")
        self.e0, self.e1 = expert_bounds(self.num_experts, rank, world)
        self.ws, self.w2s = ExpertShard(ws, self.e0, self.e1), ExpertShard(w2s, self.e0, self.e1)
        if getattr(ws, "decode_twin", None) is not None and self.ws.qweight.is_cuda:
            _moe.build_decode_twins(self.ws, self.w2s)  # the unsharded block had decode twins: so does the shard (of ITS experts)

    def forward(self, hidden_states):
        batch_size, sequence_length, hidden_dim = hidden_states.shape
        x = hidden_states.view(-1, hidden_dim)
        router_logits = self.gate(x)
        out = apply_moe_weights_local(self.ws, self.w2s, x, router_logits, self.top_k, True, self.e0)
        if self.world > 1:
            import torch.distributed as dist

            small = self.collective is not None and out.numel() % 4 == 0 and out.numel() <= getattr(self.collective, "max_halfs", 0)
            if small and out.dtype == torch.float16 and out.is_contiguous():
                self.collective(out)  # one-shot xGMI all-reduce (csrc/allreduce.hip)
            elif dist.is_available() and dist.is_initialized():  # one process per GPU: RCCL over xGMI
                dist.all_reduce(out, group=self.group)
            else:  # (single-process tests add the ranks' partial sums themselves: apply_moe_weights_local)
                raise RuntimeError("ExpertParallelSparseMoeBlock: world > 1 but neither a collective nor an initialised process "
                                   "group: returning this rank's partial sum would be silently wrong")
        return out.view(batch_size, sequence_length, hidden_dim)


def shard_sparse_moe(block, rank, world, group=None):
    """FusedSparseMoeBlock -> this rank's ExpertParallelSparseMoeBlock (the block's stacked tensors are sliced, not
    copied whole: call it before moving the full stack to the device if the stack does not fit one GPU)."""
    return ExpertParallelSparseMoeBlock(block.top_k, block.gate, block.ws, block.w2s, rank, world, group=group)
