"""Tensor-parallel sharding of GEMM-layout WQLinear buffers (new component: the reference has no
distributed code at all -- SURVEY.md 2.3 / 8e; the oracle for it is the unsharded result).

Legal split axes follow from how the reference itself concatenates packed tensors
(awq/utils/fused_utils.py:87-96):
  * column-parallel (qkv, gate/up): slice output columns at multiples of 8 -- qweight[:, n0/8:n1/8],
    qzeros[:, n0/8:n1/8], scales[:, n0:n1], bias[n0:n1]; no communication;
  * row-parallel (o_proj, down_proj): slice input rows at multiples of group_size --
    qweight[k0:k1], qzeros[k0/g:k1/g], scales[k0/g:k1/g]; partial sums need ONE all-reduce;
    bias is added once (rank 0).
Shards only need to be group aligned, not equal: Llama-2-7B down_proj has 86 groups, so TP=8 is
6 ranks x 11 groups + 2 ranks x 10 groups, with the matching uneven column split of gate/up.
One process per GPU; the all-reduce is torch.distributed (backend "nccl" = RCCL over xGMI).
"""
import torch
import torch.nn as nn

from .modules.linear.gemm import WQLinear_GEMM
from .modules.linear.gemv import WQLinear_GEMV
from .utils.packing import GEMV_ORDER, calculate_zeros_width, pack_rows_int4


def split_even_units(total_units, world):
    """Contiguous split of `total_units` indivisible units over `world` ranks, sizes differ by <= 1.
    Returns [(start, count)] per rank."""
    base, rem = divmod(total_units, world)
    out, start = [], 0
    for r in range(world):
        cnt = base + (1 if r < rem else 0)
        out.append((start, cnt))
        start += cnt
    return out


def column_shard(qweight, qzeros, scales, bias, n0, n1):
    assert n0 % 8 == 0 and n1 % 8 == 0, "column shards must be multiples of 8 output columns"
    return (qweight[:, n0 // 8:n1 // 8].contiguous(), qzeros[:, n0 // 8:n1 // 8].contiguous(),
            scales[:, n0:n1].contiguous(), None if bias is None else bias[n0:n1].contiguous())


def row_shard(qweight, qzeros, scales, k0, k1, group_size):
    assert k0 % group_size == 0 and k1 % group_size == 0, "row shards must be whole groups"
    g0, g1 = k0 // group_size, k1 // group_size
    return qweight[k0:k1].contiguous(), qzeros[g0:g1].contiguous(), scales[g0:g1].contiguous()


# ---- the GEMV layout (qweight [N, K/8], qzeros [N, ZW], scales [N, 8 ZW]: awq/modules/linear/gemv.py:45-69)
def column_shard_gemv(qweight, qzeros, scales, bias, n0, n1):
    """Output columns are ROWS of every GEMV-layout buffer: a plain row slice, any n0 / n1."""
    return (qweight[n0:n1].contiguous(), qzeros[n0:n1].contiguous(), scales[n0:n1].contiguous(),
            None if bias is None else bias[n0:n1].contiguous())


def row_shard_gemv(qweight, qzeros, scales, k0, k1, group_size):
    """Input rows [k0, k1) (whole groups): the packed K words k0/8 .. k1/8 of every row, and the groups'
    zero points / scales re-packed to the zeros width of the SHARD's in_features (the padding rule of
    calculate_zeros_width depends on K, gemv.py:12-24)."""
    assert k0 % group_size == 0 and k1 % group_size == 0, "row shards must be whole groups"
    g0, g1 = k0 // group_size, k1 // group_size
    N = qweight.shape[0]
    zw = calculate_zeros_width(k1 - k0, group_size)
    shifts = torch.arange(0, 32, 4, device=qzeros.device, dtype=torch.int32)
    znib = ((qzeros.unsqueeze(-1) >> shifts) & 0xF).reshape(N, -1)[:, g0:g1]
    zpad = torch.zeros((N, zw * 8), dtype=torch.int32, device=qzeros.device)
    zpad[:, : g1 - g0] = znib
    spad = torch.zeros((N, zw * 8), dtype=scales.dtype, device=scales.device)
    spad[:, : g1 - g0] = scales[:, g0:g1]
    return qweight[:, k0 // 8:k1 // 8].contiguous(), pack_rows_int4(zpad, GEMV_ORDER), spad


def _module_from_gemv(qweight, qzeros, scales, bias, group_size):
    N, K = qweight.shape[0], qweight.shape[1] * 8
    m = WQLinear_GEMV(4, group_size, K, N, bias is not None, qweight.device)
    m.qweight, m.qzeros, m.scales = qweight, qzeros, scales
    if bias is not None:
        m.bias = bias
    return m


def _module_from(qweight, qzeros, scales, bias, group_size):
    K, N = qweight.shape[0], qweight.shape[1] * 8
    m = WQLinear_GEMM(4, group_size, K, N, bias is not None, qweight.device)
    m.qweight, m.qzeros, m.scales = qweight, qzeros, scales
    if bias is not None:
        m.bias = bias
    return m


class ColumnParallelWQLinear(nn.Module):
    """Holds this rank's output-column slice (GEMM or GEMV layout); forward returns the local slice (no collective)."""

    def __init__(self, full, rank, world, unit=8, bounds=None):
        super().__init__()
        N = full.out_features
        if bounds is None:
            assert N % unit == 0
            s, c = split_even_units(N // unit, world)[rank]
            bounds = (s * unit, (s + c) * unit)
        self.bounds = bounds
        if isinstance(full, WQLinear_GEMV):
            self.shard = _module_from_gemv(*column_shard_gemv(full.qweight, full.qzeros, full.scales, full.bias, *bounds),
                                           full.group_size)
        else:
            self.shard = _module_from(*column_shard(full.qweight, full.qzeros, full.scales, full.bias, *bounds),
                                      full.group_size)

    def forward(self, x):
        return self.shard(x)


class RowParallelWQLinear(nn.Module):
    """Holds this rank's input-row slice (whole groups; GEMM or GEMV layout); forward all-reduces the partial sums."""

    def __init__(self, full, rank, world, bounds=None, group=None, collective=None):
        """collective: a callable summing a contiguous fp16 tensor over the ranks in place -- e.g. an
        autoawq_amd.comm.OneShotAllReduce (one kernel launch, hipGraph-capturable; decode-sized outputs) -- or None for
        torch.distributed.all_reduce (RCCL)."""
        super().__init__()
        self.collective = collective
        g = full.group_size
        if bounds is None:
            s, c = split_even_units(full.in_features // g, world)[rank]
            bounds = (s * g, (s + c) * g)
        self.bounds = bounds
        self.rank, self.world, self.group = rank, world, group
        if isinstance(full, WQLinear_GEMV):
            qw, qz, sc = row_shard_gemv(full.qweight, full.qzeros, full.scales, bounds[0], bounds[1], g)
            self.shard = _module_from_gemv(qw, qz, sc, full.bias if rank == 0 else None, g)
        else:
            qw, qz, sc = row_shard(full.qweight, full.qzeros, full.scales, bounds[0], bounds[1], g)
            self.shard = _module_from(qw, qz, sc, full.bias if rank == 0 else None, g)

    def forward(self, x_local):
        y = self.shard(x_local)
        if self.world > 1:
            import torch.distributed as dist

            small = self.collective is not None and y.numel() % 4 == 0 and y.numel() <= getattr(self.collective, "max_halfs", 0)
            if small and y.dtype == torch.float16 and y.is_contiguous():
                self.collective(y)  # one-shot xGMI all-reduce (csrc/allreduce.hip)
            elif dist.is_available() and dist.is_initialized():  # one process per GPU: RCCL over xGMI
                dist.all_reduce(y, group=self.group)
            else:
                raise RuntimeError("RowParallelWQLinear: world > 1 but neither a collective nor an initialised process group: "
                                   "returning this rank's partial sum would be silently wrong")
        return y


# ---- a whole decoder layer: which columns / rows each rank owns
def llama_layer_bounds(n_heads, n_kv_heads, head_dim, intermediate, group_size, rank, world):
    """Megatron-style split of one Llama-style decoder layer.  Attention is split by HEADS: a rank keeps whole query heads
    and the whole KV head(s) they attend to, so GQA groups stay together.
      * n_kv_heads % world == 0: a rank owns n_kv_heads / world KV heads and all their query heads;
      * world % n_kv_heads == 0 (fewer KV heads than ranks, e.g. 4 KV heads at TP = 8): KV-head REPLICATION -- each KV head
        is kept by world / n_kv_heads ranks, which divide its query heads among themselves (the k / v projections and the
        cache rows of that head exist on each of them; the query heads, hence the o_proj rows, are still a partition, so the
        all-reduce after o_proj is unchanged).  `kv_replicas` says how many ranks hold this rank's KV head.
    The MLP is split by whole quantisation GROUPS of the down projection's input rows, gate / up columns following the same
    bounds (uneven splits allowed: 86 groups over 8 ranks = 6 x 11 + 2 x 10).  Returns a dict of [start, stop) bounds."""
    if n_heads % n_kv_heads:
        raise ValueError(f"n_heads {n_heads} must be a multiple of n_kv_heads {n_kv_heads}")
    gq = n_heads // n_kv_heads
    if n_kv_heads % world == 0:
        rep = 1
        kv0, kvc = split_even_units(n_kv_heads, world)[rank]
        h0, h1 = kv0 * gq, (kv0 + kvc) * gq
    elif world % n_kv_heads == 0 and gq % (world // n_kv_heads) == 0:
        rep = world // n_kv_heads
        kv0, kvc = rank // rep, 1
        per = gq // rep                                  # query heads of this KV head per replica
        h0 = kv0 * gq + (rank % rep) * per
        h1 = h0 + per
    else:
        raise ValueError(f"n_kv_heads {n_kv_heads} (x {gq} query heads each) cannot be split or replicated over {world} ranks: "
                         "need n_kv_heads % world == 0, or world % n_kv_heads == 0 with the query heads of a KV head "
                         "divisible by world / n_kv_heads")
    q0, q1 = h0 * head_dim, h1 * head_dim
    k0, k1 = kv0 * head_dim, (kv0 + kvc) * head_dim
    if intermediate % group_size:
        raise ValueError("intermediate size must be a multiple of the group size")
    g0, gc = split_even_units(intermediate // group_size, world)[rank]
    i0, i1 = g0 * group_size, (g0 + gc) * group_size
    return {"q": (q0, q1), "kv": (k0, k1), "heads": (h0, h1), "kv_heads": (kv0, kv0 + kvc), "kv_replicas": rep,
            "mlp": (i0, i1)}


def shard_llama_layer(q_proj, k_proj, v_proj, o_proj, gate_proj, up_proj, down_proj, n_heads, n_kv_heads, head_dim,
                      rank, world, group=None):
    """This rank's slices of the seven GEMM-layout Linears of a decoder layer: q / k / v / gate / up
    column-parallel (plain WQLinear_GEMM modules, no communication), o / down row-parallel
    (RowParallelWQLinear: one all-reduce each, bias on rank 0)."""
    b = llama_layer_bounds(n_heads, n_kv_heads, head_dim, gate_proj.out_features, down_proj.group_size, rank, world)

    def col(m, lo, hi):
        if isinstance(m, WQLinear_GEMV):
            return _module_from_gemv(*column_shard_gemv(m.qweight, m.qzeros, m.scales, m.bias, lo, hi), m.group_size)
        return _module_from(*column_shard(m.qweight, m.qzeros, m.scales, m.bias, lo, hi), m.group_size)

    return {
        "q_proj": col(q_proj, *b["q"]), "k_proj": col(k_proj, *b["kv"]), "v_proj": col(v_proj, *b["kv"]),
        "o_proj": RowParallelWQLinear(o_proj, rank, world, bounds=b["q"], group=group),
        "gate_proj": col(gate_proj, *b["mlp"]), "up_proj": col(up_proj, *b["mlp"]),
        "down_proj": RowParallelWQLinear(down_proj, rank, world, bounds=b["mlp"], group=group),
        "bounds": b,
    }
