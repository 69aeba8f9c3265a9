"""Expert-parallel sparse MoE (autoawq_amd/ep.py) on two gloo ranks: each rank keeps half of the stacked experts,
routes identically, computes only the pairs of its own experts and all-reduces the block output -- which must be the
single-device result of the reference's apply_moe_weights (awq/modules/fused/moe.py:45-91, restated by
oracle.moe_forward).  The arithmetic of the grouped GEMMs is done by the CPU oracle here (no GPU); on the GPU the same
routing tensors feed the HIP kernels (tests/test_gpu_ep.py)."""
import os
import sys
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def oracle_moe_ops(ops, oracle):
    """ops.grouped_gemm_forward / silu_and_mul on the CPU oracle (test infrastructure; the product never does this)."""

    def grouped_gemm_forward(x, qweight, scales, qzeros, topk_weights, sorted_token_ids, expert_ids, num_tokens_post_padded,
                             mul_weights, split_k_iters=8, block_rows=16, zero_init=False):
        T, topk = topk_weights.shape
        E, K, NW = qweight.shape
        g = K // qzeros.shape[1]
        x2 = x.reshape(-1, K)
        x_div = topk if x.shape[1] == 1 else 1
        # rows nobody writes stay poisoned unless the caller asked for a zeroed output, like the product's torch.empty / zeros
        y = torch.full((T * topk, NW * 8), 0.0 if zero_init else float("nan"), dtype=torch.float16)
        npad = int(num_tokens_post_padded.reshape(-1)[0])
        for blk in range(npad // block_rows):
            e = int(expert_ids[blk])
            assert 0 <= e < E
            for r in range(block_rows):
                pid = int(sorted_token_ids[blk * block_rows + r])
                if pid >= T * topk:
                    continue
                y32, _ = oracle.linear_gemm(x2[pid // x_div: pid // x_div + 1].numpy(), qweight[e].numpy(), qzeros[e].numpy(),
                                            scales[e].numpy(), g)
                row = torch.from_numpy(y32)[0]
                if mul_weights:
                    row = row * float(topk_weights.reshape(-1)[pid])
                y[pid] = row.half()
        return y.reshape(T, topk, NW * 8)

    def silu_and_mul(gate_up, out=None):
        r = torch.from_numpy(oracle.silu_and_mul(np.nan_to_num(gate_up.numpy())))
        if out is None:
            return r
        out.copy_(r)
        return out

    ops.grouped_gemm_forward, ops.silu_and_mul = grouped_gemm_forward, silu_and_mul


def make_case(E, H, I, T, seed):
    gen = torch.Generator().manual_seed(seed)
    lim = 0x7FFFFFFF

    def stack(K, N):
        return types.SimpleNamespace(
            qweight=torch.randint(-lim - 1, lim, (E, K, N // 8), dtype=torch.int32, generator=gen),
            qzeros=torch.randint(-lim - 1, lim, (E, K // 128, N // 8), dtype=torch.int32, generator=gen),
            scales=(torch.rand((E, K // 128, N), generator=gen) * 0.02 + 0.005).half(), group_size=128)

    ws, w2s = stack(H, 2 * I), stack(I, H)
    gate = torch.nn.Linear(H, E, bias=False)
    with torch.no_grad():
        gate.weight.copy_(torch.randn((E, H), generator=gen) * 0.5)
    x = torch.randn((1, T, H), generator=gen).half()
    return ws, w2s, gate.half(), x


def _worker(rank, world, port, q, E, T, topk):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from autoawq_amd import ep, ops
        from oracle import awq_oracle

        oracle_moe_ops(ops, awq_oracle)
        ws, w2s, gate, x = make_case(E, 256, 128, T, seed=11)  # the same data on every rank
        gate = gate.float()  # CPU matmul; the routing below is what every rank must agree on
        block = ep.ExpertParallelSparseMoeBlock(topk, lambda h: gate(h.float()), ws, w2s, rank, world)
        got = block(x)  # all-reduced inside
        logits = gate(x.view(-1, x.shape[-1]).float()).detach()
        want, ids, _ = awq_oracle.moe_forward(x.view(-1, x.shape[-1]).numpy(), logits.numpy(),
                                              dict(qweight=ws.qweight.numpy(), qzeros=ws.qzeros.numpy(), scales=ws.scales.numpy()),
                                              dict(qweight=w2s.qweight.numpy(), qzeros=w2s.qzeros.numpy(), scales=w2s.scales.numpy()), topk, 128)
        g, w = got.view(-1, got.shape[-1]).float().numpy(), want.astype(np.float32)
        err = float(np.abs(g - w).max() / np.abs(w).max())
        owned = int(((ids >= block.e0) & (ids < block.e1)).sum())
        q.put((rank, err, block.e0, block.e1, owned, bool(np.isfinite(g).all()), tuple(block.ws.qweight.shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("E,T,topk", [(8, 4, 2), (5, 9, 2), (4, 40, 3)])
def test_expert_parallel_moe_world2_gloo(E, T, topk):
    """Mixtral-like (8 experts, bs 4, top-2), an uneven expert split (5 over 2 ranks) and a prefill-sized batch that
    takes the 16-row blocks: both ranks return the reference result; each holds only its slice of the stack."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29650 + (os.getpid() + E * 7 + T) % 300
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, E, T, topk)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0, "a rank died (see its traceback above)"
    res = sorted(q.get(timeout=10) for _ in procs)
    assert [r[2:4] for r in res] == [(0, (E + 1) // 2), ((E + 1) // 2, E)]
    assert sum(r[4] for r in res) == T * topk                       # every pair is owned by exactly one rank
    for rank, err, e0, e1, owned, finite, shape in res:
        assert finite, "a row of a foreign pair leaked into the sum"
        assert shape[0] == e1 - e0
        assert err < 4e-3, (rank, err)                               # fp16 roundings of the partial sums + their order


def test_local_routing_cuts_the_foreign_bucket(oracle):
    """local_routing against the oracle's block alignment of the owned pairs alone."""
    from autoawq_amd import ep

    gen = torch.Generator().manual_seed(2)
    ids = torch.randint(0, 8, (13, 2), generator=gen)
    for e0, n in [(0, 4), (4, 4), (2, 3), (7, 1)]:
        for rows in (8, 16):
            s, e, npost, owned = ep.local_routing(ids, e0, n, rows)
            npost = int(npost)
            assert npost % rows == 0 and int(owned.sum()) == int(((ids >= e0) & (ids < e0 + n)).sum())
            flat = ids.reshape(-1)
            seen = []
            for b in range(npost // rows):
                for r in range(rows):
                    pid = int(s[b * rows + r])
                    if pid < flat.numel():
                        assert int(flat[pid]) - e0 == int(e[b]), "a pair sits in a block of another expert"
                        seen.append(pid)
            assert sorted(seen) == [i for i in range(flat.numel()) if e0 <= int(flat[i]) < e0 + n]
            assert int(e.max()) < n
