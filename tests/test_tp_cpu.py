"""World-size-2 gloo test of the tensor-parallel sharding: column shards concatenate
exactly, row shards all-reduce to the unsharded product, bias is added once.  The arithmetic of
each shard is done by the CPU oracle here (no GPU); on the GPU the same shards feed the HIP
kernels (autoawq_amd/tp.py)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from autoawq_amd import tp
        from oracle import awq_oracle

        gen = torch.Generator().manual_seed(3)  # same data on every rank
        K, N, g, M = 768, 96, 128, 3            # 6 groups -> uneven-free; 12 column units of 8
        lim = 0x7FFFFFFF
        qw = torch.randint(-lim - 1, lim, (K, N // 8), dtype=torch.int32, generator=gen)
        qz = torch.randint(-lim - 1, lim, (K // g, N // 8), dtype=torch.int32, generator=gen)
        sc = (torch.rand((K // g, N), generator=gen) * 0.02 + 0.005).half()
        bias = torch.randn((N,), generator=gen).half()
        x = torch.randn((M, K), generator=gen).half()
        full32, _ = awq_oracle.linear_gemm(x.numpy(), qw.numpy(), qz.numpy(), sc.numpy(), g, bias.numpy())

        # column parallel: local slice, all_gather, compare exactly
        s, c = tp.split_even_units(N // 8, world)[rank]
        n0, n1 = s * 8, (s + c) * 8
        cq, cz, cs, cb = tp.column_shard(qw, qz, sc, bias, n0, n1)
        loc32, _ = awq_oracle.linear_gemm(x.numpy(), cq.numpy(), cz.numpy(), cs.numpy(), g, cb.numpy())
        parts = [torch.empty((M, N // world), dtype=torch.float32) for _ in range(world)]
        dist.all_gather(parts, torch.from_numpy(loc32))
        col_ok = np.array_equal(torch.cat(parts, 1).numpy(), full32)

        # row parallel: whole groups of rows, partial sums, ONE all-reduce, bias on rank 0 only
        gs, gc = tp.split_even_units(K // g, world)[rank]
        k0, k1 = gs * g, (gs + gc) * g
        rq, rz, rs = tp.row_shard(qw, qz, sc, k0, k1, g)
        part32, _ = awq_oracle.linear_gemm(x[:, k0:k1].numpy(), rq.numpy(), rz.numpy(), rs.numpy(), g,
                                           bias.numpy() if rank == 0 else None)
        t = torch.from_numpy(part32.astype(np.float64))
        dist.all_reduce(t)
        row_err = float(np.abs(t.numpy() - full32).max() / np.abs(full32).max())
        q.put((rank, col_ok, row_err))
    finally:
        dist.destroy_process_group()


def test_tp_sharding_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, col_ok, row_err in res:
        assert col_ok, f"rank {rank}: column-parallel concat differs"
        assert row_err < 1e-6, f"rank {rank}: row-parallel all-reduce off by {row_err}"


def test_uneven_group_split_llama7b_down():
    """86 groups over 8 ranks: 6 x 11 + 2 x 10, contiguous, whole groups."""
    from autoawq_amd.tp import split_even_units

    parts = split_even_units(86, 8)
    assert [c for _, c in parts] == [11] * 6 + [10] * 2
    assert parts[0][0] == 0 and all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(7))
    assert parts[-1][0] + parts[-1][1] == 86
