"""The one-shot small-message all-reduce (csrc/allreduce.hip, autoawq_amd/comm.py) on ONE GPU: P "ranks" in one process
(their staging / flag buffers are local allocations, exactly what the C ABI takes as peer pointers), one stream per rank."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def run_ranks(ranks, xs, streams, outs=None):
    """all ranks of the single-process group in ONE launch (co-residency guaranteed; P streams of one process may share a
    hardware queue and would then run the ranks one after the other)"""
    from autoawq_amd.comm import OneShotAllReduce

    OneShotAllReduce.group_call(ranks, xs, outs)


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("n", [4, 4096, 8192, 32768, 20004])
def test_oneshot_allreduce_sums_in_rank_order_bitwise_identical(world, n):
    from autoawq_amd.comm import OneShotAllReduce

    ranks = OneShotAllReduce.local_group(world, max_halfs=32768)
    streams = [torch.cuda.Stream() for _ in range(world)]
    gen = torch.Generator(device="cuda").manual_seed(n + world)
    for it in range(6):  # epochs alternate the staging halves; sizes may change between calls
        m = n if it % 2 == 0 else max(4, (n // 8) * 4)
        xs = [torch.randn((m,), device="cuda", generator=gen).half() for _ in range(world)]
        want = torch.zeros((m,), device="cuda", dtype=torch.float32)
        for x in xs:  # the kernel's order: rank 0 first, fp32 accumulation, one rounding
            want += x.float()
        want = want.half()
        torch.cuda.synchronize()
        run_ranks(ranks, xs, streams)
        torch.cuda.synchronize()
        for r in range(world):
            assert torch.equal(xs[r], want), f"rank {r} epoch {it}: max diff {(xs[r].float() - want.float()).abs().max()}"
    for ar in ranks:
        assert ar.status() == (6, 0)


def test_oneshot_allreduce_out_of_place_and_argument_checks():
    from autoawq_amd import _lib
    from autoawq_amd.comm import OneShotAllReduce

    ranks = OneShotAllReduce.local_group(2, max_halfs=1024)
    streams = [torch.cuda.Stream() for _ in range(2)]
    xs = [torch.full((1024,), float(r + 1), device="cuda", dtype=torch.float16) for r in range(2)]
    outs = [torch.empty_like(x) for x in xs]
    torch.cuda.synchronize()
    run_ranks(ranks, xs, streams, outs)
    torch.cuda.synchronize()
    assert all(bool((o == 3).all()) for o in outs) and bool((xs[0] == 1).all())
    with pytest.raises(_lib.AwqHipError):
        ranks[0](torch.zeros((1028,), device="cuda", dtype=torch.float16))  # larger than max_halfs
    with pytest.raises(_lib.AwqHipError):
        ranks[0](torch.zeros((6,), device="cuda", dtype=torch.float16))     # not a multiple of 4
    with pytest.raises(_lib.AwqHipError):
        ranks[0](torch.zeros((8,), device="cuda", dtype=torch.float32))


def test_oneshot_allreduce_replays_inside_one_hipgraph():
    """What bench.py --gpus N needs: the collective is a plain kernel launch, so a whole TP decode step captures.
    50 replays of a captured launch, fresh inputs each time (the epoch lives on the device)."""
    from autoawq_amd.comm import OneShotAllReduce

    world, n = 4, 4096
    ranks = OneShotAllReduce.local_group(world, max_halfs=n)
    xs = [torch.zeros((n,), device="cuda", dtype=torch.float16) for _ in range(world)]
    outs = [torch.empty_like(x) for x in xs]
    main = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(main):
        with torch.cuda.graph(g, stream=main):
            OneShotAllReduce.group_call(ranks, xs, outs)
    gen = torch.Generator(device="cuda").manual_seed(3)
    for it in range(50):
        vals = [torch.randn((n,), device="cuda", generator=gen).half() for _ in range(world)]
        for x, v in zip(xs, vals):
            x.copy_(v)
        want = torch.zeros((n,), device="cuda", dtype=torch.float32)
        for v in vals:
            want += v.float()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        for r in range(world):
            assert torch.equal(outs[r], want.half()), (it, r)
    assert all(ar.status() == (50, 0) for ar in ranks)


def test_oneshot_allreduce_missing_peer_raises_the_error_word_instead_of_hanging():
    from autoawq_amd.comm import OneShotAllReduce

    ranks = OneShotAllReduce.local_group(2, max_halfs=64)
    x = torch.ones((64,), device="cuda", dtype=torch.float16)
    ranks[0](x)  # rank 1 never calls
    torch.cuda.synchronize()
    done, err = ranks[0].status()
    assert done == 1 and err != 0
    assert bool(torch.isnan(x).all()), "a rank that gave up on a peer must not return a plausible-looking sum (ADVICE r03)"
    with pytest.raises(Exception, match="gave up waiting for peer 1"):
        ranks[0].check()


def test_oneshot_allreduce_buffers_are_library_allocated_uncached_memory():
    """Flags and staging come from awq_allreduce_alloc (uncached fine-grained device memory, zeroed), not from torch's pool."""
    from autoawq_amd.comm import OneShotAllReduce, _DeviceBytes

    ranks = OneShotAllReduce.local_group(2, max_halfs=256)
    for t in ranks[0].staging + ranks[0].flags:
        assert t.dtype == torch.uint8 and t.is_cuda and int(t.sum()) == 0
    b = _DeviceBytes(4096)
    t = b.tensor(torch.device("cuda", torch.cuda.current_device()))
    assert t.data_ptr() == b.ptr and t.numel() == 4096 and len(b.ipc_handle()) == 64
    xs = [torch.full((256,), float(r + 1), device="cuda", dtype=torch.float16) for r in range(2)]
    OneShotAllReduce.group_call(ranks, xs)
    torch.cuda.synchronize()
    assert all(bool((x == 3).all()) for x in xs) and ranks[0].check() == 1
