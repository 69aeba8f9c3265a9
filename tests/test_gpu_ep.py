"""Expert-parallel MoE on the HIP kernels: the ranks' local parts (autoawq_amd/ep.py), run one after the other on one
GPU, sum to the unsharded fused block; a rank's step is hipGraph-capturable (its routing never leaves the device)."""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu


def _stack(E, K, N, gen, dev):
    lim = 0x7FFFFFFF
    return types.SimpleNamespace(
        qweight=torch.randint(-lim - 1, lim, (E, K, N // 8), dtype=torch.int32, device=dev, generator=gen),
        qzeros=torch.randint(-lim - 1, lim, (E, K // 128, N // 8), dtype=torch.int32, device=dev, generator=gen),
        scales=(torch.rand((E, K // 128, N), device=dev, generator=gen) * 0.02 + 0.005).half(), group_size=128)


@pytest.mark.parametrize("E,T,topk,world", [(8, 4, 2, 2), (8, 4, 2, 8), (8, 1, 2, 4), (6, 33, 2, 4), (8, 200, 3, 3), (8, 1100, 2, 2)])  # the last: past the one-launch router, torch top-k + local_routing
def test_expert_parallel_ranks_sum_to_the_unsharded_block(E, T, topk, world):
    from autoawq_amd import _lib, ep
    from autoawq_amd.modules.fused.moe import apply_moe_weights

    _lib.lib()
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(E * 100 + T)
    H, I = 1024, 2816
    w1, w2 = _stack(E, H, 2 * I, gen, dev), _stack(E, I, H, gen, dev)
    x = torch.randn((T, H), device=dev, generator=gen).half()
    logits = torch.randn((T, E), device=dev, generator=gen)
    from autoawq_amd.modules.fused import moe as _moe

    saved = _moe.PREFILL_MIN_PAIRS  # like for like: the ranks run the grouped block kernel, so does the unsharded reference
    _moe.PREFILL_MIN_PAIRS = 1 << 30
    try:
        full = apply_moe_weights(w1, w2, x, logits, topk, True).float()
    finally:
        _moe.PREFILL_MIN_PAIRS = saved
    total = torch.zeros_like(full)
    covered = 0
    for r in range(world):
        e0, e1 = ep.expert_bounds(E, r, world)
        a, b = ep.ExpertShard(w1, e0, e1), ep.ExpertShard(w2, e0, e1)
        part = ep.apply_moe_weights_local(a, b, x, logits, topk, True, e0)
        assert bool(torch.isfinite(part).all()), "a row of a foreign pair leaked into the sum"
        total += part.float()
        covered += e1 - e0
    assert covered == E
    err = (total - full).abs()
    assert bool((err <= 4e-3 * full.abs() + 4e-3 * full.abs().mean()).all()), float(err.max())


def test_expert_parallel_rank_step_is_graph_capturable():
    from autoawq_amd import ep

    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(5)
    E, H, I, T, topk = 8, 1024, 2816, 4, 2
    w1, w2 = _stack(E, H, 2 * I, gen, dev), _stack(E, I, H, gen, dev)
    a, b = ep.ExpertShard(w1, 2, 5), ep.ExpertShard(w2, 2, 5)
    x = torch.randn((T, H), device=dev, generator=gen).half()
    logits = torch.randn((T, E), device=dev, generator=gen)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ref = ep.apply_moe_weights_local(a, b, x, logits, topk, True, 2)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            y = ep.apply_moe_weights_local(a, b, x, logits, topk, True, 2)
        logits2 = torch.randn((T, E), device=dev, generator=gen)   # new routing through the SAME captured graph
        want2 = None
        g.replay()
        s.synchronize()
        assert torch.equal(y, ref)
        logits.copy_(logits2)
        g.replay()
        s.synchronize()
        got2 = y.clone()
        want2 = ep.apply_moe_weights_local(a, b, x, logits, topk, True, 2)
        s.synchronize()
    assert torch.equal(got2, want2)


@pytest.mark.parametrize("E,T,topk,world", [(8, 4, 2, 2), (8, 4, 2, 8), (8, 1, 2, 4), (6, 6, 2, 3)])
def test_expert_parallel_ranks_on_decode_twins_sum_to_the_unsharded_block(E, T, topk, world):
    """round 6: a shard with GEMV-layout twins of ITS experts runs the decode step through the grouped row-streaming launch
    (the router's global ids, first_expert = the shard's first: pairs of foreign experts are skipped and stay zero); the ranks'
    partial sums add up to the unsharded block on its twins and on its GEMM-layout stacks."""
    from autoawq_amd import _lib, ep, ops
    from autoawq_amd.modules.fused import moe as _moe

    _lib.lib()
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(E * 10 + T)
    H, I = 1024, 2816
    w1, w2 = _stack(E, H, 2 * I, gen, dev), _stack(E, I, H, gen, dev)
    x = torch.randn((T, H), device=dev, generator=gen).half()
    logits = torch.randn((T, E), device=dev, generator=gen)
    base = _moe.apply_moe_weights(w1, w2, x, logits, topk, True).float()
    assert ops.last_kernel() == "gemv_mfma_grouped"
    total = torch.zeros_like(base)
    for r in range(world):
        e0, e1 = ep.expert_bounds(E, r, world)
        a, b = ep.ExpertShard(w1, e0, e1), ep.ExpertShard(w2, e0, e1)
        _moe.build_decode_twins(a, b)
        part = ep.apply_moe_weights_local(a, b, x, logits, topk, True, e0)
        assert ops.last_kernel() == "gemv_rows_grouped"
        assert bool(torch.isfinite(part).all())
        total += part.float()
    _moe.build_decode_twins(w1, w2)
    full = _moe.apply_moe_weights(w1, w2, x, logits, topk, True).float()
    assert ops.last_kernel() == "gemv_rows_grouped"
    # the same pairs through the same kernel, summed per rank first: fp16 rounding of the partial sums only
    for ref in (full, base):
        err = (total - ref).abs()
        assert bool((err <= 4e-3 * ref.abs() + 4e-3 * ref.abs().mean()).all()), float(err.max())
