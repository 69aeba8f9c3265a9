#!/usr/bin/env python3
"""BASELINE configs[4]: Mixtral-8x7B-shape AWQ int4 g128 fused MoE MLP, bs=4 decode on 1 MI355X.
E=8, top-2, hidden 4096, inter 14336: w1|w3 stacked [8, 4096, 3584] i32, w2 [8, 14336, 512] i32.
Reports us per MoE block (router excluded), bytes of the experts actually hit, GB/s -- after checking the block's
output against the same experts run one (token, expert) pair at a time through the plain decode kernel."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

E, H, I, g, T, topk = 8, 4096, 14336, 128, 4, 2


class Stack:
    pass


def run(dev=None, verbose=True, layers=2, reps=20, T=T, twins=True):
    """twins: the decode path of fuse_mixtral(decode_layout="auto") -- GEMV-layout copies of the expert stacks, every pair one
    batch-1 call of the row-streaming kernel (modules/fused/moe.py::_apply_moe_rows); the GEMM-layout grouped kernel
    (rounds 1-5) is timed beside it and the two outputs are compared."""
    from autoawq_amd import ops
    from autoawq_amd.modules.fused import moe as moe_mod
    from autoawq_amd.modules.fused.moe import apply_moe_weights
    from bench import algorithmic_bytes

    dev = dev or torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(0)
    lim = 0x7FFFFFFF

    def experts(K, N):
        s = Stack()
        s.qweight = torch.randint(-lim - 1, lim, (E, K, N // 8), dtype=torch.int32, device=dev, generator=gen)
        s.qzeros = torch.randint(-lim - 1, lim, (E, K // g, N // 8), dtype=torch.int32, device=dev, generator=gen)
        s.scales = (torch.rand((E, K // g, N), device=dev, generator=gen) * 0.02 + 0.005).half()
        return s

    stacks = [(experts(H, 2 * I), experts(I, H)) for _ in range(layers)]  # 2 layers x 733 MB: defeats the 256 MiB L3
    x = torch.randn((T, H), device=dev, generator=gen).half()
    logits = torch.randn((T, E), device=dev, generator=gen)
    w, ids = ops.fused_topk(logits, topk, True)
    hit = len(set(ids.reshape(-1).tolist()))
    by = hit * (algorithmic_bytes(H, 2 * I, 1, g) + algorithmic_bytes(I, H, 1, g))

    # correctness before timing: the block vs one (token, expert) pair at a time through awq_gemm_forward
    w1, w2 = stacks[0]
    got = apply_moe_weights(w1, w2, x, logits, topk, True).float()
    want = torch.zeros((T, H), dtype=torch.float32, device=dev)
    for t in range(T):
        for j in range(topk):
            e = int(ids[t, j])
            gu = ops.gemm_forward(x[t:t + 1], w1.qweight[e], w1.scales[e], w1.qzeros[e])
            d = ops.gemm_forward(ops.silu_and_mul(gu), w2.qweight[e], w2.scales[e], w2.qzeros[e])
            want[t] += (float(w[t, j]) * d[0].float()).half().float()
    rel = float((got - want).abs().max() / want.abs().max())
    assert rel < 5e-3, f"MoE block differs from the per-pair computation by {rel}"

    def step():
        for a, b in stacks:
            apply_moe_weights(a, b, x, logits, topk, True)

    def timed():
        s = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s):
            step()
            step()
            s.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                step()
            gr.replay()
            s.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(reps):
                gr.replay()
            e1.record(s)
            e1.synchronize()
        del gr
        return e0.elapsed_time(e1) * 1e3 / (reps * len(stacks))

    us_gemm = timed()
    us, rel_t, kern = us_gemm, None, ops.last_kernel()
    if twins:
        for a, b in stacks:
            moe_mod.build_decode_twins(a, b)
        got_t = apply_moe_weights(w1, w2, x, logits, topk, True).float()
        kern = ops.last_kernel()
        rel_t = float((got_t - want).abs().max() / want.abs().max())
        assert rel_t < 5e-3, f"MoE block on the decode twins differs from the per-pair computation by {rel_t}"
        us = timed()
    del stacks
    torch.cuda.empty_cache()
    if verbose:
        print(f"Mixtral-8x7B-shape MoE MLP, bs={T}, top-{topk}, {hit} experts hit: {us:.1f} us per block [{kern}] "
              f"({by / 1e6:.0f} MB of expert weights streamed -> {by / us / 1e3:.0f} GB/s, {by / us / 80e3:.1f}% of 8 TB/s); "
              f"GEMM-layout grouped kernel {us_gemm:.1f} us ({by / us_gemm / 80e3:.1f}%); hipGraph-captured, no host reads; "
              f"output within {rel:.1e} (GEMM layout) / {rel_t if rel_t is None else format(rel_t, '.1e')} (twins) of the per-pair computation")
    return {"us_per_block": us, "kernel": kern, "gemm_layout_grouped_us": us_gemm, "bytes": by, "experts_hit": hit, "tokens": T,
            "checked_against": f"per-(token, expert) awq_gemm_forward calls, max rel diff {rel:.1e} (GEMM-layout grouped kernel)"
                               + (f", {rel_t:.1e} (row-streaming kernel on the GEMV-layout twins)" if rel_t is not None else "")}


def run_prefill(T=512, dev=None, verbose=True, reps=5):
    """The same block at a prefill-sized token count (T x top-2 = 1024 pairs, ~128 rows per expert): the grouped prefill path of
    apply_moe_weights (round 4: ONE launch of the register-decoded MFMA GEMM per projection, tiles dealt over the experts from
    device-side offsets) vs round 3's one-GEMM-per-expert loop (host read-back) and the 16-row-block grouped decode kernel;
    MFMA TFLOP/s (2 * pairs * (K N) flops of both projections)."""
    from autoawq_amd.modules.fused import moe

    dev = dev or torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(1)
    lim = 0x7FFFFFFF

    def experts(K, N):
        s = Stack()
        s.qweight = torch.randint(-lim - 1, lim, (E, K, N // 8), dtype=torch.int32, device=dev, generator=gen)
        s.qzeros = torch.randint(-lim - 1, lim, (E, K // g, N // 8), dtype=torch.int32, device=dev, generator=gen)
        s.scales = (torch.rand((E, K // g, N), device=dev, generator=gen) * 0.004 + 0.001).half()
        return s

    w1, w2 = experts(H, 2 * I), experts(I, H)
    x = torch.randn((T, H), device=dev, generator=gen).half()
    logits = torch.randn((T, E), device=dev, generator=gen)
    fl = 2.0 * T * topk * (H * 2 * I + I * H)

    def timeit(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            y = fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps, y

    us_p, yp = timeit(lambda: moe.apply_moe_weights(w1, w2, x, logits, topk, True))
    saved = moe.PREFILL_MIN_PAIRS
    try:
        moe.PREFILL_MIN_PAIRS = 1 << 30
        us_b, yb = timeit(lambda: moe.apply_moe_weights(w1, w2, x, logits, topk, True))
    finally:
        moe.PREFILL_MIN_PAIRS = saved
    rel = float((yp.float() - yb.float()).abs().max() / yb.float().abs().max())

    def per_expert():  # round 3's path, for the A/B
        tw, ti = moe.fused_topk(logits, topk, True)
        flat = ti.reshape(-1).long()
        order = torch.argsort(flat, stable=True)
        counts = torch.bincount(flat, minlength=E).int()
        ys = moe._per_expert_gemms(w1, w2, x.index_select(0, order // topk), counts)
        out = torch.empty((T * topk, H), dtype=torch.float16, device=dev)
        out.index_copy_(0, order, (ys.float() * tw.reshape(-1).index_select(0, order).float()[:, None]).half())
        return out.view(T, topk, H).sum(dim=1)

    us_e, _ = timeit(per_expert)
    if verbose:
        print(f"Mixtral-8x7B-shape MoE MLP, T={T}, top-{topk}: grouped prefill GEMMs {us_p:.0f} us ({fl / us_p / 1e6:.0f} TF), "
              f"one GEMM per expert (r03) {us_e:.0f} us, 16-row-block grouped kernel {us_b:.0f} us ({fl / us_b / 1e6:.0f} TF); max rel diff {rel:.1e}")
    return {"tokens": T, "grouped_prefill_us": us_p, "per_expert_fused_gemm_us_r03": us_e, "block16_grouped_us": us_b, "flops": fl,
            "roofline": {"bound": "mfma", "achieved": fl / us_p / 1e6, "peak": 2500.0, "unit": "TFLOP/s", "frac": fl / us_p / 1e6 / 2500.0},
            "paths_max_rel_diff": rel, "note": "eager timing; the grouped path reads nothing back (hipGraph-capturable), round 3's per-expert path synchronises once per block"}


def run_ep(world, dev=None, layers=2, reps=20, verbose=True):
    """Expert-parallel split of the same block (autoawq_amd/ep.py) with the ranks run one after the other on THIS GPU:
    per-rank time of the local part (routing + the two grouped GEMMs over the owned experts that were hit, without the
    final all-reduce of 32 KiB), and the sum over ranks checked against the unsharded block."""
    from autoawq_amd import ep, ops
    from autoawq_amd.modules.fused.moe import apply_moe_weights
    from bench import algorithmic_bytes

    dev = dev or torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(0)
    lim = 0x7FFFFFFF

    def experts(K, N):
        s = Stack()
        s.group_size = g
        s.qweight = torch.randint(-lim - 1, lim, (E, K, N // 8), dtype=torch.int32, device=dev, generator=gen)
        s.qzeros = torch.randint(-lim - 1, lim, (E, K // g, N // 8), dtype=torch.int32, device=dev, generator=gen)
        s.scales = (torch.rand((E, K // g, N), device=dev, generator=gen) * 0.02 + 0.005).half()
        return s

    stacks = [(experts(H, 2 * I), experts(I, H)) for _ in range(layers)]
    x = torch.randn((T, H), device=dev, generator=gen).half()
    logits = torch.randn((T, E), device=dev, generator=gen)
    _, ids = ops.fused_topk(logits, topk, True)
    full = apply_moe_weights(stacks[0][0], stacks[0][1], x, logits, topk, True).float()
    total = torch.zeros_like(full)
    per_rank = []
    for r in range(world):
        e0, e1 = ep.expert_bounds(E, r, world)
        shards = [(ep.ExpertShard(a, e0, e1), ep.ExpertShard(b, e0, e1)) for a, b in stacks]
        total += ep.apply_moe_weights_local(shards[0][0], shards[0][1], x, logits, topk, True, e0).float()
        hit = len({int(v) for v in ids.reshape(-1).tolist() if e0 <= int(v) < e1})

        def step():
            for a, b in shards:
                ep.apply_moe_weights_local(a, b, x, logits, topk, True, e0)

        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            step()
            st.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st):
                step()
            gr.replay()
            st.synchronize()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record(st)
            for _ in range(reps):
                gr.replay()
            t1.record(st)
            t1.synchronize()
        us = t0.elapsed_time(t1) * 1e3 / (reps * len(shards))
        by = hit * (algorithmic_bytes(H, 2 * I, 1, g) + algorithmic_bytes(I, H, 1, g))
        per_rank.append({"rank": r, "experts": [e0, e1], "experts_hit": hit, "us_per_block": us, "MB": by / 1e6})
        if verbose:
            print(f"  EP={world} rank {r}: experts [{e0}, {e1}), {hit} hit, {us:6.1f} us per block, {by / 1e6:5.0f} MB"
                  + (f" -> {by / us / 1e3:.0f} GB/s" if hit else ""), flush=True)
        del gr, shards
        torch.cuda.empty_cache()
    rel = float((total - full).abs().max() / full.abs().max())
    assert rel < 5e-3, f"sum over the expert-parallel ranks differs from the unsharded block by {rel}"
    worst = max(p["us_per_block"] for p in per_rank)
    if verbose:
        print(f"  EP={world}: busiest rank {worst:.1f} us per block (+ one 32 KiB all-reduce); sum over ranks within {rel:.1e} of the unsharded block")
    return {"world": world, "per_rank": per_rank, "busiest_rank_us": worst, "sum_vs_unsharded_max_rel": rel}


if __name__ == "__main__" and "--sweep" in sys.argv:
    # where the row-streaming path hands over to the GEMM-layout grouped kernel (modules/fused/moe.py::ROWS_MAX_PAIRS)
    from autoawq_amd.modules.fused import moe as _m

    _m.ROWS_MAX_PAIRS = 1 << 30
    for t_ in (1, 2, 4, 8, 12, 16, 24, 32):
        run(T=t_, layers=2 if t_ <= 8 else 1, reps=10)
    sys.exit(0)
if __name__ == "__main__" and "--prefill" in sys.argv:
    run_prefill()
    sys.exit(0)
if __name__ == "__main__":
    import argparse

    ap = argparse.ArgumentParser()
    ap.add_argument("--ep", type=int, nargs="*", default=[], help="also time the expert-parallel split on W ranks (run one after the other here)")
    a = ap.parse_args()
    run()
    for w_ in a.ep:
        run_ep(w_)
