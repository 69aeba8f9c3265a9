#!/usr/bin/env python3
"""BASELINE config 4 on one GPU: per-rank kernel time of the Llama-3-70B-shape TP=8 shards
(qkv 8192->1280 column, o 1024->8192 row, gate+up 8192->7168 column,
down 3584->8192 row; 80 layers; bs = 1), captured in one hipGraph with distinct weights per layer, in the
GEMV layout (the decode layout: csrc/gemv_rows.hip) and in the GEMM layout.  The 160 all-reduces per token cannot
be measured on a 1-GPU box: the one-shot all-reduce kernel (csrc/allreduce.hip) is timed with its 8 ranks as ONE
group launch on this GPU (its own latency without xGMI), and the tok/s lines that include a collective are
MODELLED (per-rank compute + an assumed collective latency) and say so."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from autoawq_amd import ops
from autoawq_amd.comm import OneShotAllReduce
from bench import algorithmic_bytes, rand_packed, rand_packed_nk

dev = torch.device("cuda")
shapes = [("qkv", 8192, 1280), ("o", 1024, 8192), ("gate_up", 8192, 7168), ("down", 3584, 8192)]
LAYERS = 80
bytes_tok = LAYERS * sum(algorithmic_bytes(K, N, 1, 128) for _, K, N in shapes)
s = torch.cuda.Stream()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(fn, reps):
    with torch.cuda.stream(s):
        fn(); fn(); s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn()
        for _ in range(3): g.replay()
        s.synchronize()
        e0.record(s)
        for _ in range(reps): g.replay()
        e1.record(s); e1.synchronize()
    return e0.elapsed_time(e1) / reps


best = None
for layout in ("gemv", "gemm"):
    gen = torch.Generator(device=dev).manual_seed(0)
    pack = (lambda K, N: rand_packed_nk(K, N, 128, dev, gen)) if layout == "gemv" else (lambda K, N: rand_packed(K, N, 128, dev, gen))
    run = (lambda x, qw, qz, sc: ops.gemv_forward(x, qw, sc, qz, 128)) if layout == "gemv" else (lambda x, qw, qz, sc: ops.gemm_forward(x, qw, sc, qz))
    model = [[(pack(K, N), torch.randn((1, K), device=dev, generator=gen).half()) for _, K, N in shapes] for _ in range(LAYERS)]
    outs = []

    def step():
        outs.clear()
        for layer in model:
            for (qw, qz, sc), x in layer:
                outs.append(run(x, qw, qz, sc))

    ms = timed(step, 50)
    best = ms if best is None else min(best, ms)
    print(f"70B TP=8 per-rank shard set, bs=1, {layout} layout: {ms:.3f} ms/token compute ({LAYERS * 4} launches, "
          f"{ms * 1e3 / (LAYERS * 4):.2f} us/launch), {bytes_tok / 1e9:.2f} GB/rank/token = {bytes_tok / (ms * 1e-3) / 1e12:.2f} TB/s per rank "
          f"({bytes_tok / (ms * 1e-3) / 8e12:.3f} of 8 TB/s)")
    for i, (nm, K, N) in enumerate(shapes):
        sets = [model[l][i] for l in range(LAYERS)]

        def one():
            outs.clear()
            for (qw, qz, sc), x in sets:
                outs.append(run(x, qw, qz, sc))

        us = timed(one, 20) * 1e3 / LAYERS
        by = algorithmic_bytes(K, N, 1, 128)
        print(f"  {nm:8s} {K:5d} -> {N:5d}: {us:6.2f} us  {by / us / 1e3:7.0f} GB/s  kernel {ops.last_kernel()}")
    del model

# the one-shot all-reduce kernel itself: 8 ranks as one group launch on this GPU, [1, 8192] fp16 (16 KiB), 160 per token
ranks = OneShotAllReduce.local_group(8, 8192, dev)
xs = [torch.randn((1, 8192), device=dev).half() for _ in range(8)]
ys = [torch.empty_like(x) for x in xs]


def reduces():
    for _ in range(160):
        OneShotAllReduce.group_call(ranks, xs, ys)


ar_us = timed(reduces, 20) * 1e3 / 160
ref = sum(x.float() for x in xs).half()
assert torch.equal(ys[3], ref) or float((ys[3].float() - ref.float()).abs().max()) < 2e-2
print(f"one-shot all-reduce kernel, 8 ranks in one group launch on ONE GPU, [1, 8192] fp16: {ar_us:.2f} us per all-reduce "
      f"(staging writes, flags and the fp32 sum in HBM of this GPU; over xGMI the peer writes cross one link instead)")
for us in (ar_us, 8.0, 15.0, 25.0):
    tot = best + 160 * us * 1e-3
    print(f"  MODELLED with {us:.1f} us per all-reduce (160 per token): {1000.0 / tot:7.1f} tok/s")
