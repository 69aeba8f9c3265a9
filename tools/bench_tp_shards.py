#!/usr/bin/env python3
"""BASELINE config 4 on one GPU: per-rank kernel time of the Llama-3-70B-shape TP=8 shards
(qkv 8192->1280 column, o 1024->8192 row, gate+up 8192->7168 column,
down 3584->8192 row; 80 layers; bs = 1), captured in one hipGraph with distinct weights per layer.
The 160 all-reduces per token cannot be measured on a 1-GPU box: the tok/s lines that include
them are MODELLED (per-rank compute + assumed collective latency) and say so."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from autoawq_amd import ops
from bench import algorithmic_bytes, rand_packed

dev = torch.device("cuda")
gen = torch.Generator(device=dev).manual_seed(0)
shapes = [("qkv", 8192, 1280), ("o", 1024, 8192), ("gate_up", 8192, 7168), ("down", 3584, 8192)]
LAYERS = 80
model = []
for _ in range(LAYERS):
    model.append([(rand_packed(K, N, 128, dev, gen), torch.randn((1, K), device=dev, generator=gen).half()) for _, K, N in shapes])
bytes_tok = LAYERS * sum(algorithmic_bytes(K, N, 1, 128) for _, K, N in shapes)
outs = []
def step():
    outs.clear()
    for layer in model:
        for (qw, qz, sc), x in layer:
            outs.append(ops.gemm_forward(x, qw, sc, qz))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    step(); step(); s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        step()
    for _ in range(5): g.replay()
    s.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    reps = 50
    for _ in range(reps): g.replay()
    e1.record(s); e1.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"70B TP=8 per-rank shard set, bs=1: {ms:.3f} ms/token compute ({LAYERS * 4} launches, {ms * 1e3 / (LAYERS * 4):.2f} us/launch), "
      f"{bytes_tok / 1e9:.2f} GB/rank/token -> {bytes_tok / ms / 1e9:.1f} TB/s... = {bytes_tok / (ms * 1e-3) / 1e12:.2f} TB/s per rank")
# per-shape
for i, (nm, K, N) in enumerate(shapes):
    sets = [model[l][i] for l in range(LAYERS)]
    with torch.cuda.stream(s):
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, stream=s):
            for (qw, qz, sc), x in sets:
                outs.append(ops.gemm_forward(x, qw, sc, qz))
        g2.replay(); s.synchronize()
        e0.record(s)
        for _ in range(20): g2.replay()
        e1.record(s); e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (20 * LAYERS)
    by = algorithmic_bytes(K, N, 1, 128)
    print(f"  {nm:8s} {K:5d} -> {N:5d}: {us:6.2f} us  {by / us / 1e3:7.0f} GB/s  kernel {ops.last_kernel()}")
for ar_us in (8.0, 15.0, 25.0):
    tot = ms + 160 * ar_us * 1e-3
    print(f"  MODELLED with {ar_us:.0f} us per [1, 8192] fp16 all-reduce (160 per token): {1000.0 / tot:7.1f} tok/s")
