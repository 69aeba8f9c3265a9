#!/usr/bin/env python3
"""BASELINE configs[4]: Mixtral-8x7B-shape AWQ int4 g128 fused MoE MLP, bs=4 decode on 1 MI355X.
E=8, top-2, hidden 4096, inter 14336: w1|w3 stacked [8, 4096, 3584] i32, w2 [8, 14336, 512] i32.
Reports us per MoE block (router excluded), bytes of the experts actually hit, GB/s."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from autoawq_amd.modules.fused.moe import apply_moe_weights
from bench import algorithmic_bytes

dev = torch.device("cuda")
gen = torch.Generator(device=dev).manual_seed(0)
E, H, I, g, T, topk = 8, 4096, 14336, 128, 4, 2
lim = 0x7FFFFFFF


class Stack:
    pass


def experts(K, N):
    s = Stack()
    s.qweight = torch.randint(-lim - 1, lim, (E, K, N // 8), dtype=torch.int32, device=dev, generator=gen)
    s.qzeros = torch.randint(-lim - 1, lim, (E, K // g, N // 8), dtype=torch.int32, device=dev, generator=gen)
    s.scales = (torch.rand((E, K // g, N), device=dev, generator=gen) * 0.02 + 0.005).half()
    return s


layers = [(experts(H, 2 * I), experts(I, H)) for _ in range(2)]  # 2 layers x 733 MB: defeats the 256 MiB L3
x = torch.randn((T, H), device=dev, generator=gen).half()
logits = torch.randn((T, E), device=dev, generator=gen)
hit = len(set(torch.topk(torch.softmax(logits.float(), -1), topk, -1)[1].reshape(-1).tolist()))
by = hit * (algorithmic_bytes(H, 2 * I, 1, g) + algorithmic_bytes(I, H, 1, g))


def step():
    for w1, w2 in layers:
        apply_moe_weights(w1, w2, x, logits, topk, True)


s = torch.cuda.Stream()
with torch.cuda.stream(s):
    step(); step()
    s.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        step()
    gr.replay()
    s.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    reps = 20
    for _ in range(reps):
        gr.replay()
    e1.record(s)
    e1.synchronize()
us = e0.elapsed_time(e1) * 1e3 / (reps * len(layers))
print(f"Mixtral-8x7B-shape MoE MLP, bs={T}, top-{topk}, {hit} experts hit: {us:.1f} us per block "
      f"({by / 1e6:.0f} MB of expert weights streamed -> {by / us / 1e3:.0f} GB/s, {by / us / 80e3:.1f}% of 8 TB/s); hipGraph-captured, no host reads")
