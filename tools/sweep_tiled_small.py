#!/usr/bin/env python3
"""Fused tiled kernel in the batched-decode regime (16 < M <= 128): auto vs forced split-K."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from autoawq_amd import ops
from bench import algorithmic_bytes, rand_packed
from bench_small_m import time_graph  # noqa: E402  (module body runs its own table only under __main__)

dev = torch.device("cuda")
gen = torch.Generator(device=dev).manual_seed(0)
BNS = [int(v) for v in os.environ.get("BNS", "1,2").split(",")]
Ms = [int(v) for v in os.environ.get("MS", "17,32,64,128").split(",")]
ops.workspace(dev, 16384 + (64 << 20))
for K, N in [(4096, 11008), (4096, 4096), (11008, 4096), (4096, 22016)]:
    per = K * N // 2
    nsets = max(4, min(48, (640 << 20) // per))
    sets = [rand_packed(K, N, 128, dev, gen) for _ in range(nsets)]
    for M in Ms:
        x = torch.randn((M, K), device=dev, generator=gen).half()
        by = algorithmic_bytes(K, N, M, 128)
        row = f"K{K} N{N} M{M:3d}:"
        for bn, sk in [(b, k) for b in BNS for k in (0, 4, 8, 16)]:
            fl = ops.gemm_flags(ops.KERNEL_TILED, nlog=bn, splitk=sk)
            try:
                us = time_graph(lambda i: ops.gemm_forward(x, sets[i][0], sets[i][2], sets[i][1], flags=fl), nsets)
            except Exception as e:
                row += f"  b{bn}s{sk} fail"
                continue
            row += f"  b{bn}s{sk} {us:5.1f}"
        us = time_graph(lambda i: torch.matmul(x, ops.dequantize_weights(sets[i][0], sets[i][2], sets[i][1])), nsets)
        row += f"  | two-pass {us:6.1f}  | ideal@6.5TB/s {by / 6.5e6:5.1f} us"
        print(row, flush=True)
    del sets
    torch.cuda.empty_cache()
