#!/usr/bin/env python3
"""Sweep of the MFMA GEMV kernel for 8 < M <= 16 (unit / waves / split-K) -- run through gpurun."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from autoawq_amd import ops
from bench import algorithmic_bytes, rand_packed
from sweep_gemv import time_variant

dev = torch.device("cuda")
gen = torch.Generator(device=dev).manual_seed(0)
Ms = [int(v) for v in os.environ.get("MS", "8,12,16").split(",")]
for K, N in [(4096, 11008), (4096, 4096), (11008, 4096), (4096, 22016)]:
    per = K * N // 2
    nsets = max(4, min(64, (640 << 20) // per))
    sets = [rand_packed(K, N, 128, dev, gen) for _ in range(nsets)]
    for M in Ms:
        x = torch.randn((M, K), device=dev, generator=gen).half()
        by = algorithmic_bytes(K, N, M, 128)
        rows = []
        us, _ = time_variant(sets, x, 0)
        rows.append((us, "auto"))
        for wv in (2, 4):
            for un in (2, 4, 8):
                for sk in (4, 8, 16):
                    try:
                        us, _ = time_variant(sets, x, ops.gemm_flags(ops.KERNEL_MFMA_GEMV, nlog=2, splitk=sk, waves=wv, unit=un))
                    except Exception as e:
                        continue
                    rows.append((us, f"v{wv} u{un} s{sk}"))
        rows.sort()
        auto = [r for r in rows if r[1] == "auto"][0][0]
        print(f"K{K} N{N} M{M}: auto {auto:.2f} us ({by / auto / 1e3:.0f} GB/s) | best: " + "  ".join(f"{n} {u:.2f}" for u, n in rows[:6]), flush=True)
