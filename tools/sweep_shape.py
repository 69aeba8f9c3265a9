#!/usr/bin/env python3
"""One-shape sweep of the decode kernel: SHAPE=K,N M=1 python tools/sweep_shape.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from autoawq_amd import ops
from bench import algorithmic_bytes, rand_packed
from sweep_gemv import time_variant
dev = torch.device("cuda"); gen = torch.Generator(device=dev).manual_seed(0)
K, N = [int(v) for v in os.environ.get("SHAPE", "4096,12288").split(",")]
M = int(os.environ.get("M", "1"))
nsets = max(4, min(64, (640 << 20) // (K * N // 2)))
sets = [rand_packed(K, N, 128, dev, gen) for _ in range(nsets)]
x = torch.randn((M, K), device=dev, generator=gen).half()
by = algorithmic_bytes(K, N, M, 128)
rows = []
for rep in range(2):
    us, _ = time_variant(sets, x, 0, reps=10); rows.append((us, f"auto#{rep}"))
for wv in (4, 8):
    for un in (2, 4):
        for sk in (4, 6, 8, 10, 12, 16, 24, 32):
            try:
                us, _ = time_variant(sets, x, ops.gemm_flags(ops.KERNEL_MFMA_GEMV, nlog=2, splitk=sk, waves=wv, unit=un), reps=10)
            except Exception:
                continue
            rows.append((us, f"v{wv} u{un} s{sk}"))
rows.sort()
print(f"K{K} N{N} M{M}: " + "  ".join(f"{n} {u:.2f}" for u, n in rows[:14]), flush=True)
print("   worst: " + "  ".join(f"{n} {u:.2f}" for u, n in rows[-4:]))
