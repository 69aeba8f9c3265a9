#!/usr/bin/env python3
"""Ablation timing of the GEMV kernel phases (tuning only; ablated variants compute garbage)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from autoawq_amd import ops
from bench import algorithmic_bytes, rand_packed
from tools.sweep_gemv import time_variant

dev = torch.device("cuda")
gen = torch.Generator(device=dev).manual_seed(0)
for K, N in [(4096, 4096), (4096, 22016), (11008, 4096)]:
    per = K * N // 2
    nsets = max(4, min(96, (640 << 20) // per))
    sets = [rand_packed(K, N, 128, dev, gen) for _ in range(nsets)]
    x = torch.randn((1, K), device=dev, generator=gen).half()
    by = algorithmic_bytes(K, N, 1, 128)
    for sk in (0, 8, 16, 32):
        for ab, nm in [(0, "full"), (1, "no-splitk-combine"), (2, "no-compute"), (3, "loads+lds only")]:
            us, _ = time_variant(sets, x, ops.gemm_flags(ops.KERNEL_VALU, nlog=3, splitk=sk, ablate=ab))
            print(f"K{K} N{N} s{sk:2d} {nm:20s} {us:7.2f} us {by/us/1e3:7.0f} GB/s", flush=True)
    del sets
    torch.cuda.empty_cache()
