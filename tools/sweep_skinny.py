#!/usr/bin/env python3
"""Batched decode, 17 <= M <= 64: the register-decoded skinny kernel (csrc/gemm_skinny.hip) over K splits vs the LDS-tiled
kernel and dequant + vendor GEMM, cold weights (>= 600 MB of distinct matrices per shape), hipGraph-timed."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from autoawq_amd import ops

dev = torch.device("cuda")
gen = torch.Generator(device=dev).manual_seed(0)
st = torch.cuda.Stream()
shapes = [(4096, 11008), (4096, 4096), (11008, 4096), (4096, 22016)]
if len(sys.argv) > 2:
    shapes = [(int(sys.argv[1]), int(sys.argv[2]))]
for K, N in shapes:
    per = K * N // 2
    sets = [bench.rand_packed(K, N, 128, dev, gen) for _ in range(max(4, min(40, (640 << 20) // per)))]
    for M in (17, 32, 48, 64):
        x = torch.randn((M, K), device=dev, generator=gen).half()
        ref = ops.gemm_forward(x, sets[0][0], sets[0][2], sets[0][1], flags=ops.gemm_flags(ops.KERNEL_TILED))
        row = f"K{K} N{N} M{M:2d}:"
        for name, fl in [("tiled", ops.gemm_flags(ops.KERNEL_TILED))] + [(f"skinny s{sk}", ops.gemm_flags(ops.KERNEL_SKINNY, splitk=sk)) for sk in (0, 2, 4, 6, 8, 12, 16)]:
            try:
                y = ops.gemm_forward(x, sets[0][0], sets[0][2], sets[0][1], flags=fl)
            except Exception:
                continue
            ok = float((y.float() - ref.float()).abs().max()) < 2e-3 * float(ref.float().abs().max())

            def fn():
                for qw, qz, sc in sets:
                    ops.gemm_forward(x, qw, sc, qz, flags=fl)

            us = bench.graph_time(fn, st, reps=4) / len(sets)
            row += f"  {name} {us:6.2f}{'' if ok else ' MISMATCH'}"
        print(row, flush=True)
    del sets
    torch.cuda.empty_cache()
