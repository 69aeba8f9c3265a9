#!/usr/bin/env python3
"""Split-K / wave / unit sweep of the decode kernel at M = 8 and 16 (the "bs=8 GEMM" of the north star), cold weights."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from autoawq_amd import ops

dev = torch.device("cuda")
gen = torch.Generator(device=dev).manual_seed(0)
K, N = (int(v) for v in (sys.argv[1:3] if len(sys.argv) > 2 else (4096, 11008)))
sets = [bench.rand_packed(K, N, 128, dev, gen) for _ in range(28)]
st = torch.cuda.Stream()
for M in (8, 16):
    x = torch.randn((M, K), device=dev, generator=gen).half()
    ref = ops.gemm_forward(x, sets[0][0], sets[0][2], sets[0][1], flags=ops.gemm_flags(ops.KERNEL_TILED))  # another kernel
    for sk in (0, 6, 8, 12, 16):
        for waves in (0, 4, 8):
            for unit in (0, 2, 4):
                fl = ops.gemm_flags(ops.KERNEL_MFMA_GEMV, nlog=2, splitk=sk, waves=waves, unit=unit) if (sk or waves or unit) else 0
                try:
                    y = ops.gemm_forward(x, sets[0][0], sets[0][2], sets[0][1], flags=fl)
                except Exception:
                    continue
                ok = float((y.float() - ref.float()).abs().max()) < 2e-3 * float(ref.float().abs().max())

                def fn():
                    for qw, qz, sc in sets:
                        ops.gemm_forward(x, qw, sc, qz, flags=fl)

                us = bench.graph_time(fn, st, reps=4) / len(sets)
                print(f"K{K} N{N} M{M:2d} splitk={sk:2d} waves={waves} unit={unit}: {us:6.2f} us {'ok' if ok else 'MISMATCH'}", flush=True)
