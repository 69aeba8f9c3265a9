#!/usr/bin/env python3
"""Debug helper: compare one gemm_forward variant against the naive kernel on a small case."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from autoawq_amd import ops
from bench import rand_packed

dev = torch.device("cuda")
gen = torch.Generator(device=dev).manual_seed(0)
for (K, N, g, M, fl, nm) in [
    (128, 256, 128, 1, ops.gemm_flags(ops.KERNEL_MFMA_GEMV, nlog=2, splitk=1), "w2 s1"),
    (256, 256, 128, 1, ops.gemm_flags(ops.KERNEL_MFMA_GEMV, nlog=2, splitk=1), "w2 s1 K256"),
    (4096, 4096, 128, 1, ops.gemm_flags(ops.KERNEL_MFMA_GEMV, nlog=2, splitk=1), "w2 s1 big"),
    (4096, 4096, 128, 1, ops.gemm_flags(ops.KERNEL_MFMA_GEMV, nlog=2, splitk=8), "w2 s8 big"),
    (128, 512, 128, 1, ops.gemm_flags(ops.KERNEL_MFMA_GEMV, nlog=4, splitk=1), "w4 s1"),
    (128, 256, 128, 3, ops.gemm_flags(ops.KERNEL_MFMA_GEMV, nlog=2, splitk=1), "w2 s1 M3"),
    (128, 256, 128, 12, ops.gemm_flags(ops.KERNEL_MFMA_GEMV, nlog=2, splitk=1), "w2 s1 M12"),
    (4096, 4096, 128, 1, 0, "auto big"),
    (512, 1056, 32, 3, 0, "auto ragged g32"),
]:
    qw, qz, sc = rand_packed(K, N, g, dev, gen)
    x = torch.randn((M, K), device=dev, generator=gen).half()
    ref = ops.gemm_forward(x, qw, sc, qz, flags=ops.gemm_flags(ops.KERNEL_NAIVE)).float()
    y = ops.gemm_forward(x, qw, sc, qz, flags=fl).float()
    torch.cuda.synchronize()
    bad = ~torch.isfinite(y)
    d = (y - ref).abs()
    d[bad] = 0
    print(f"{nm}: K{K} N{N} M{M}: nan {int(bad.sum())}/{y.numel()}  maxerr {float(d.max()):.4f} (ref max {float(ref.abs().max()):.2f})")
    if bad.any() or d.max() > 0.05:
        idx = torch.nonzero(bad | (d > 0.05))
        print("   first bad idx:", idx[:12].tolist())
        print("   y  :", y[0, :16].tolist())
        print("   ref:", ref[0, :16].tolist())
