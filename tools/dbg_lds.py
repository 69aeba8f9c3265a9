#!/usr/bin/env python3
"""Bring-up + timing probe for csrc/gemv_lds.hip (GEMV layout, 2 <= M <= 16 through LDS + MFMA)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from autoawq_amd import ops
from tools.sweep_gemv_rows import rand_nk, graph_us, gen, dev
LDS = 3
for K, N in [(4096, 64), (4096, 4096), (4096, 11008), (4096, 22016), (11008, 4096), (1024, 200), (2048, 4099), (8192, 1280)]:
    qw, qz, sc = rand_nk(K, N, 128)
    Wt = ops.dequantize_weights_gemv(qw, sc, qz, 128).float()
    for M in (2, 3, 4, 8, 16):
        x = torch.randn((M, K), device=dev, generator=gen).half()
        ref = x.float() @ Wt.t()
        for fl in (dict(), dict(splitk=1), dict(splitk=2), dict(splitk=4), dict(unit=2), dict(unit=3, splitk=2)):
            try:
                y = ops.gemv_forward(x, qw, sc, qz, 128, flags=ops.gemm_flags(kernel=LDS, **fl)).float()
            except Exception as e:
                if "code -3" not in str(e):
                    print(f"K{K} N{N} M{M} {fl}: {e}")
                continue
            torch.cuda.synchronize()
            err = (y - ref).abs()
            tol = 2e-3 * ref.abs() + 2e-3 * ref.pow(2).mean().sqrt()
            ok = bool((err <= tol).all())
            print(f"K{K} N{N} M{M} {fl} {ops.last_kernel()}: {'ok' if ok else 'MISMATCH'} max err {float(err.max()):.3g} bad {int((err > tol).sum())}/{err.numel()}", flush=True)
for K, N in [(4096, 11008), (4096, 4096), (4096, 22016)]:
    nsets = max(4, min(96, (640 << 20) // (K * N // 2)))
    sets = [rand_nk(K, N, 128) for _ in range(nsets)]
    for M in (2, 4, 8):
        x = torch.randn((M, K), device=dev, generator=gen).half()
        for name, f in [("auto", 0), ("lds8 ks1", ops.gemm_flags(kernel=LDS, splitk=1)), ("lds8 ks2", ops.gemm_flags(kernel=LDS, splitk=2)),
                        ("lds8 ks4", ops.gemm_flags(kernel=LDS, splitk=4)), ("lds4d2 k1", ops.gemm_flags(kernel=LDS, splitk=1, unit=2)),
                        ("lds4d2 k4", ops.gemm_flags(kernel=LDS, splitk=4, unit=2)), ("tile16", ops.gemm_flags(kernel=1))]:
            def run():
                for qw, qz, sc in sets:
                    ops.gemv_forward(x, qw, sc, qz, 128, flags=f)
            try:
                us = graph_us(run, nsets)
            except Exception as e:
                print(f"K{K} N{N} M{M} {name}: {str(e)[:80]}")
                continue
            print(f"K{K} N{N} M{M} {name:10s} {ops.last_kernel():10s} {us:7.2f} us", flush=True)
    del sets
    torch.cuda.empty_cache()
