#!/usr/bin/env python3
"""csrc/gemv_batch.hip above 32 rows: the row parts inside one block (round 6's first form, flags WAVES = 1) against row parts ACROSS
blocks of one XCD (WAVES = 2 | 3 | 4), GEMV and GEMVFast layouts: check against the dequantised weights + fp32 matmul, then time over
distinct matrices (cold weights, one call each per hipGraph replay).
    gpurun -- 'python tools/sweep_batch_parts.py > gpurun_out/sweep_batch_parts.txt 2>&1'"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from autoawq_amd import ops  # noqa: E402

BATCH = 5
ROWS = (8, 16, 24, 32, 33, 48, 64, 80, 96, 128)
# (row parts: 0 = auto, 1 = inside the block, 2 .. 4 = across blocks of one XCD; ring slots per wave: 0 = auto, 1 = one, 2 = two (lazy))
CONFIGS = ((0, 0), (1, 0), (2, 0), (3, 0), (4, 0), (0, 1), (0, 2))


def main():
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(11)
    quick = "--quick" in sys.argv
    bad = 0
    for K, N in [(4096, 11008), (11008, 4096), (4096, 4096), (1024, 8192), (4096, 1040)]:
        qw, qz, sc = bench.rand_packed_nk(K, N, 128, dev, gen)
        wt = ops.dequantize_weights_gemv(qw, sc, qz, 128).float()
        for M in ROWS:
            x = (torch.randn((M, K), device=dev, generator=gen) * 0.5).half()
            ref = x.float() @ wt.t()
            for parts, depth in CONFIGS:
                try:
                    y = ops.gemv_forward(x, qw, sc, qz, 128, flags=ops.gemm_flags(kernel=BATCH, waves=parts, splitk=depth))
                except Exception as e:
                    assert "code -3" in str(e), e
                    continue
                err = (y.float() - ref).abs()
                ok = bool((err <= ref.abs() * 2.0 ** -9 + 2e-2).all()) and bool(torch.isfinite(y).all())
                bad += not ok
                if not ok:
                    print(f"MISMATCH K={K} N={N} M={M} parts={parts} depth={depth}: max err {float(err.max()):.4g}")
    print("check:", "FAILED" if bad else "all within tolerance", flush=True)
    st = torch.cuda.Stream(device=dev)
    shapes = [(4096, 11008)] if quick else [(4096, 11008), (4096, 12288), (4096, 4096), (4096, 22016), (11008, 4096), (8192, 7168), (8192, 28672)]
    for K, N in shapes:
        nsets = max(4, min(28, int(640e6 / (K * N / 2))))
        mats = [bench.rand_packed_nk(K, N, 128, dev, gen) for _ in range(nsets)]
        for M in ROWS:
            x = torch.randn((M, K), device=dev, generator=gen).half()
            line = []
            for parts, depth in CONFIGS:
                fl = ops.gemm_flags(kernel=BATCH, waves=parts, splitk=depth)

                def f():
                    for qw, qz, sc in mats:
                        ops.gemv_forward(x, qw, sc, qz, 128, flags=fl)
                try:
                    us = bench.graph_time(f, st, reps=10, min_seconds=0.1) / len(mats)
                    line.append(f"p{parts}d{depth} {us:6.2f}")
                except Exception:
                    line.append(f"p{parts}d{depth}    - ")
            print(f"K={K} N={N} M={M:3d}: " + "  ".join(line), flush=True)
        del mats
    # the same kernel on the GEMVFast layout's buffers
    for K, N in ([(4096, 11008)] if quick else [(4096, 11008), (4096, 4096), (11008, 4096)]):
        qw, qz, sc = bench.rand_packed_nk(K, N, 128, dev, gen, fast=True)
        wt = ops.dequantize_weights_gemv_fast(qw, sc, qz, 128).float()
        nsets = max(4, min(28, int(640e6 / (K * N / 2))))
        mats = [bench.rand_packed_nk(K, N, 128, dev, gen, fast=True) for _ in range(nsets)]
        for M in (1, 2, 4) + ROWS:
            x = (torch.randn((M, K), device=dev, generator=gen) * 0.5).half()
            ref = x.float() @ wt.t()
            line = []
            for parts, depth in CONFIGS:
                fl = ops.gemm_flags(kernel=BATCH, waves=parts, splitk=depth)
                try:
                    y = ops.gemv_fast_forward(x, qw, sc, qz, 128, flags=fl)
                except Exception as e:
                    assert "code -3" in str(e), e
                    line.append(f"p{parts}d{depth}    - ")
                    continue
                err = (y.float() - ref).abs()
                ok = bool((err <= ref.abs() * 2.0 ** -9 + 2e-2).all()) and bool(torch.isfinite(y).all())
                bad += not ok
                if not ok:
                    print(f"MISMATCH fast K={K} N={N} M={M} parts={parts} depth={depth}: max err {float(err.max()):.4g}")

                def f():
                    for a, b, c in mats:
                        ops.gemv_fast_forward(x, a, c, b, 128, flags=fl)
                us = bench.graph_time(f, st, reps=10, min_seconds=0.1) / len(mats)
                line.append(f"p{parts}d{depth} {us:6.2f}")
            print(f"fast K={K} N={N} M={M:3d}: " + "  ".join(line), flush=True)
        del mats
    print("check (incl. GEMVFast):", "FAILED" if bad else "all within tolerance")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
