#!/usr/bin/env python3
"""Where should AUTO hand over from the row-streaming kernel (gemv_rows, M <= 4) / the 16-row GEMVFast kernel to csrc/gemv_batch.hip?
M = 1 .. 5 on both layouts: parity of the FORCED batched kernel against the bit-exact dequantised weights + fp32 matmul, then
us per call, AUTO beside forced BATCH, over distinct matrices (cold weights, one call each per hipGraph replay).
    gpurun -- 'python tools/sweep_small_batch.py > gpurun_out/sweep_small_batch.txt 2>&1'"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from autoawq_amd import ops  # noqa: E402

BATCH = ops.GEMV_KERNEL_BATCH
SHAPES = [(4096, 11008), (4096, 4096), (4096, 12288), (4096, 22016), (11008, 4096),  # 7B (qkv and gate|up fused)
          (8192, 1280), (1024, 8192), (8192, 7168), (3584, 8192)]                   # 70B TP = 8 per-rank shards


def main():
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(23)
    bad = 0
    for fast in (False, True):
        for K, N in [(4096, 11008), (11008, 4096), (4096, 4096), (1024, 8192)]:
            qw, qz, sc = bench.rand_packed_nk(K, N, 128, dev, gen, fast=fast)
            wt = (ops.dequantize_weights_gemv_fast if fast else ops.dequantize_weights_gemv)(qw, sc, qz, 128).float()
            for M in (1, 2, 3, 4):
                x = (torch.randn((M, K), device=dev, generator=gen) * 0.5).half()
                ref = x.float() @ wt.t()
                fwd = ops.gemv_fast_forward if fast else ops.gemv_forward
                y = fwd(x, qw, sc, qz, 128, flags=ops.gemm_flags(kernel=BATCH))
                assert ops.last_kernel().startswith("gemv_batch"), ops.last_kernel()
                err = (y.float() - ref).abs()
                ok = bool((err <= ref.abs() * 2.0 ** -9 + 2e-2).all()) and bool(torch.isfinite(y).all())
                bad += not ok
                if not ok:
                    print(f"MISMATCH fast={fast} K={K} N={N} M={M}: max err {float(err.max()):.4g}")
    print("check:", "FAILED" if bad else "all within tolerance", flush=True)
    st = torch.cuda.Stream(device=dev)
    for fast in (False, True):
        fwd = ops.gemv_fast_forward if fast else ops.gemv_forward
        for K, N in SHAPES:
            nsets = max(4, min(28, int(640e6 / (K * N / 2))))
            mats = [bench.rand_packed_nk(K, N, 128, dev, gen, fast=fast) for _ in range(nsets)]
            for M in (1, 2, 3, 4, 5):
                x = torch.randn((M, K), device=dev, generator=gen).half()
                line = []
                for name, fl in (("auto", 0), ("batch", ops.gemm_flags(kernel=BATCH))):
                    def f():
                        for qw, qz, sc in mats:
                            fwd(x, qw, sc, qz, 128, flags=fl)
                    try:
                        us = bench.graph_time(f, st, reps=10, min_seconds=0.1) / len(mats)
                        line.append(f"{name}[{ops.last_kernel()}] {us:.2f}")
                    except Exception as e:  # refused
                        line.append(f"{name} - ({str(e)[:40]})")
                print(f"{'gemvfast' if fast else 'gemv'} K={K} N={N} M={M}: " + "  ".join(line), flush=True)
            del mats
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
