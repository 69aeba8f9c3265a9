#!/usr/bin/env python3
"""WQLinear_GEMV's prefill routes at 4096 x 11008 by token count: "repack" (csrc/repack.hip + the fused MFMA GEMM on the temporary),
"two_pass" (dequantise + vendor GEMM), "fused" (the N-major form of gemm_regb), the batched-decode kernel in launches of <= 128 rows; and the
repack kernel alone against its bytes (K N / 2 read + written)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from autoawq_amd import ops  # noqa: E402


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(2)
    for K, N in [(4096, 11008), (11008, 4096)]:
        qw, qz, sc = bench.rand_packed_nk(K, N, 128, dev, gen)
        us = timeit(lambda: ops.repack_gemv_to_gemm(qw, sc, qz, 128), reps=50)
        by = K * N  # K N / 2 read + K N / 2 written
        print(f"repack {K}x{N}: {us:.1f} us = {by / us / 1e3:.0f} GB/s of {by / 1e6:.1f} MB")
        gq, gs, gz = ops.repack_gemv_to_gemm(qw, sc, qz, 128)
        wt = ops.dequantize_weights_gemv(qw, sc, qz, 128)
        for M in (64, 96, 128, 192, 256, 384, 512, 1024, 2048, 4096, 16384):
            x = torch.randn((M, K), device=dev, generator=gen).half()
            fl = 2.0 * M * K * N
            r = {}
            r["repack"] = timeit(lambda: ops.gemv_prefill_repack(x, qw, sc, qz, 128))
            r["two_pass"] = timeit(lambda: torch.matmul(x, ops.dequantize_weights_gemv(qw, sc, qz, 128).t()))
            r["fused_nk"] = timeit(lambda: ops.gemv_forward(x, qw, sc, qz, 128, flags=ops.gemm_flags(kernel=ops.GEMV_KERNEL_PREFILL)))
            if M <= 1024:
                r["batch"] = timeit(lambda: ops.gemv_forward(x, qw, sc, qz, 128, flags=ops.gemm_flags(kernel=ops.GEMV_KERNEL_BATCH)))
            # the fused GEMM alone on the repacked temporary, both tile heights (round 6: which one AUTO should take per shape)
            for bm in (1, 2):
                r[f"regb_bm{128 * bm}"] = timeit(lambda: ops.gemm_forward(x, gq, gs, gz, flags=ops.gemm_flags(ops.KERNEL_REGB, nlog=bm)))
            r["vendor_gemm_only"] = timeit(lambda: torch.matmul(x, wt.t()))
            print(f"  M={M}: " + "  ".join(f"{k} {v:.1f} us ({fl / v / 1e6:.0f} TF)" for k, v in r.items()), flush=True)
        # the GEMVFast format's routes (round 6): repack + the FZ form of the fused GEMM, against dequantise + vendor GEMM
        del qw, qz, sc
        fq, fz, fs = bench.rand_packed_nk(K, N, 128, dev, gen, fast=True)
        us = timeit(lambda: ops.repack_gemvfast_to_gemm(fq), reps=50)
        print(f"repack (GEMVFast words) {K}x{N}: {us:.1f} us = {K * N / us / 1e3:.0f} GB/s")
        for M in (64, 128, 192, 256, 512, 2048, 4096, 16384):
            x = torch.randn((M, K), device=dev, generator=gen).half()
            fl = 2.0 * M * K * N
            if M <= 512:
                us_b = timeit(lambda: ops.gemv_fast_forward(x, fq, fs, fz, 128, flags=ops.gemm_flags(kernel=ops.GEMV_KERNEL_BATCH)))
                print(f"  GEMVFast M={M}: batch {us_b:.1f} us ({fl / us_b / 1e6:.0f} TF)")
            r = {"fused_fz": timeit(lambda: ops.gemv_fast_prefill(x, fq, fs, fz, 128)),
                 "fused_fz_bm256": timeit(lambda: ops.gemv_fast_prefill(x, fq, fs, fz, 128, flags=ops.gemm_flags(nlog=2))),
                 "two_pass": timeit(lambda: torch.matmul(x, ops.dequantize_weights_gemv_fast(fq, fs, fz, 128).t()))}
            print(f"  GEMVFast M={M}: " + "  ".join(f"{k} {v:.1f} us ({fl / v / 1e6:.0f} TF)" for k, v in r.items()), flush=True)


if __name__ == "__main__":
    main()
