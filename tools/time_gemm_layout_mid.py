#!/usr/bin/env python3
"""GEMM layout (the default checkpoint format), 65 .. 1024 rows: what awq_gemm_forward's AUTO takes (gemm_tiled with split-K, gemm_regb from
256 tiles) against gemm_skinny walked in chunks of <= 64 rows and dequantise + dense GEMM.  Distinct matrices per call (cold weights).
    gpurun -- 'python tools/time_gemm_layout_mid.py > gpurun_out/gemm_layout_mid.txt 2>&1'"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from autoawq_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(4)
    st = torch.cuda.Stream(device=dev)
    for K, N in [(4096, 11008), (11008, 4096), (4096, 4096), (4096, 12288)]:
        nsets = max(4, min(28, int(640e6 / (K * N / 2))))
        mats = [bench.rand_packed(K, N, 128, dev, gen) for _ in range(nsets)]
        for M in (48, 64, 65, 96, 128, 192, 256, 384, 512, 1024):
            x = torch.randn((M, K), device=dev, generator=gen).half()
            res = {}

            def auto():
                for qw, qz, sc in mats:
                    ops.gemm_forward(x, qw, sc, qz)
            res["auto"] = bench.graph_time(auto, st, reps=5, min_seconds=0.05) / nsets
            kern = ops.last_kernel()
            nch = (M + 63) // 64
            rows = (M + nch - 1) // nch
            parts = [x[i:i + rows] for i in range(0, M, rows)]

            def chunks():
                for qw, qz, sc in mats:
                    for xp in parts:
                        ops.gemm_forward(xp, qw, sc, qz, flags=ops.gemm_flags(ops.KERNEL_SKINNY))
            try:
                res[f"skinny x{nch}"] = bench.graph_time(chunks, st, reps=5, min_seconds=0.05) / nsets
            except Exception as e:
                res[f"skinny x{nch}"] = float("nan")

            def two_pass():
                for qw, qz, sc in mats:
                    torch.matmul(x, ops.dequantize_weights(qw, sc, qz))
            res["dequant+gemm"] = bench.graph_time(two_pass, st, reps=5, min_seconds=0.05) / nsets
            print(f"K={K} N={N} M={M:4d}: auto[{kern}] {res['auto']:7.2f}  " + "  ".join(f"{k} {v:7.2f}" for k, v in res.items() if k != "auto"), flush=True)
        del mats
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
