#!/usr/bin/env python3
"""Batched-decode regime (M = 1..64) on the BASELINE shapes: decode kernel (auto), chunked decode
kernel (ceil(M/16) launches), fused tiled kernel, two-pass (dequant + fp16 GEMM).  Cold weights."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from autoawq_amd import ops
from bench import algorithmic_bytes, rand_packed

dev = torch.device("cuda")
gen = torch.Generator(device=dev).manual_seed(0)


def time_graph(fn_per_set, nsets, reps=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(nsets):
            fn_per_set(i)
        s.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for i in range(nsets):
                fn_per_set(i)
        gr.replay()
        s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            gr.replay()
        e1.record(s)
        e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * nsets)


def main():
    for K, N in [(4096, 11008), (4096, 4096), (11008, 4096)]:
        per = K * N // 2
        nsets = max(4, min(48, (640 << 20) // per))
        sets = [rand_packed(K, N, 128, dev, gen) for _ in range(nsets)]
        for M in [1, 2, 4, 8, 12, 16, 24, 32, 48, 64]:
            x = torch.randn((M, K), device=dev, generator=gen).half()
            by = algorithmic_bytes(K, N, M, 128)
            row = f"K{K} N{N} M{M:3d}:"

            def gemv_chunked(i):
                qw, qz, sc = sets[i]
                return torch.cat([ops.gemm_forward(x[m:m + 16], qw, sc, qz) for m in range(0, M, 16)])

            variants = [("gemv(chunks of 16)", gemv_chunked)]
            if M > 16:
                variants.append(("tiled", lambda i: ops.gemm_forward(x, sets[i][0], sets[i][2], sets[i][1], flags=ops.gemm_flags(ops.KERNEL_TILED, nlog=1))))
            variants.append(("two-pass", lambda i: torch.matmul(x, ops.dequantize_weights(sets[i][0], sets[i][2], sets[i][1]))))
            for nm, f in variants:
                us = time_graph(f, nsets)
                row += f"  {nm} {us:7.1f} us ({by / us / 1e3:6.0f} GB/s)"
            print(row, flush=True)
        del sets
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
