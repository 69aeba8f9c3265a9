#!/usr/bin/env python3
"""GPU tuning sweep for awq_gemm_forward variants (run through gpurun).

For every (shape, M, variant) it cycles a working set >= 600 MB of distinct weight buffers
(defeats L2 + the 256 MiB Infinity Cache), captures the launches in one hipGraph, replays it and
reports microseconds per launch and algorithmic GB/s.  Output: table on stdout +
gpurun_out/sweep_gemv.json.
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from autoawq_amd import ops  # noqa: E402
from bench import algorithmic_bytes, rand_packed  # noqa: E402


def time_variant(sets, x, flags, reps=6):
    outs = [None] * len(sets)

    def run():
        for i, (qw, qz, sc) in enumerate(sets):
            outs[i] = ops.gemm_forward(x, qw, sc, qz, flags=flags)

    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        run()
        run()
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            run()
        g.replay()
        s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            g.replay()
        e1.record(s)
        e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * len(sets)), outs[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--valu", action="store_true")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "sweep_gemv.json"))
    a = ap.parse_args()
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(0)
    shapes = [(4096, 4096), (4096, 12288), (4096, 22016), (11008, 4096)]
    Ms = [1] if a.quick else [1, 4, 8, 16]
    results = []
    for K, N in shapes:
        per = K * N // 2
        nsets = max(4, min(96, (640 << 20) // per))
        sets = [rand_packed(K, N, 128, dev, gen) for _ in range(nsets)]
        for M in Ms:
            x = torch.randn((M, K), device=dev, generator=gen).half()
            ref = ops.gemm_forward(x, sets[0][0], sets[0][2], sets[0][1], flags=ops.gemm_flags(ops.KERNEL_NAIVE)).float()
            by = algorithmic_bytes(K, N, M, 128)
            variants = {}
            if a.valu and M <= 4:
                for nlog in (2, 3):
                    for sk in (0, 8, 16):
                        variants[f"valu n{nlog} s{sk}"] = ops.gemm_flags(ops.KERNEL_VALU, nlog=nlog, splitk=sk)
            for wpl in (2, 4):
                for wv in (2, 4, 8):
                    for un in (2, 4, 8):
                        for sk in ((0, 4, 8, 16, 32) if M == 1 else (0,)):
                            variants[f"mfma w{wpl} v{wv} u{un} s{sk}"] = ops.gemm_flags(ops.KERNEL_MFMA_GEMV, nlog=wpl, splitk=sk, waves=wv, unit=un)
            variants["mfma w2 v4 u4 s0 2pass"] = ops.gemm_flags(ops.KERNEL_MFMA_GEMV, nlog=2, waves=4, unit=4, two_pass=True)
            variants["auto"] = 0
            for name, fl in variants.items():
                try:
                    us, y = time_variant(sets, x, fl)
                except Exception as e:
                    continue
                err = float((y.float() - ref).abs().max() / ref.abs().max())
                gbs = by / us / 1e3
                results.append(dict(K=K, N=N, M=M, variant=name, us=us, gbs=gbs, relerr=err))
                print(f"K{K:6d} N{N:6d} M{M} {name:24s} {us:8.2f} us  {gbs:8.1f} GB/s  ({gbs / 80:.1f}% of 8 TB/s) err {err:.1e}", flush=True)
        del sets
        torch.cuda.empty_cache()
    # dequant kernel
    for K, N in [(4096, 4096), (4096, 11008), (11008, 4096)]:
        qw, qz, sc = rand_packed(K, N, 128, dev, gen)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ops.dequantize_weights(qw, sc, qz)
        e0.record()
        for _ in range(20):
            W = ops.dequantize_weights(qw, sc, qz)
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        by = K * N // 2 + (K // 128) * (N // 8) * 4 + (K // 128) * N * 2 + K * N * 2
        results.append(dict(K=K, N=N, M=0, variant="dequant", us=us, gbs=by / us / 1e3))
        print(f"dequant K{K} N{N}: {us:.1f} us {by / us / 1e3:.0f} GB/s (cache-warm input)")
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(results, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
