#!/usr/bin/env python3
"""Feasibility probe (round 6): what would the row-streaming decode kernel (csrc/gemv_rows.hip) reach on Mixtral's expert
shapes?  Times plain awq_gemv_forward launches at M = 1 / 2 on GEMV-layout matrices of the w1|w3 (4096 -> 28672) and w2
(14336 -> 4096) shapes, `hit` distinct experts per projection, two layers' worth of distinct matrices (cache-honest), and
prints GB/s of the algorithmic bytes -- the grouped launch of modules/fused/moe.py measures 0.43-0.44 of 8 TB/s there.

    gpurun -- python tools/probe_moe_rows.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from autoawq_amd import ops
from bench import algorithmic_bytes, rand_packed_nk, graph_time

if __name__ == "__main__":
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(0)
    H, I, g = 4096, 14336, 128
    for hit in (5, 6, 8):
        for M in (1, 2):
            mats1 = [rand_packed_nk(H, 2 * I, g, dev, gen) for _ in range(2 * hit)]
            mats2 = [rand_packed_nk(I, H, g, dev, gen) for _ in range(2 * hit)]
            x1 = torch.randn((M, H), device=dev, generator=gen).half()
            x2 = torch.randn((M, I), device=dev, generator=gen).half()

            def step():
                for qw, qz, sc in mats1:
                    ops.gemv_forward(x1, qw, sc, qz, g)
                for qw, qz, sc in mats2:
                    ops.gemv_forward(x2, qw, sc, qz, g)

            st = torch.cuda.Stream(device=dev)
            us = graph_time(step, st, reps=10, min_seconds=0.2) / 2  # per "block" = hit experts x (w1 + w2)
            by = hit * (algorithmic_bytes(H, 2 * I, M, g) + algorithmic_bytes(I, H, M, g))
            print(f"hit={hit} M={M}: {us:7.1f} us per block of {2 * hit} launches, {by / 1e6:.0f} MB -> {by / us / 1e3:.0f} GB/s = {by / us / 8e6:.3f} of 8 TB/s", flush=True)
            del mats1, mats2
            torch.cuda.empty_cache()
    # per shape
    for (K, N) in ((H, 2 * I), (I, H)):
        for M in (1, 2, 4):
            mats = [rand_packed_nk(K, N, g, dev, gen) for _ in range(8)]
            x = torch.randn((M, K), device=dev, generator=gen).half()
            def step():
                for qw, qz, sc in mats:
                    ops.gemv_forward(x, qw, sc, qz, g)
            st = torch.cuda.Stream(device=dev)
            us = graph_time(step, st, reps=10, min_seconds=0.2) / 8
            by = algorithmic_bytes(K, N, M, g)
            print(f"K={K} N={N} M={M}: {us:6.2f} us per launch [{ops.last_kernel()}] -> {by / us / 8e6:.3f} of 8 TB/s", flush=True)
            del mats
            torch.cuda.empty_cache()
