#!/usr/bin/env python3
"""Probe (round 6): dense batch-2..8 decode on the GEMV layout as M virtual batch-1 calls of the row-streaming kernel in ONE launch
(the MoE decode launch with a single expert: awq_grouped_gemv_forward, E = 1) against awq_gemv_forward's AUTO kernels.
    gpurun -- python tools/probe_pairs_dense.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from autoawq_amd import ops
from bench import rand_packed_nk, graph_time

if __name__ == "__main__":
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(0)
    g = 128
    for K, N in ((4096, 11008), (4096, 12288), (4096, 4096), (4096, 22016), (11008, 4096)):
        nsets = max(8, min(32, (700 << 20) // (K * N // 2)))
        mats = [rand_packed_nk(K, N, g, dev, gen) for _ in range(nsets)]
        st = torch.cuda.Stream(device=dev)
        for M in (1, 2, 3, 4, 6, 8):
            x = torch.randn((M, K), device=dev, generator=gen).half()
            ids = torch.zeros((M, 1), dtype=torch.int32, device=dev)

            def auto():
                for qw, qz, sc in mats:
                    ops.gemv_forward(x, qw, sc, qz, g)

            def pairs(parts=0):
                for qw, qz, sc in mats:
                    ops.grouped_gemv_forward(x, qw.unsqueeze(0), sc.unsqueeze(0), qz.unsqueeze(0), ids, g, parts=parts)

            ya = ops.gemv_forward(x, *[mats[0][i] for i in (0, 2, 1)], g)
            ka = ops.last_kernel()
            yp = ops.grouped_gemv_forward(x, mats[0][0].unsqueeze(0), mats[0][2].unsqueeze(0), mats[0][1].unsqueeze(0), ids, g).view(M, N)
            rel = float((ya.float() - yp.float()).abs().max() / ya.float().abs().max())
            ua = graph_time(auto, st, 10, min_seconds=0.1) / nsets
            line = f"K={K} N={N} M={M}: auto[{ka}] {ua:6.2f} us   pairs:"
            for parts in (0, 64, 128, 256):
                line += f"  parts {parts}: {graph_time(lambda: pairs(parts), st, 10, min_seconds=0.1) / nsets:6.2f}"
            print(line + f"   (max rel diff {rel:.1e})", flush=True)
        del mats
        torch.cuda.empty_cache()
