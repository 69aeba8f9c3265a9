#!/usr/bin/env python3
"""GPU tuning sweep for the GEMV-layout kernel (awq_gemv_forward): waves x unroll per shape, cold
weights (working set >= 600 MB cycled inside one hipGraph)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from autoawq_amd import ops
from autoawq_amd.utils.packing import calculate_zeros_width

dev = torch.device("cuda")
gen = torch.Generator(device=dev).manual_seed(0)
lim = 0x7FFFFFFF


def rand_nk(K, N, g):
    zw = calculate_zeros_width(K, g)
    qw = torch.randint(-lim - 1, lim, (N, K // 8), dtype=torch.int32, device=dev, generator=gen)
    qz = torch.randint(-lim - 1, lim, (N, zw), dtype=torch.int32, device=dev, generator=gen)
    sc = (torch.rand((N, zw * 8), device=dev, generator=gen) * 0.02 + 0.005).half()
    return qw, qz, sc


def time_variant(sets, x, g, flags, reps=6):
    outs = [None] * len(sets)

    def run():
        for i, (qw, qz, sc) in enumerate(sets):
            outs[i] = ops.gemv_forward(x, qw, sc, qz, g, flags=flags)

    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        run(); run()
        s.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            run()
        gr.replay()
        s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            gr.replay()
        e1.record(s)
        e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * len(sets))


def main():
  for K, N in [(4096, 4096), (4096, 12288), (4096, 22016), (11008, 4096)]:
      g = 128
      per = K * N // 2
      nsets = max(4, min(96, (640 << 20) // per))
      sets = [rand_nk(K, N, g) for _ in range(nsets)]
      for M in (1, 8, 16):
          x = torch.randn((M, K), device=dev, generator=gen).half()
          by = K * N // 2 + (K // g) * N // 2 + (K // g) * N * 2 + M * K * 2 + M * N * 2
          for wv, un in [(0, 0), (4, 4), (4, 8), (8, 4), (8, 8), (16, 4)]:
              try:
                  us = time_variant(sets, x, g, ops.gemm_flags(waves=wv, unit=un))
              except Exception as e:
                  print(f"K{K} N{N} M{M} v{wv} u{un}: {e}")
                  continue
              print(f"K{K:6d} N{N:6d} M{M:2d} nk v{wv:2d} u{un}   {us:8.2f} us  {by / us / 1e3:8.1f} GB/s  ({by / us / 80e3:.1f}% of 8 TB/s)", flush=True)
      del sets
      torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
