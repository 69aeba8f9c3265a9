#!/usr/bin/env python3
"""csrc/gemv_batch.hip on an MI355X: a quick check against the bit-exact dequantised weights + fp32 matmul, then timing by batch
size and configuration (activations through the LDS staging area | direct fragment loads, pieces in flight 1 .. 3) over distinct matrices (cold weights, one call each
per hipGraph replay) beside the older kernels of the layout (gemv_lds / gemv_nk in 16-row chunks).
    gpurun -- 'python tools/sweep_gemv_batch.py > gpurun_out/sweep_gemv_batch.txt 2>&1'"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from autoawq_amd import ops  # noqa: E402

BATCH, LDS, TILE = 5, 3, 1


def main():
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(11)
    quick = "--quick" in sys.argv
    bad = 0
    for K, N in [(4096, 11008), (11008, 4096), (4096, 4096), (1024, 8192)]:
        qw, qz, sc = bench.rand_packed_nk(K, N, 128, dev, gen)
        wt = ops.dequantize_weights_gemv(qw, sc, qz, 128).float()
        for M in (5, 8, 16, 17, 32, 33, 64):
            x = (torch.randn((M, K), device=dev, generator=gen) * 0.5).half()
            ref = x.float() @ wt.t()
            for unit, depth in ((0, 0), (1, 1), (1, 2), (2, 1), (2, 3)):
                try:
                    y = ops.gemv_forward(x, qw, sc, qz, 128, flags=ops.gemm_flags(kernel=BATCH, unit=unit, splitk=depth))
                except Exception as e:  # the staged form where it does not fit
                    assert "code -3" in str(e) and unit == 1, e
                    continue
                err = (y.float() - ref).abs()
                ok = bool((err <= ref.abs() * 2.0 ** -9 + 2e-2).all()) and bool(torch.isfinite(y).all())
                bad += not ok
                if not ok:
                    print(f"MISMATCH K={K} N={N} M={M} unit={unit} depth={depth}: max err {float(err.max()):.4g}")
    print("check:", "FAILED" if bad else "all within tolerance")
    st = torch.cuda.Stream(device=dev)
    shapes = [(4096, 11008)] if quick else [(4096, 11008), (4096, 12288), (4096, 4096), (4096, 22016), (11008, 4096), (8192, 1280), (1024, 8192), (8192, 7168), (3584, 8192)]
    for K, N in shapes:
        nsets = max(4, min(28, int(640e6 / (K * N / 2))))
        mats = [bench.rand_packed_nk(K, N, 128, dev, gen) for _ in range(nsets)]
        for M in ((1, 4, 5, 8, 16, 17, 32, 48, 64, 128) if (K, N) == (4096, 11008) else (8, 16, 32, 64)):
            x = torch.randn((M, K), device=dev, generator=gen).half()
            line = []
            configs = [("auto", 0)]
            if M >= 5:
                configs += [(f"{'xs' if u == 1 else 'direct'}-d{d}", ops.gemm_flags(kernel=BATCH, unit=u, splitk=d)) for u, d in ((1, 1), (1, 2), (1, 3), (2, 1), (2, 2))]
            if M <= 16 and (K, N) == (4096, 11008):
                configs += [("lds", ops.gemm_flags(kernel=LDS)), ("tile16", ops.gemm_flags(kernel=TILE))]
            for name, fl in configs:
                def f():
                    for qw, qz, sc in mats:
                        ops.gemv_forward(x, qw, sc, qz, 128, flags=fl)
                try:
                    us = bench.graph_time(f, st, reps=10, min_seconds=0.1) / len(mats)
                    k = ops.last_kernel()
                    line.append(f"{name}[{k}] {us:.2f}")
                except Exception as e:  # refused
                    line.append(f"{name} -")
            best = bench.algorithmic_bytes(K, N, M, 128)
            print(f"K={K} N={N} M={M}: " + "  ".join(line) + f"   (algorithmic {best / 1e6:.1f} MB)", flush=True)
        del mats
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
