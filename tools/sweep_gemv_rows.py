#!/usr/bin/env python3
"""GPU check + tuning sweep for the row-streaming GEMV-layout kernel (csrc/gemv_rows.hip): correctness against the
fp32 product of the bit-exact dequantised weights, then (waves, ring, blocks per CU) per shape and batch, cold weights
(>= 600 MB of distinct matrices cycled inside one hipGraph), next to the 16-row MFMA tile kernel and the GEMM layout."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from autoawq_amd import ops
from autoawq_amd.utils.packing import calculate_zeros_width

dev = torch.device("cuda")
gen = torch.Generator(device=dev).manual_seed(0)
lim = 0x7FFFFFFF
ROWS, TILE16 = 2, 1


def rand_nk(K, N, g):
    zw = calculate_zeros_width(K, g)
    qw = torch.randint(-lim - 1, lim, (N, K // 8), dtype=torch.int32, device=dev, generator=gen)
    qz = torch.randint(-lim - 1, lim, (N, zw), dtype=torch.int32, device=dev, generator=gen)
    sc = (torch.rand((N, zw * 8), device=dev, generator=gen) * 0.02 + 0.005).half()
    return qw, qz, sc


def rand_gemm(K, N, g):
    qw = torch.randint(-lim - 1, lim, (K, N // 8), dtype=torch.int32, device=dev, generator=gen)
    qz = torch.randint(-lim - 1, lim, (K // g, N // 8), dtype=torch.int32, device=dev, generator=gen)
    sc = (torch.rand((K // g, N), device=dev, generator=gen) * 0.02 + 0.005).half()
    return qw, qz, sc


def graph_us(run, n, reps=6):
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
    return e0.elapsed_time(e1) * 1e3 / (reps * n)


def check():
    bad = 0
    for K, N, g in [(4096, 4096, 128), (11008, 4096, 128), (4096, 11008, 128), (1024, 72, 64), (512, 40, 32), (2048, 200, 2048),
                    (256, 16, 128), (8192, 1280, 128), (3584, 8192, 128), (28672, 1024, 128), (4096, 22016, 128), (1280, 10, 256), (384, 7, 128),
                    (4096, 4099, 128)]:
        qw, qz, sc = rand_nk(K, N, g)
        Wt = ops.dequantize_weights_gemv(qw, sc, qz, g).float()
        for M in (1, 2, 3, 4):
            x = torch.randn((M, K), device=dev, generator=gen).half()
            ref = x.float() @ Wt.t()
            for fl in (ops.gemm_flags(kernel=ROWS), ops.gemm_flags(kernel=ROWS, waves=4, unit=1, splitk=1), ops.gemm_flags(kernel=ROWS, unit=2, splitk=3),
                       ops.gemm_flags(kernel=ROWS, nlog=1), ops.gemm_flags(kernel=ROWS, nlog=2), ops.gemm_flags(kernel=ROWS, nlog=3),
                       ops.gemm_flags(kernel=ROWS, nlog=4), ops.gemm_flags(kernel=ROWS, nlog=6), ops.gemm_flags(kernel=ROWS, nlog=8)):
                try:
                    y = ops.gemv_forward(x, qw, sc, qz, g, flags=fl).float()
                except Exception as e:
                    if "code -3" not in str(e):
                        print(f"K{K} N{N} g{g} M{M} f{fl:x}: {e}")
                    continue
                err = (y - ref).abs()
                tol = 2e-3 * ref.abs() + 2e-3 * ref.pow(2).mean().sqrt()
                ok = bool((err <= tol).all())
                bad += not ok
                if not ok:
                    print(f"MISMATCH K{K} N{N} g{g} M{M} f{fl:x} kernel {ops.last_kernel()} max err {float(err.max()):.4g} "
                          f"rms ref {float(ref.pow(2).mean().sqrt()):.4g} bad {int((err > tol).sum())}/{err.numel()}")
    print("check:", "OK" if bad == 0 else f"{bad} FAILED", flush=True)


def main():
    check()
    for K, N in [(4096, 4096), (4096, 12288), (4096, 22016), (11008, 4096)]:
        g = 128
        per = K * N // 2
        nsets = max(4, min(96, (640 << 20) // per))
        sets = [rand_nk(K, N, g) for _ in range(nsets)]
        for M in (1, 2, 4):
            x = torch.randn((M, K), device=dev, generator=gen).half()
            by = K * N // 2 + (K // g) * N // 2 + (K // g) * N * 2 + M * K * 2 + M * N * 2
            variants = [("rows auto", ops.gemm_flags(kernel=ROWS))]
            if M == 1:
                variants += [(f"rows s{sl} w{w} d{u} b{b}", ops.gemm_flags(kernel=ROWS, waves=w, unit=u, splitk=b, nlog=sl))
                             for sl in ((1, 2) if K == 4096 else (1, 2, 3, 6)) for w in (4, 8) for u in (1, 2) for b in (1, 2, 3)]
            variants += [("tile16 (nk)", ops.gemm_flags(kernel=TILE16))]
            for name, fl in variants:
                outs = [None] * nsets

                def run():
                    for i, (qw, qz, sc) in enumerate(sets):
                        outs[i] = ops.gemv_forward(x, qw, sc, qz, g, flags=fl)
                try:
                    us = graph_us(run, nsets)
                except Exception as e:
                    print(f"K{K} N{N} M{M} {name}: {e}")
                    continue
                print(f"K{K:6d} N{N:6d} M{M:2d} {name:18s} {ops.last_kernel():10s} {us:8.2f} us  {by / us / 1e3:8.1f} GB/s  ({by / us / 80e3:.1f}% of 8 TB/s)", flush=True)
        del sets
        torch.cuda.empty_cache()
        sets = [rand_gemm(K, N, g) for _ in range(nsets)]
        x = torch.randn((1, K), device=dev, generator=gen).half()

        def run2():
            for qw, qz, sc in sets:
                ops.gemm_forward(x, qw, sc, qz)
        us = graph_us(run2, nsets)
        by = K * N // 2 + (K // g) * N // 2 + (K // g) * N * 2 + K * 2 + N * 2
        print(f"K{K:6d} N{N:6d} M 1 GEMM layout        {ops.last_kernel():10s} {us:8.2f} us  {by / us / 1e3:8.1f} GB/s  ({by / us / 80e3:.1f}% of 8 TB/s)", flush=True)
        del sets
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
