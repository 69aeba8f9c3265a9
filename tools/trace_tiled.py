#!/usr/bin/env python3
"""Per-wave phase timeline of the fused tiled kernel (diagnostics; wall_clock64 = 100 MHz ticks).
    gpurun -- python tools/trace_tiled.py     (builds tools/bin/libawq_hip_trace.so via trace_gemv)"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import trace_gemv
trace_gemv.build()
if "--build-only" in sys.argv:
    sys.exit(0)
import numpy as np
import torch
from autoawq_amd import _lib
_lib.LIB_PATH = trace_gemv.OUT
from autoawq_amd import ops
from bench import rand_packed

L = _lib.lib()
L.awq_debug_set_trace_tiled.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda")
gen = torch.Generator(device=dev).manual_seed(0)
ops.workspace(dev, 16384 + (64 << 20))
cases = [(4096, 11008, 32, 0), (4096, 11008, 32, 2), (4096, 4096, 32, 8), (4096, 11008, 128, 4)]
for (K, N, M, sk) in cases:
    per = K * N // 2
    nsets = max(4, min(24, (600 << 20) // per))
    sets = [rand_packed(K, N, 128, dev, gen) for _ in range(nsets)]
    x = torch.randn((M, K), device=dev, generator=gen).half()
    trace = torch.zeros(4096 * 8 * 16, dtype=torch.int64, device=dev)
    flags = ops.gemm_flags(ops.KERNEL_TILED, nlog=1, splitk=sk)
    L.awq_debug_set_trace_tiled(None)
    for i in range(nsets - 1):
        ops.gemm_forward(x, sets[i][0], sets[i][2], sets[i][1], flags=flags)
    torch.cuda.synchronize()
    L.awq_debug_set_trace_tiled(trace.data_ptr())
    qw, qz, sc = sets[-1]
    ops.gemm_forward(x, qw, sc, qz, flags=flags)
    torch.cuda.synchronize()
    L.awq_debug_set_trace_tiled(None)
    t = trace.cpu().numpy().reshape(-1, 16).astype(np.float64)
    t = t[t[:, 0] != 0]
    t0 = t[:, 0].min()
    t = np.where(t > 0, (t - t0) / 100.0, np.nan)

    def q(a):
        a = a[~np.isnan(a)]
        if a.size == 0:
            return "   (none)"
        return " ".join(f"{v:6.2f}" for v in np.percentile(a, [0, 10, 50, 90, 100])) + f"   n={a.size}"
    print(f"\n=== K{K} N{N} M{M} splitk={sk}: {t.shape[0]} waves traced; kernel span {np.nanmax(t):.2f} us")
    print("  wave start                 (p0 p10 p50 p90 p100):", q(t[:, 0]))
    print("  +prologue loads issued                          :", q(t[:, 1] - t[:, 0]))
    print("  +first tile arrived, decoded, barrier           :", q(t[:, 2] - t[:, 1]))
    print("  +K loop                                         :", q(t[:, 3] - t[:, 2]))
    print("  +producer: slab stores issued                   :", q(t[:, 4] - t[:, 3]))
    print("  +reducer: polls + sums + re-arm                 :", q(t[:, 5] - t[:, 3]))
    print("  +epilogue (y stores issued)                     :", q(t[:, 6] - t[:, 5]))
    print("  abs: K loop done                                :", q(t[:, 3]))
    print("  abs: producers done                             :", q(t[:, 4]))
    print("  abs: reducers done                              :", q(t[:, 6]))
    del sets
    torch.cuda.empty_cache()
