#!/usr/bin/env python3
"""Per-wave phase timeline of the GEMV-layout kernel (diagnostics build, see tools/trace_gemv.py)."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import trace_gemv
trace_gemv.build()
if "--build-only" in sys.argv:
    sys.exit(0)
from autoawq_amd import _lib
_lib.LIB_PATH = trace_gemv.OUT
from autoawq_amd import ops
from tools.sweep_gemv_nk import rand_nk  # noqa: E402  (runs nothing at import? guarded below)

L = _lib.lib()
L.awq_debug_set_trace_nk.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda")
gen = torch.Generator(device=dev).manual_seed(0)
for (K, N, wv, un) in [(4096, 4096, 8, 4), (4096, 22016, 4, 4), (11008, 4096, 8, 8)]:
    per = K * N // 2
    nsets = max(4, min(40, (600 << 20) // per))
    sets = [rand_nk(K, N, 128) for _ in range(nsets)]
    x = torch.randn((1, K), device=dev, generator=gen).half()
    trace = torch.zeros(4096 * 16 * 16, dtype=torch.int64, device=dev)
    flags = ops.gemm_flags(waves=wv, unit=un)
    L.awq_debug_set_trace_nk(None)
    for i in range(nsets - 1):
        ops.gemv_forward(x, sets[i][0], sets[i][2], sets[i][1], 128, flags=flags)
    torch.cuda.synchronize()
    L.awq_debug_set_trace_nk(trace.data_ptr())
    qw, qz, sc = sets[-1]
    ops.gemv_forward(x, qw, sc, qz, 128, flags=flags)
    torch.cuda.synchronize()
    L.awq_debug_set_trace_nk(None)
    t = trace.cpu().numpy().reshape(-1, 16).astype(np.float64)
    t = t[t[:, 0] != 0]
    t0 = t[:, 0].min()
    t = np.where(t > 0, (t - t0) / 100.0, np.nan)

    def q(a):
        a = a[~np.isnan(a)]
        return " ".join(f"{v:6.2f}" for v in np.percentile(a, [0, 10, 50, 90, 100])) + f"   n={a.size}" if a.size else "(none)"
    print(f"\n=== GEMV layout K{K} N{N} waves={wv} unroll={un}: {t.shape[0]} waves; kernel span {np.nanmax(t):.2f} us")
    print("  wave start              (p0 p10 p50 p90 p100):", q(t[:, 0]))
    print("  +staging (x, zeros, scales -> LDS)            :", q(t[:, 1] - t[:, 0]))
    print("  +first batch of weight loads issued           :", q(t[:, 2] - t[:, 1]))
    print("  +staging barrier                              :", q(t[:, 3] - t[:, 2]))
    print("  +K loop (latency + decode + MFMA + folds)     :", q(t[:, 4] - t[:, 3]))
    print("  +LDS fold across waves + y store              :", q(t[:, 5] - t[:, 4]))
    print("  abs: K loop done                              :", q(t[:, 4]))
    del sets
    torch.cuda.empty_cache()
