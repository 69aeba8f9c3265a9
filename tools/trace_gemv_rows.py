#!/usr/bin/env python3
"""Per-wave phase timeline of the row-streaming GEMV kernel (csrc/gemv_rows.hip; diagnostics build, see tools/trace_gemv.py).
    CASES="K,N,waves,depth,bpc,M;..."  gpurun -- python tools/trace_gemv_rows.py"""
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
from tools.sweep_gemv_rows import rand_nk, ROWS

L = _lib.lib()
L.awq_debug_set_trace_rows.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda")
gen = torch.Generator(device=dev).manual_seed(0)
cases = [(4096, 4096, 0, 0, 0, 1), (4096, 12288, 0, 0, 0, 1), (4096, 22016, 0, 0, 0, 1), (11008, 4096, 0, 0, 0, 1)]  # 0 = the launcher's defaults
if os.environ.get("CASES"):
    cases = [tuple(int(v) for v in c.split(",")) for c in os.environ["CASES"].split(";")]
for (K, N, wv, dp, bpc, M) in cases:
    per = K * N // 2
    nsets = max(4, min(40, (600 << 20) // per))
    sets = [rand_nk(K, N, 128) for _ in range(nsets)]
    x = torch.randn((M, K), device=dev, generator=gen).half()
    trace = torch.zeros(1024 * 8 * 8, dtype=torch.int64, device=dev)
    flags = ops.gemm_flags(kernel=ROWS, waves=wv, unit=dp, splitk=bpc)
    L.awq_debug_set_trace_rows(None)
    for i in range(nsets - 1):
        ops.gemv_forward(x, sets[i][0], sets[i][2], sets[i][1], 128, flags=flags)
    torch.cuda.synchronize()
    L.awq_debug_set_trace_rows(trace.data_ptr())
    qw, qz, sc = sets[-1]
    ops.gemv_forward(x, qw, sc, qz, 128, flags=flags)
    torch.cuda.synchronize()
    L.awq_debug_set_trace_rows(None)
    t = trace.cpu().numpy().reshape(-1, 8).astype(np.float64)
    t = t[t[:, 0] != 0]
    t0 = t[:, 0].min()
    t = np.where(t > 0, (t - t0) / 100.0, np.nan)

    def q(a):
        a = a[~np.isnan(a)]
        return " ".join(f"{v:6.2f}" for v in np.percentile(a, [0, 10, 50, 90, 100])) + f"   n={a.size}" if a.size else "(none)"
    print(f"\n=== rows K{K} N{N} M{M} waves={wv} depth={dp} bpc={bpc}: {t.shape[0]} waves; kernel span {np.nanmax(t):.2f} us")
    print("  wave start                  (p0 p10 p50 p90 p100):", q(t[:, 0]))
    print("  +x DMA + first requests issued                    :", q(t[:, 1] - t[:, 0]))
    print("  +x landed, barrier                                :", q(t[:, 2] - t[:, 1]))
    print("  +x LDS -> registers, constants                    :", q(t[:, 3] - t[:, 2]))
    print("  +first SU's data arrived                          :", q(t[:, 4] - t[:, 3]))
    print("  +stream (all SUs consumed, drain)                 :", q(t[:, 5] - t[:, 4]))
    print("  +final barrier                                    :", q(t[:, 6] - t[:, 5]))
    print("  +fold + y stores issued                           :", q(t[:, 7] - t[:, 6]))
    print("  abs: first data                                   :", q(t[:, 4]))
    print("  abs: stream done                                  :", q(t[:, 5]))
    print("  abs: end                                          :", q(t[:, 7]))
    del sets
    torch.cuda.empty_cache()
