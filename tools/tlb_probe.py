#!/usr/bin/env python3
"""Is the slow start of each GEMV launch address translation?  Times the traced GEMV kernel on
(a) cold weights, (b) weights just read by the previous launch, (c) cold weights whose pages were
touched (one word per 4 KiB / 64 KiB / 2 MiB) right before the launch."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import trace_gemv
trace_gemv.build()
from autoawq_amd import _lib
_lib.LIB_PATH = trace_gemv.OUT
from autoawq_amd import ops
from bench import rand_packed

L = _lib.lib()
L.awq_debug_set_trace.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda")
gen = torch.Generator(device=dev).manual_seed(0)
for (K, N) in [(4096, 4096), (4096, 22016), (11008, 4096)]:
    per = K * N // 2
    nsets = max(4, min(40, (600 << 20) // per))
    sets = [rand_packed(K, N, 128, dev, gen) for _ in range(nsets)]
    x = torch.randn((1, K), device=dev, generator=gen).half()
    trace = torch.zeros(8192 * 8 * 16, dtype=torch.int64, device=dev)
    flags = ops.gemm_flags(ops.KERNEL_MFMA_GEMV, nlog=2)

    def thrash():
        for i in range(nsets - 1):
            ops.gemm_forward(x, sets[i][0], sets[i][2], sets[i][1], flags=flags)

    def traced(label, pre=None):
        thrash()
        qw, qz, sc = sets[-1]
        if pre is not None:
            pre(qw)
        torch.cuda.synchronize()
        trace.zero_()
        L.awq_debug_set_trace(trace.data_ptr())
        ops.gemm_forward(x, qw, sc, qz, flags=flags)
        torch.cuda.synchronize()
        L.awq_debug_set_trace(None)
        t = trace.cpu().numpy().reshape(-1, 16).astype(np.float64)
        t = t[t[:, 0] != 0]
        t0 = t[:, 0].min()
        t = np.where(t > 0, (t - t0) / 100.0, np.nan)
        iss = t[:, 1] - t[:, 0]
        pro = t[:, 7] - t[:, 0]
        c0 = t[:, 8] - t[:, 1]
        i1 = t[:, 9] - t[:, 8]
        c1 = t[:, 10] - t[:, 9]
        kl = t[:, 2]
        end = np.nanmax(t)
        print(f"K{K} N{N} {label:34s} prologue dt p0/p50/p100 {np.nanmin(pro):5.2f} {np.nanmedian(pro):5.2f} {np.nanmax(pro):5.2f} | loads-issued dt p0/p50/p100 {np.nanmin(iss):5.2f} {np.nanmedian(iss):5.2f} {np.nanmax(iss):5.2f} | "
              f"K-loop done abs p50/p100 {np.nanmedian(kl):5.2f} {np.nanmax(kl):5.2f} | span {end:5.2f} us")
        if not np.isnan(i1).all():
            print(f"      it0 compute p50 {np.nanmedian(c0):5.2f} | it1 loads-issued p0/p50/p100 {np.nanmin(i1):5.2f} {np.nanmedian(i1):5.2f} {np.nanmax(i1):5.2f} | it1 compute p50 {np.nanmedian(c1):5.2f}")

    traced("cold")
    traced("cold (repeat)")
    traced("warm: same weights just used", pre=lambda qw: ops.gemm_forward(x, qw, sets[-1][2], sets[-1][1], flags=flags))
    del sets
    torch.cuda.empty_cache()
