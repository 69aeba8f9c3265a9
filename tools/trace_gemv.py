#!/usr/bin/env python3
"""Per-wave phase timeline of the MFMA GEMV kernel (diagnostics; wall_clock64 = 100 MHz ticks).

    python tools/trace_gemv.py --build-only      # here (no GPU): builds tools/bin/libawq_hip_trace.so
    gpurun -- python tools/trace_gemv.py         # on the GPU box

The trace build is the product source compiled with -DAWQ_GEMV_TRACE (extra timestamp stores);
it is never the library that ships or that bench.py measures.
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "autoawq_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "bin", "libawq_hip_trace.so")


def build():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    newest = max(os.path.getmtime(f) for f in srcs + [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")])
    if os.path.exists(OUT) and os.path.getmtime(OUT) >= newest:
        return
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
           "-fno-slp-vectorize", "-Wno-inline-asm", "-DAWQ_GEMV_TRACE", "-DAWQ_BUILDING_LIB", "-I" + os.path.join(ROOT, "include")] + srcs + ["-o", OUT]
    subprocess.check_call(cmd)


if __name__ == "__main__":
    if os.environ.get("AWQ_TRACE_LIB"):  # a prebuilt trace library (A/B of two kernel versions)
        OUT = os.environ["AWQ_TRACE_LIB"]
    else:
        build()
    if "--build-only" in sys.argv:
        sys.exit(0)
    import ctypes
    import numpy as np
    import torch
    from autoawq_amd import _lib
    _lib.LIB_PATH = OUT
    from autoawq_amd import ops
    from bench import rand_packed

    L = _lib.lib()
    L.awq_debug_set_trace.argtypes = [ctypes.c_void_p]
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(0)
    cases = [(4096, 4096, 0, 0, 0, 1), (4096, 4096, 16, 8, 2, 1), (4096, 22016, 8, 4, 4, 1), (11008, 4096, 16, 8, 4, 1)]
    if os.environ.get("CASES"):  # "K,N,splitk,waves,unit,M;..."
        cases = [tuple(int(v) for v in c.split(",")) for c in os.environ["CASES"].split(";")]
    for (K, N, sk, wv, un, MM) in cases:
        per = K * N // 2
        nsets = max(4, min(40, (600 << 20) // per))
        sets = [rand_packed(K, N, 128, dev, gen) for _ in range(nsets)]
        x = torch.randn((MM, K), device=dev, generator=gen).half()
        trace = torch.zeros(8192 * 8 * 16, dtype=torch.int64, device=dev)
        flags = ops.gemm_flags(ops.KERNEL_MFMA_GEMV, nlog=2, splitk=sk, waves=wv, unit=un)
        L.awq_debug_set_trace(None)
        for i in range(nsets - 1):  # thrash caches with other sets, keep the kernel itself warm
            ops.gemm_forward(x, sets[i][0], sets[i][2], sets[i][1], flags=flags)
        torch.cuda.synchronize()
        L.awq_debug_set_trace(trace.data_ptr())
        qw, qz, sc = sets[-1]
        ops.gemm_forward(x, qw, sc, qz, flags=flags)
        torch.cuda.synchronize()
        L.awq_debug_set_trace(None)
        t = trace.cpu().numpy().reshape(-1, 16).astype(np.float64)
        t = t[t[:, 0] != 0]
        t0 = t[:, 0].min()
        t = np.where(t > 0, (t - t0) / 100.0, np.nan)  # us
        nw = t.shape[0]

        def q(a):
            a = a[~np.isnan(a)]
            if a.size == 0:
                return "   (none)"
            return " ".join(f"{v:6.2f}" for v in np.percentile(a, [0, 10, 50, 90, 100])) + f"   n={a.size}"
        print(f"\n=== K{K} N{N} M{MM} splitk={sk} waves={wv} unit={un}: {nw} waves traced; kernel span {np.nanmax(t):.2f} us")
        print("  wave start                (p0 p10 p50 p90 p100):", q(t[:, 0]))
        print("  +staging loads->LDS stores issued              :", q(t[:, 7] - t[:, 0]))
        print("  +first unit's weight loads issued              :", q(t[:, 1] - t[:, 7]))
        print("  +first unit computed (barrier, latency, MFMA)  :", q(t[:, 8] - t[:, 1]))
        print("  +rest of K loop + barrier                      :", q(t[:, 2] - t[:, 8]))
        print("  +LDS write + barrier                           :", q(t[:, 3] - t[:, 2]))
        print("  +producer: block sum + slab stores issued      :", q(t[:, 4] - t[:, 3]))
        print("  +reducer: poll + sum + y stores + re-arm       :", q(t[:, 5] - t[:, 3]))
        print("  abs: K loop done                               :", q(t[:, 2]))
        print("  abs: producers done                            :", q(t[:, 4]))
        print("  abs: reducers done                             :", q(t[:, 5]))
        del sets
        torch.cuda.empty_cache()
