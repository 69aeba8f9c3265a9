#!/usr/bin/env python3
"""Per-wave phase timeline of the skinny MFMA kernel (diagnostics; wall_clock64 = 100 MHz ticks)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from autoawq_amd import _lib, ops
from bench import rand_packed

dev = torch.device("cuda")
gen = torch.Generator(device=dev).manual_seed(0)
L = _lib.lib()
for (K, N, sk) in [(4096, 4096, 0), (4096, 22016, 4), (4096, 22016, 8), (11008, 4096, 0)]:
    per = K * N // 2
    nsets = max(4, min(40, (600 << 20) // per))
    sets = [rand_packed(K, N, 128, dev, gen) for _ in range(nsets)]
    x = torch.randn((1, K), device=dev, generator=gen).half()
    trace = torch.zeros(4096 * 4 * 8, dtype=torch.int64, device=dev)
    L.awq_hip_set_trace_buffer(trace.data_ptr())
    flags = ops.gemm_flags(ops.KERNEL_SKINNY, nlog=2, splitk=sk)
    for i in range(nsets - 1):  # thrash caches with other sets
        ops.gemm_forward(x, sets[i][0], sets[i][2], sets[i][1], flags=flags)
    torch.cuda.synchronize()
    qw, qz, sc = sets[-1]
    ops.gemm_forward(x, qw, sc, qz, flags=flags | (1 << 24))
    torch.cuda.synchronize()
    t = trace.cpu().numpy().reshape(-1, 4, 8)
    used = t[:, 0, 0] != 0
    t = t[used].astype(np.float64)
    nb = t.shape[0]
    t0 = t[:, :, 0].min()
    t = (t - t0) / 100.0  # us
    span = t[:, :, 5].max()
    print(f"\n=== K{K} N{N} splitk={sk}: {nb} blocks, kernel span {span:.2f} us (first wave start -> last wave end)")
    def q(a): return " ".join(f"{v:6.2f}" for v in np.percentile(a, [0, 10, 50, 90, 100]))
    print("  wave start      (p0 p10 p50 p90 p100):", q(t[:, :, 0]))
    print("  loads issued  dt                     :", q(t[:, :, 1] - t[:, :, 0]))
    print("  first step done dt (HBM latency+mma) :", q(t[:, :, 2] - t[:, :, 1]))
    print("  K loop rest   dt                     :", q(t[:, :, 3] - t[:, :, 2]))
    print("  LDS fold sync dt                     :", q(t[:, :, 4] - t[:, :, 3]))
    print("  publish/collect dt                   :", q(t[:, :, 5] - t[:, :, 4]))
    print("  wave end                             :", q(t[:, :, 5]))
    print("  wave lifetime                        :", q(t[:, :, 5] - t[:, :, 0]))
    bstart = np.sort(t[:, :, 0].min(axis=1))
    print("  block start times, every 1/16th:", " ".join(f"{v:.2f}" for v in bstart[:: max(1, nb // 16)]))
    bend = np.sort(t[:, :, 5].max(axis=1))
    print("  block end   times, every 1/16th:", " ".join(f"{v:.2f}" for v in bend[:: max(1, nb // 16)]))
    del sets
    torch.cuda.empty_cache()
