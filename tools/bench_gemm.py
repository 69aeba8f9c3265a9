#!/usr/bin/env python3
"""Prefill / batched GEMM timing: fused tiled MFMA kernel vs the two-pass route (HIP dequant +
vendor fp16 GEMM, what awq/modules/linear/gemm.py:48-54 does) on the BASELINE shape 4096x11008."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from autoawq_amd import ops
from bench import algorithmic_bytes, rand_packed

dev = torch.device("cuda")
gen = torch.Generator(device=dev).manual_seed(0)


def timeit(fn, reps):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for K, N in [(4096, 11008), (11008, 4096)]:
    qw, qz, sc = rand_packed(K, N, 128, dev, gen)
    for M in ([int(v) for v in os.environ['MS'].split(',')] if os.environ.get('MS') else [1024, 4096, 16384]):
        x = torch.randn((M, K), device=dev, generator=gen).half()
        reps = 20 if M <= 1024 else 5
        fl = 2.0 * M * K * N
        row = f"K{K} N{N} M{M:6d}:"
        for nm, f in [("tiled128", lambda: ops.gemm_forward(x, qw, sc, qz, flags=ops.gemm_flags(ops.KERNEL_TILED, nlog=1))),
                      ("tiled256", lambda: ops.gemm_forward(x, qw, sc, qz, flags=ops.gemm_flags(ops.KERNEL_TILED, nlog=2))),
                      ("regb128", lambda: ops.gemm_forward(x, qw, sc, qz, flags=ops.gemm_flags(ops.KERNEL_REGB, nlog=1))),
                      ("regb256", lambda: ops.gemm_forward(x, qw, sc, qz, flags=ops.gemm_flags(ops.KERNEL_REGB, nlog=2))),
                      ("2pass", lambda: torch.matmul(x, ops.dequantize_weights(qw, sc, qz)))]:
            us = timeit(f, reps)
            row += f"  {nm} {us:9.1f} us {fl / us / 1e6:7.1f} TF"
        W = ops.dequantize_weights(qw, sc, qz)
        us = timeit(lambda: torch.matmul(x, W), reps)
        row += f"  matmul-only {us:9.1f} us {fl / us / 1e6:7.1f} TF"
        a = ops.gemm_forward(x, qw, sc, qz, flags=ops.gemm_flags(ops.KERNEL_TILED, nlog=1)).float()
        b = torch.matmul(x, W).float()
        row += f"  maxrel {float((a - b).abs().max() / b.abs().max()):.1e}"
        for nl in (1, 2):  # the register-decoded kernel multiplies exactly the same fp16 weights
            a = ops.gemm_forward(x, qw, sc, qz, flags=ops.gemm_flags(ops.KERNEL_REGB, nlog=nl)).float()
            row += f"  regb{128 * nl} maxrel {float((a - b).abs().max() / b.abs().max()):.1e}"
        print(row, flush=True)
