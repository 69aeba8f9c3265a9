#!/usr/bin/env python3
"""What bounds the register-decoded prefill GEMM (csrc/gemm_regb.hip)?  Times the 4096 x 11008, M = 16384 call with
parts of the kernel switched off (a separate -DAWQ_REGB_EXPERIMENTS build in tools/bin/, results are wrong by design).

    python tools/regb_experiments.py --build-only     # here (no GPU)
    gpurun -- python tools/regb_experiments.py
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "autoawq_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "bin", "libawq_hip_regbx.so")


def build():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    newest = max(os.path.getmtime(f) for f in srcs + [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")])
    if os.path.exists(OUT) and os.path.getmtime(OUT) >= newest:
        return
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
                           "-fno-slp-vectorize", "-Wno-inline-asm", "-DAWQ_REGB_EXPERIMENTS", "-DAWQ_BUILDING_LIB", "-I" + os.path.join(ROOT, "include")] + srcs + ["-o", OUT])


if __name__ == "__main__":
    build()
    if "--build-only" in sys.argv:
        sys.exit(0)
    import torch
    from autoawq_amd import _lib
    _lib.LIB_PATH = OUT
    from autoawq_amd import ops
    from bench import rand_packed

    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(0)
    K, N, M = 4096, 11008, 16384
    qw, qz, sc = rand_packed(K, N, 128, dev, gen)
    x = torch.randn((M, K), device=dev, generator=gen).half()
    fl = ops.gemm_flags(ops.KERNEL_REGB, nlog=2)
    fl1 = ops.gemm_flags(ops.KERNEL_REGB, nlog=1)
    for pm in (1, 2, 4, 8, 16, 32):  # shape of the tile patch an XCD runs at a time (pm M-tiles x 32/pm N-tiles), 128-row tiles
        os.environ["AWQ_REGB_PM"] = str(pm)
        for M2 in (8192, 16384):
            xs = x[:M2]
            for _ in range(3):
                ops.gemm_forward(xs, qw, sc, qz, flags=fl1)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.gemm_forward(xs, qw, sc, qz, flags=fl1)
            e1.record()
            e1.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 5
            print(f"patch {pm:2d} x {32 // pm:2d}  M={M2:5d}  {us:8.1f} us  {2.0 * M2 * K * N / us / 1e6:7.1f} TF", flush=True)
    del os.environ["AWQ_REGB_PM"]
    for dbg, what in ((0, "the kernel"), (1, "weights fetched once"), (2, "activations fetched once"), (4, "no barrier"),
                      (8, "no decode arithmetic"), (16, "output stores kept in L2"), (15, "1+2+4+8"), (0, "the kernel")):
        os.environ["AWQ_REGB_DBG"] = str(dbg)
        for _ in range(2):
            ops.gemm_forward(x, qw, sc, qz, flags=fl)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.gemm_forward(x, qw, sc, qz, flags=fl)
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 5
        print(f"dbg={dbg:2d} {what:28s} {us:8.1f} us  {2.0 * M * K * N / us / 1e6:7.1f} TF", flush=True)
