#!/usr/bin/env python3
"""Which part of a K step sets the pace of the register-decoded GEMM (csrc/gemm_regb.hip) when the grid does NOT fill the chip?
Times 4096 x 11008 and 11008 x 4096 at M = 128 / 512 / 2048 with parts of the kernel switched off (-DAWQ_REGB_EXPERIMENTS build of
tools/regb_experiments.py; results are wrong by design, timing only), 128-row and 256-row tiles.

    gpurun -- python tools/regb_mid_experiments.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import regb_experiments as rx

if __name__ == "__main__":
    rx.build()
    import torch
    from autoawq_amd import _lib
    _lib.LIB_PATH = rx.OUT
    from autoawq_amd import ops
    from bench import rand_packed

    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(0)
    for K, N in ((4096, 11008), (11008, 4096)):
        qw, qz, sc = rand_packed(K, N, 128, dev, gen)
        for M in (128, 512, 2048):
            x = torch.randn((M, K), device=dev, generator=gen).half()
            for nlog in (1, 2):
                fl = ops.gemm_flags(ops.KERNEL_REGB, nlog=nlog)
                line = []
                for dbg in (0, 1, 2, 4, 8, 15):
                    os.environ["AWQ_REGB_DBG"] = str(dbg)
                    for _ in range(3):
                        ops.gemm_forward(x, qw, sc, qz, flags=fl)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10):
                        ops.gemm_forward(x, qw, sc, qz, flags=fl)
                    e1.record()
                    e1.synchronize()
                    line.append(f"dbg{dbg}={e0.elapsed_time(e1) * 100:7.1f}")
                print(f"K={K} N={N} M={M:5d} bm={128 * nlog}: " + "  ".join(line) + "   (us; 0 kernel, 1 weights once, 2 activations once, 4 no barrier, 8 no decode, 15 all)", flush=True)
