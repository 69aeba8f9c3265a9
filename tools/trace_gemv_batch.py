#!/usr/bin/env python3
"""csrc/gemv_batch.hip: per-wave phase timeline (-DAWQ_GEMV_TRACE build) and timing with parts switched off (-DAWQ_BT_DBG=bits builds;
results wrong by design).  Libraries are built HERE into tools/bin/ (`--build-only`; they travel with the gpurun snapshot).
    gpurun -- 'python tools/trace_gemv_batch.py > gpurun_out/trace_gemv_batch.txt 2>&1'"""
import ctypes
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "autoawq_amd", "csrc")
BIN = os.path.join(ROOT, "tools", "bin")
VARIANTS = {"trace": ["-DAWQ_GEMV_TRACE"], "trace6": ["-DAWQ_GEMV_TRACE", "-DAWQ_BT_DBG=6"], "trace14": ["-DAWQ_GEMV_TRACE", "-DAWQ_BT_DBG=14"], "trace4": ["-DAWQ_GEMV_TRACE", "-DAWQ_BT_DBG=4"], "dbg1": ["-DAWQ_BT_DBG=1"], "dbg2": ["-DAWQ_BT_DBG=2"], "dbg4": ["-DAWQ_BT_DBG=4"], "dbg7": ["-DAWQ_BT_DBG=7"],
            "dbg3": ["-DAWQ_BT_DBG=3"], "dbg14": ["-DAWQ_BT_DBG=14"], "dbg20": ["-DAWQ_BT_DBG=20"], "dbg36": ["-DAWQ_BT_DBG=36"], "dbg6": ["-DAWQ_BT_DBG=6"]}


def lib_path(name):
    return os.path.join(BIN, f"libawq_hip_batch_{name}.so")


def build():
    """the product library with gemv_batch.hip recompiled under the variant's define (the other objects are reused)"""
    os.makedirs(BIN, exist_ok=True)
    from autoawq_amd.csrc import build as hip_build

    hip_build.build()
    objs = [os.path.join(hip_build.OBJ, f) for f in sorted(os.listdir(hip_build.OBJ)) if f.endswith(".o") and f != "gemv_batch.o"]
    src = os.path.join(CSRC, "gemv_batch.hip")
    stamp_src = hashlib.sha1(open(src, "rb").read() + b"".join(open(o, "rb").read() for o in objs)).hexdigest()
    for name, defs in VARIANTS.items():
        out = lib_path(name)
        stamp = stamp_src + " ".join(defs)
        if os.path.exists(out) and os.path.exists(out + ".stamp") and open(out + ".stamp").read() == stamp:
            continue
        obj = os.path.join(BIN, f"gemv_batch_{name}.o")
        subprocess.check_call([hip_build.HIPCC] + hip_build.FLAGS + defs + ["-c", src, "-o", obj])
        subprocess.check_call([hip_build.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + [obj])
        open(out + ".stamp", "w").write(stamp)


def child(name):
    """one process per library (the ctypes handle of autoawq_amd._lib is process-wide)"""
    import numpy as np
    import torch

    from autoawq_amd import _lib

    _lib.LIB_PATH = lib_path(name)
    import bench
    from autoawq_amd import ops

    L = _lib.lib()
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(3)
    st = torch.cuda.Stream(device=dev)
    BATCH = 5
    cases = [(4096, 11008, 8, 1, 2), (4096, 11008, 32, 1, 2), (4096, 11008, 64, 1, 2), (4096, 11008, 128, 1, 2), (11008, 4096, 64, 1, 2)]  # (round 6: 64 / 128 rows = two / four row parts)
    for K, N, M, gw, rd in cases:  # gw: the activations' form (1 = LDS staging area, 2 = direct fragment loads)
        nsets = 28
        mats = [bench.rand_packed_nk(K, N, 128, dev, gen) for _ in range(nsets)]
        x = torch.randn((M, K), device=dev, generator=gen).half()
        fl = ops.gemm_flags(kernel=BATCH, unit=gw, splitk=rd)

        def f():
            for qw, qz, sc in mats:
                ops.gemv_forward(x, qw, sc, qz, 128, flags=fl)

        us = bench.graph_time(f, st, reps=10, min_seconds=0.1) / len(mats)
        print(f"[{name}] K={K} N={N} M={M} form={'xs' if gw == 1 else 'direct'} rd={rd}: {us:.2f} us", flush=True)
        if name.startswith("trace"):
            L.awq_debug_set_trace_batch.argtypes = [ctypes.c_void_p]
            trace = torch.zeros(256 * 8 * 16, dtype=torch.int64, device=dev)
            for qw, qz, sc in mats[:-1]:
                ops.gemv_forward(x, qw, sc, qz, 128, flags=fl)
            torch.cuda.synchronize()
            L.awq_debug_set_trace_batch(trace.data_ptr())
            qw, qz, sc = mats[-1]
            ops.gemv_forward(x, qw, sc, qz, 128, flags=fl)
            torch.cuda.synchronize()
            L.awq_debug_set_trace_batch(None)
            t = trace.cpu().numpy().reshape(-1, 16).astype(np.float64)
            t = t[t[:, 0] != 0]
            phases = t[:, 12:] / 100.0  # accumulated us per wave over its units
            t = t[:, :12]
            t0 = t[:, 0].min()
            t = np.where(t > 0, (t - t0) / 100.0, np.nan)  # wall_clock64: 100 MHz

            def q(a):
                a = a[~np.isnan(a)]
                return " ".join(f"{v:6.2f}" for v in np.percentile(a, [0, 10, 50, 90, 100])) + f"   n={a.size}" if a.size else "(none)"
            names = ["wave start", "A + ring requested", "-", "A in registers", "piece 0 landed", "piece 0 consumed + next requested",
                     "piece 1 landed", "piece 1 consumed", "piece 2 landed", "piece 2 consumed", "stream done", "end (y stored)"]
            print(f"   {t.shape[0]} waves, absolute us since the first wave started (p0 p10 p50 p90 p100):")
            for i, nm in enumerate(names):
                print(f"   {nm:36s}: {q(t[:, i])}")
            print("   per wave, summed over its units (us):")
            for i, nm in enumerate(["waiting for the piece", "requesting", "consuming (LDS reads, decode, MFMA)", "partial-tile exchange (barrier)"]):
                print(f"   {nm:36s}: {q(phases[:, i])}")
        del mats
        torch.cuda.empty_cache()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
        sys.exit(0)
    build()
    if "--build-only" in sys.argv:
        print("built", ", ".join(VARIANTS))
        sys.exit(0)
    only = [a.split("=", 1)[1].split(",") for a in sys.argv if a.startswith("--only=")]
    for name in (only[0] if only else VARIANTS):
        subprocess.call([sys.executable, os.path.abspath(__file__), "--child", name])
