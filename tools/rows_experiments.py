#!/usr/bin/env python3
"""Where does a launch of the row-streaming GEMV kernel (csrc/gemv_rows.hip) spend its time?  Times the four 7B shapes
at M = 1 with parts of the kernel switched off (separate -DAWQ_ROWS_DBG=bits builds in tools/bin/, results wrong by design).

    python tools/rows_experiments.py --build-only     # here (no GPU)
    gpurun -- python tools/rows_experiments.py
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "autoawq_amd", "csrc")
VARIANTS = [(0, "the kernel"), (64, "dot products on the VALU (v_dot2c) instead of MFMA 4x4x4"), (1, "no decode / dot products"),
            (2, "no x DMA / barrier"), (4, "no scale / transpose-reduce"), (8, "no final fold"), (16, "no scale / zero loads"),
            (32, "x requested after the first weights"), (31, "1+2+4+8+16 (weight requests + waits + LDS partials)"),
            (128, "scales / zeros requested behind the first round instead of ahead of the ring"),
            (256, "scales / zeros by LDS-DMA even when they fit two registers per lane")]


def so(bits):
    return os.path.join(ROOT, "tools", "bin", f"libawq_hip_rowsx{bits}.so")


def build():
    os.makedirs(os.path.join(ROOT, "tools", "bin"), exist_ok=True)
    srcs = [os.path.join(CSRC, f) for f in ("gemv_rows.hip",)]
    others = [os.path.join(CSRC, "build", f[:-4] + ".o") for f in sorted(os.listdir(CSRC)) if f.endswith(".hip") and f != "gemv_rows.hip"]
    for bits, _ in VARIANTS:
        if bits == 0:
            continue
        obj = os.path.join(ROOT, "tools", "bin", f"gemv_rows_x{bits}.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
                               "-fno-slp-vectorize", "-Wno-inline-asm", f"-DAWQ_ROWS_DBG={bits}", "-DAWQ_BUILDING_LIB",
                               "-I" + os.path.join(ROOT, "include"), "-c"] + srcs + ["-o", obj])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so(bits), obj] + others)
        os.remove(obj)


def child(bits):
    import torch
    from autoawq_amd import _lib
    if bits:
        _lib.LIB_PATH = so(bits)
    from autoawq_amd import ops
    from tools.sweep_gemv_rows import rand_nk, graph_us, ROWS
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(0)
    res = []
    for K, N, fl in [(4096, 4096, {}), (4096, 12288, {}), (4096, 22016, {}), (11008, 4096, {})]:  # the launcher's defaults
        nsets = max(4, min(96, (640 << 20) // (K * N // 2)))
        sets = [rand_nk(K, N, 128) for _ in range(nsets)]
        x = torch.randn((1, K), device=dev, generator=gen).half()
        f = ops.gemm_flags(kernel=ROWS, **fl)

        def run():
            for qw, qz, sc in sets:
                ops.gemv_forward(x, qw, sc, qz, 128, flags=f)
        res.append(graph_us(run, nsets))
        del sets
        torch.cuda.empty_cache()
    print(f"dbg={bits:2d} {dict(VARIANTS)[bits]:58s} " + "  ".join(f"{u:6.2f}" for u in res) + "   us  (o, qkv, gate_up, down)", flush=True)


if __name__ == "__main__":
    if "--build-only" in sys.argv:
        build()
    elif "--child" in sys.argv:
        child(int(sys.argv[sys.argv.index("--child") + 1]))
    else:
        for bits, _ in VARIANTS:
            subprocess.call([sys.executable, os.path.abspath(__file__), "--child", str(bits)])
