#!/usr/bin/env python3
"""Decode attention: how many sequence splits?  Builds csrc/decoder.hip with other (blocks in flight, rows per split beyond the
single-split length of 256) targets into tools/bin/ and times the whole 7B-shape decoder (tools/bench_decode_model.py, GEMV layout) with each library.
    python tools/attn_split_ab.py --build-only   (here)      gpurun -- python tools/attn_split_ab.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "autoawq_amd", "csrc")
VARIANTS = [(1024, 128, 0), (1024, 128, 2), (1024, 128, 0), (1024, 128, 2)]  # VARIANTS[0] = csrc/decoder.hip as built; third field: bit 0 = AWQ_ATTN_NO_EARLY, bit 1 = AWQ_ATTN_NO_DPP


def so(v):
    return os.path.join(ROOT, "tools", "bin", f"libawq_hip_attn_{v[0]}_{v[1]}_{v[2]}.so")


def build():
    os.makedirs(os.path.join(ROOT, "tools", "bin"), exist_ok=True)
    others = [os.path.join(CSRC, "build", f[:-4] + ".o") for f in sorted(os.listdir(CSRC)) if f.endswith(".hip") and f != "decoder.hip"]
    for v in sorted(set(VARIANTS[1:]) - {VARIANTS[0]}):
        obj = os.path.join(ROOT, "tools", "bin", "decoder_x.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
                               "-fno-slp-vectorize", "-Wno-inline-asm", f"-DAWQ_ATTN_BLOCKS={v[0]}", f"-DAWQ_ATTN_ROWS={v[1]}", f"-DAWQ_ATTN_NO_EARLY={v[2] & 1}", f"-DAWQ_ATTN_NO_DPP={v[2] >> 1}",
                               "-DAWQ_BUILDING_LIB", "-I" + os.path.join(ROOT, "include"), "-c", os.path.join(CSRC, "decoder.hip"), "-o", obj])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so(v), obj] + others)
        os.remove(obj)


def child(i):
    from autoawq_amd import _lib
    if VARIANTS[i] != VARIANTS[0]:
        _lib.LIB_PATH = so(VARIANTS[i])
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_decode_model
    r = bench_decode_model.run(contexts=(64, 200, 512, 2048), steps=48, verbose=False, check=False, layout="gemv")
    print(f"blocks <= {VARIANTS[i][0]:5d}  rows per split {VARIANTS[i][1]:4d}  early K/V request {'no ' if VARIANTS[i][2] & 1 else 'yes'}  DPP row sums {'no ' if VARIANTS[i][2] & 2 else 'yes'}: " + "  ".join(f"ctx {c}: {1000.0 / ms:6.1f} tok/s" for c, ms in r.items()), flush=True)


if __name__ == "__main__":
    if "--build-only" in sys.argv:
        build()
    elif "--child" in sys.argv:
        child(int(sys.argv[sys.argv.index("--child") + 1]))
    else:
        for i in range(len(VARIANTS)):
            subprocess.call([sys.executable, os.path.abspath(__file__), "--child", str(i)])
