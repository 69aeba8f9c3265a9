#!/usr/bin/env python3
"""Builds autoawq_amd/csrc/libawq_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python autoawq_amd/csrc/build.py [--force] [--verbose] [--save-temps]

One object per .hip translation unit (parallel, cached by mtime), then one shared link.
"""
import argparse
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "libawq_hip.so")
OBJ = os.path.join(HERE, "build")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
         "-fvisibility-inlines-hidden", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function", "-Wno-inline-asm",
         "-DAWQ_BUILDING_LIB", "-I" + os.path.join(ROOT, "include")]


# per-file extras: the prefill attention takes its running maxima with plain v_max_f32 (-fno-honor-nans drops the canonicalising
# v_max(x, x) hipcc puts in front of every fmaxf: +3 .. 5 %, profiles/r05_first_call/prefill_attn.txt); masked scores are -inf, never NaN
EXTRA = {"prefill_attn.hip": ["-fno-honor-nans"]}


def sources():
    return sorted(f for f in os.listdir(HERE) if f.endswith(".hip"))


def headers_mtime():
    hs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")]
    hs.append(os.path.join(ROOT, "include", "awq_hip.h"))
    hs.append(os.path.abspath(__file__))
    return max(os.path.getmtime(h) for h in hs)


def compile_one(src, force, verbose, save_temps):
    obj = os.path.join(OBJ, src[:-4] + ".o")
    sp = os.path.join(HERE, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(sp), headers_mtime()):
        return obj, False
    cmd = [HIPCC] + FLAGS + EXTRA.get(src, []) + ["-c", sp, "-o", obj]
    if save_temps:
        cmd += ["-save-temps=obj"]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=OBJ)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed on " + src)
    if r.stderr.strip() and verbose:
        sys.stderr.write(r.stderr)
    return obj, True


def build(force=False, verbose=False, save_temps=False):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sources()
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(lambda s: compile_one(s, force, verbose, save_temps), srcs))
    objs = [o for o, _ in res]
    manifest = os.path.join(OBJ, "linked.txt")  # relink when a source was added or removed, not only when one changed
    linked = open(manifest).read().split() if os.path.exists(manifest) else []
    for stale in set(os.listdir(OBJ)) - {os.path.basename(o) for o in objs} - {"linked.txt"}:
        if stale.endswith(".o"):
            os.remove(os.path.join(OBJ, stale))
    if any(c for _, c in res) or not os.path.exists(OUT) or force or linked != [os.path.basename(o) for o in objs]:
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(manifest, "w") as f:
            f.write("\n".join(os.path.basename(o) for o in objs))
    return OUT


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--save-temps", action="store_true")
    a = ap.parse_args()
    print(build(a.force, a.verbose, a.save_temps))
