"""EXPERIMENTAL (round 4, never run on a GPU yet): builds csrc/gemm_skinny.hip + gemm_skinny_nk.patch into tools/bin/libskinny_nk.so and checks it on
an MI355X against the product's own bit-exact dequantisation of the same GEMV-layout buffers + a dense fp32 matmul, then times
it beside what WQLinear_GEMV does today for 17 ... 64 rows (16-row chunks of the decode kernels).

    gpurun --timeout 600 -- 'python tools/experimental/gemm_skinny_nk/probe.py > gpurun_out/skinny_nk.txt 2>&1'

Exit status 0 = every shape within tolerance.  Nothing under autoawq_amd/ imports this file or the library it builds."""
import ctypes
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "bin", "libskinny_nk.so")


def build():
    """csrc/gemm_skinny.hip + gemm_skinny_nk.patch -> tools/bin/gemm_skinny_nk.hip -> tools/bin/libskinny_nk.so (the experimental
    kernel is kept as a patch against the product kernel it extends, not as a second copy of it).  tools/bin/ is git-ignored and
    travels with a gpurun snapshot: build HERE first (`--build-only`); a stamp file holds the hash of the inputs (file times do not
    survive the copy)."""
    import hashlib

    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    base = os.path.join(ROOT, "autoawq_amd", "csrc", "gemm_skinny.hip")
    patch = os.path.join(HERE, "gemm_skinny_nk.patch")
    src = os.path.join(os.path.dirname(OUT), "gemm_skinny_nk.hip")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-fno-slp-vectorize", "-Wno-unused-function",
             "-Wno-inline-asm", "-DAWQ_BUILDING_LIB"]
    stamp = hashlib.sha1(open(base, "rb").read() + open(patch, "rb").read() + " ".join(flags).encode()).hexdigest()
    if os.path.exists(OUT) and os.path.exists(OUT + ".stamp") and open(OUT + ".stamp").read() == stamp:
        return
    subprocess.check_call(["patch", "--quiet", "-o", src, base, patch])
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "autoawq_amd", "csrc"),
                           "-shared", "-o", OUT, src])
    open(OUT + ".stamp", "w").write(stamp)


def random_gemv_layer(K, N, g, dev, seed):
    gen = torch.Generator(device="cpu").manual_seed(seed)
    zw = -(-(K // g) // 8)
    qweight = torch.randint(-2**31, 2**31 - 1, (N, K // 8), dtype=torch.int64, generator=gen).to(torch.int32)
    qzeros = torch.randint(-2**31, 2**31 - 1, (N, zw), dtype=torch.int64, generator=gen).to(torch.int32)
    scales = (torch.rand((N, 8 * zw), generator=gen) * 0.02 + 0.005).to(torch.float16)
    return qweight.to(dev), qzeros.to(dev), scales.to(dev)


def main():
    if "--build-only" in sys.argv:
        build()
        print("built", OUT)
        return 0
    build()
    from autoawq_amd import ops

    lib = ctypes.CDLL(OUT)
    fn = lib.awq_exp_gemm_skinny
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 7 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    bad = 0

    def run(x, qw, sc, qz, g, splitk=0, mode=1):
        M, K = x.shape
        N = qw.shape[0]
        y = torch.full((M, N), float("nan"), dtype=torch.float16, device=dev)
        ws = ops.workspace(dev, 64 << 20)
        rc = fn(x.data_ptr(), qw.data_ptr(), sc.data_ptr(), qz.data_ptr(), y.data_ptr(), M, K, N, g, qz.shape[1], mode, splitk,
                ws.data_ptr(), ws.numel(), stream)
        return rc, y

    shapes = [(4096, 4096, 128), (4096, 11008, 128), (11008, 4096, 128), (4096, 12288, 128), (4096, 4096, 64), (1024, 264, 128),
              (8192, 1024, 128), (4096, 4104, 128), (14336, 4096, 128)]
    for K, N, g in shapes:
        qw, qz, sc = random_gemv_layer(K, N, g, dev, seed=K + N + g)
        wt = ops.dequantize_weights_gemv(qw, sc, qz, g).float()  # [N, K], bit-exact vs the oracle (tests/test_gpu_parity.py)
        for M, splitk, mode in [(M, sk, md) for M in (1, 5, 8, 9, 16, 17, 31, 32, 33, 48, 64) for sk in (0, 1) for md in (1, 2)]:
            if True:
                x = (torch.randn((M, K), generator=torch.Generator().manual_seed(M), dtype=torch.float32) * 0.5).to(torch.float16).to(dev)
                rc, y = run(x, qw, sc, qz, g, splitk, mode)
                if rc != 0:
                    print(f"K={K} N={N} g={g} M={M} splitk={splitk} mode={mode}: rc={rc} (declined)")
                    continue
                ref = x.float() @ wt.t()
                # fp32 accumulation of exact fp16 products, one rounding to fp16 at the end: half an ulp of the result + slack
                err = (y.float() - ref).abs()
                tol = ref.abs() * 2.0**-10 + 2e-2
                ok = bool((err <= tol).all()) and bool(torch.isfinite(y).all())
                bad += not ok
                print(f"K={K} N={N} g={g} M={M} splitk={splitk} mode={mode}: max err {float(err.max()):.4g} "
                      f"(max |ref| {float(ref.abs().max()):.4g}) {'ok' if ok else 'MISMATCH'}")
                # run-to-run: the combine order is fixed, two launches must agree bit for bit
                rc2, y2 = run(x, qw, sc, qz, g, splitk, mode)
                if rc2 == 0 and not torch.equal(y, y2):
                    bad += 1
                    print("   NOT reproducible run to run")
    ops.check_workspaces()

    # timing inside a hipGraph (bench.graph_time), the shape of bench.py's gemm_bs leg; cold weights: 16 distinct matrices per replay
    import bench
    K, N, g = 4096, 11008, 128
    mats = [random_gemv_layer(K, N, g, dev, seed=100 + i) for i in range(16)]
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        ws = ops.workspace(dev, 64 << 20)
    for M in (8, 16, 17, 32, 64):  # (M <= 16: the experimental one-tile instantiation; today = gemv_lds / gemv_nk)
        x = torch.randn((M, K), dtype=torch.float16, device=dev)
        y = torch.empty((M, N), dtype=torch.float16, device=dev)

        def exp(mode):
            def f():
                cur = torch.cuda.current_stream().cuda_stream
                for qw, qz, sc in mats:
                    rc = fn(x.data_ptr(), qw.data_ptr(), sc.data_ptr(), qz.data_ptr(), y.data_ptr(), M, K, N, g, qz.shape[1], mode, 0,
                            ws.data_ptr(), ws.numel(), cur)
                    assert rc == 0, rc
            return f

        def today():
            for qw, qz, sc in mats:
                ops.gemv_forward(x, qw, sc, qz, g)

        for name, f in (("skinny_nk (64-wide steps)", exp(1)), ("skinny_nk wide (256-wide steps)", exp(2)), ("today (gemv_forward)", today)):
            us = bench.graph_time(f, st, reps=20, min_seconds=0.2) / len(mats)
            print(f"M={M} {name}: {us:.1f} us per matrix ({bench.algorithmic_bytes(K, N, M, g) / us / 1e3:.0f} GB/s algorithmic)")
    print("FAILED" if bad else "ALL OK")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
