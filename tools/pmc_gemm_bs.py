#!/usr/bin/env python3
"""HBM traffic of the batched-decode legs of bench.py (`gemm_bs`: 4096 x 11008 g128, cold weights) from the TCC counters.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -o bs -- python tools/pmc_gemm_bs.py run
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <dir2> -o bs -- python tools/pmc_gemm_bs.py run     (own pass)
    python tools/pmc_gemm_bs.py summarize <dir> [<dir2>] [<probe dir>]

`run` issues, per (layout, M), one call on each of 28 distinct matrices behind a MARKER launch (awq_silu_and_mul with a grid size
that names the group), so that the summary can attribute every dispatch: FETCH_SIZE counts half of a wide streaming read on gfx950
(MI355X_MICROARCH.md, HBM section; the factor is re-measured on tools/stream_probe2 when its CSV is given), WRITE_SIZE is reported
as it is (KiB)."""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
K, N, NSETS = 4096, 11008, 28
GROUPS = [("gemm", 1), ("gemm", 8), ("gemm", 64), ("gemv", 1), ("gemv", 8), ("gemv", 64), ("gemv", 96), ("gemv", 128)]


def run():
    import torch
    import bench
    from autoawq_amd import ops

    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(5)
    mk_in = torch.zeros((256 * 16, 16), dtype=torch.float16, device=dev)
    for layout in ("gemm", "gemv"):
        sets = [bench.rand_packed(K, N, 128, dev, gen) if layout == "gemm" else bench.rand_packed_nk(K, N, 128, dev, gen) for _ in range(NSETS)]
        for gi, (lay, M) in enumerate(GROUPS):
            if lay != layout:
                continue
            x = torch.randn((M, K), device=dev, generator=gen).half()
            for rep in range(2):  # the first round warms nothing that matters (631 MB of weights >> 256 MiB Infinity Cache); both are counted
                ops.silu_and_mul(mk_in[: 256 * (gi + 1)])  # marker: grid = 256 x (gi + 1) threads
                for qw, qz, sc in sets:
                    if layout == "gemm":
                        ops.gemm_forward(x, qw, sc, qz)
                    else:
                        ops.gemv_forward(x, qw, sc, qz, 128, flags=int(os.environ.get("AWQ_PMC_GEMV_FLAGS", "0"), 0))
            torch.cuda.synchronize()
            print(f"{layout} M={M}: kernel {ops.last_kernel()}", flush=True)
        del sets
        torch.cuda.empty_cache()


def rows(d, counter):
    out = []
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                out.append(r)
    out.sort(key=lambda r: int(r["Dispatch_Id"]))
    return out


def per_group(d, counter):
    """{group index: (sum of the counter over the awq_ GEMM / GEMV dispatches, calls, kernel names)}"""
    acc = collections.defaultdict(lambda: [0.0, 0, set()])
    cur = None
    for r in rows(d, counter):
        name = r["Kernel_Name"]
        if "awq_silu_and_mul" in name:
            cur = int(r["Grid_Size"]) // 256 - 1
            acc[cur][1] += NSETS
            continue
        if cur is None or not name.startswith("awq_") and "awq_" not in name:
            continue
        if any(k in name for k in ("awq_gemv", "awq_gemm")):
            acc[cur][0] += float(r["Counter_Value"])
            acc[cur][2].add(name.replace("void ", "").replace("(anonymous namespace)::", "").split("<")[0].split("(")[0])
    return acc


def summarize(argv):
    sys.path.insert(0, ROOT)
    import bench

    fetch_dir = argv[0]
    write_dir = argv[1] if len(argv) > 1 and glob.glob(os.path.join(argv[1], "**", "*counter_collection.csv"), recursive=True) else None
    probe = argv[2] if len(argv) > 2 else None
    factor = 2.0
    if probe:
        vals = [float(r["Counter_Value"]) for r in rows(probe, "FETCH_SIZE") if "linear_read" in r["Kernel_Name"]]
        if vals:
            med = sorted(vals)[len(vals) // 2]
            print(f"# calibration (tools/stream_probe2 linear read): {med:.0f} KiB reported for 8192 KiB read -> factor {8192 / med:.3f}; the table uses 2")
    f = per_group(fetch_dir, "FETCH_SIZE")
    w = per_group(write_dir, "WRITE_SIZE") if write_dir else {}
    print("# per CALL (one Linear, all its launches): traffic = FETCH_SIZE x 2 x 1024 B; written = WRITE_SIZE x 1024 B; algorithmic = SURVEY 8(d)")
    for gi, (lay, M) in enumerate(GROUPS):
        if gi not in f or not f[gi][1]:
            continue
        t = f[gi][0] * factor * 1024 / f[gi][1]
        alg = bench.algorithmic_bytes(K, N, M, 128)
        wr = f", written {w[gi][0] * 1024 / w[gi][1] / 1e6:.3f} MB (y = {M * N * 2 / 1e6:.3f} MB)" if gi in w and w[gi][1] else ""
        print(f"{lay} M={M}: traffic {t / 1e6:.3f} MB  algorithmic {alg / 1e6:.3f} MB  ratio {t / alg:.3f}{wr}  kernels {sorted(f[gi][2])}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "run":
        run()
    elif len(sys.argv) > 2 and sys.argv[1] == "summarize":
        summarize(sys.argv[2:])
    else:
        sys.exit(__doc__)
