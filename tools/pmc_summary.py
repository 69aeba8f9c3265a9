#!/usr/bin/env python3
"""HBM traffic of bench.py's decode kernels from a rocprofv3 --pmc FETCH_SIZE pass (CSV output).

    tools/pmc_summary.py <dir with *counter_collection.csv> [probe dir]

FETCH_SIZE is reported in KiB and counts HALF of a wide coalesced read on gfx950 (MI355X_MICROARCH.md, HBM section);
the factor is re-measured on the linear-read probe (tools/stream_probe2) when its CSV is given."""
import collections, csv, glob, os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import GROUP, HIDDEN, INTER, algorithmic_bytes


def rows(d):
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "FETCH_SIZE":
                yield r


KERNEL = os.environ.get("KERNEL", "awq_gemv_rows_kernel")  # the headline's decode kernel (round 2: awq_gemv_mfma_kernel)
factor = 2.0
if len(sys.argv) > 2:
    vals = collections.defaultdict(list)
    for r in rows(sys.argv[2]):
        vals[(r["Kernel_Name"][:40], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    print("calibration (linear-read probe): FETCH_SIZE KiB per dispatch by (kernel, grid):")
    for k, v in vals.items():
        if "linear_read" in k[0]:  # reads 8192 KiB of a 4096 x 4096 int4 buffer (tools/stream_probe2.hip)
            print(f"  {k}: {sorted(v)[len(v) // 2]:.0f} KiB reported for 8192 KiB read -> factor {8192 / sorted(v)[len(v) // 2]:.3f} (n={len(v)})")
shapes = {"qkv 4096->12288": (HIDDEN, 3 * HIDDEN), "o 4096->4096": (HIDDEN, HIDDEN), "gate+up 4096->22016": (HIDDEN, 2 * INTER),
          "down 11008->4096": (INTER, HIDDEN)}
by_shape = collections.defaultdict(list)  # every dispatch is matched to the shape whose algorithmic bytes are closest to its traffic
for r in rows(sys.argv[1]):                # (the row-streaming kernel launches o and qkv with the same grid and template arguments)
    if KERNEL in r["Kernel_Name"]:
        targs = r["Kernel_Name"].split("<")[1].split(">")[0] if "<" in r["Kernel_Name"] else ""
        t = float(r["Counter_Value"]) * factor * 1024
        name = min(shapes, key=lambda k: abs(algorithmic_bytes(shapes[k][0], shapes[k][1], 1, GROUP) - t))
        by_shape[(name, int(r["Grid_Size"]), int(r["Workgroup_Size"]), targs)].append(t)
print(f"\nbench.py decode kernels ({KERNEL}), traffic = FETCH_SIZE x {factor:.0f} x 1024 B; by shape (grid size in threads):")
tot_t = tot_a = n = 0
for g, v in sorted(by_shape.items()):
    med = sorted(v)[len(v) // 2]
    K, N = shapes[g[0]]
    alg = algorithmic_bytes(K, N, 1, GROUP)
    print(f"  {g[0]:22s} grid {g[1]:7d} wg {g[2]:3d} <{g[3]}>: traffic {med / 1e6:8.3f} MB  algorithmic {alg / 1e6:8.3f} MB  ratio {med / alg:5.3f}  (n={len(v)})")
    tot_t += med * len(v); tot_a += alg * len(v); n += len(v)
if n:
    print(f"\nper launch (weighted mean): traffic {tot_t / n / 1e6:.3f} MB, algorithmic {tot_a / n / 1e6:.3f} MB, ratio {tot_t / tot_a:.3f}")
