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
by_grid = collections.defaultdict(list)
for r in rows(sys.argv[1]):
    if KERNEL in r["Kernel_Name"]:
        targs = r["Kernel_Name"].split("<")[1].split(">")[0] if "<" in r["Kernel_Name"] else ""
        by_grid[(int(r["Grid_Size"]), int(r["Workgroup_Size"]), targs)].append(float(r["Counter_Value"]))
print(f"\nbench.py decode kernels ({KERNEL}), traffic = FETCH_SIZE x {factor:.0f} x 1024 B; by grid size (threads):")
tot_t = tot_a = n = 0
for g, v in sorted(by_grid.items()):
    med = sorted(v)[len(v) // 2] * factor * 1024
    # which shape: blocks = tiles * S; match by algorithmic bytes closest to the traffic
    name, (K, N) = min(shapes.items(), key=lambda kv: abs(algorithmic_bytes(kv[1][0], kv[1][1], 1, GROUP) - med))
    alg = algorithmic_bytes(K, N, 1, GROUP)
    print(f"  grid {g[0]:7d} wg {g[1]:3d} <{g[2]}>: traffic {med / 1e6:8.3f} MB  ~ {name:22s} algorithmic {alg / 1e6:8.3f} MB  ratio {med / alg:5.3f}  (n={len(v)})")
    tot_t += med * len(v); tot_a += alg * len(v); n += len(v)
if n:
    print(f"\nper launch (weighted mean): traffic {tot_t / n / 1e6:.3f} MB, algorithmic {tot_a / n / 1e6:.3f} MB, ratio {tot_t / tot_a:.3f}")
