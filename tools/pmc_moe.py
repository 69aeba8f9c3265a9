#!/usr/bin/env python3
"""HBM traffic of the MoE decode launches from a rocprofv3 --pmc FETCH_SIZE pass of tools/bench_moe.py (CSV output):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR -o moe -- python tools/bench_moe.py
    python tools/pmc_moe.py DIR

FETCH_SIZE is reported in KiB and counts half of a wide coalesced read on gfx950 (factor 2, calibrated on the linear-read probe:
profiles/r06_pmc_fetch_size.txt).  Algorithmic bytes: the DISTINCT experts hit x (packed weights + scales + zeros) per projection
(bench_moe.py's routing at seed 0 hits 5 experts with its 8 pairs)."""
import collections, csv, glob, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import algorithmic_bytes

H, I, g = 4096, 14336, 128
acc = collections.defaultdict(list)
for fn in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        if r.get("Counter_Name") == "FETCH_SIZE" and ("awq_gemv_rows_kernel" in r["Kernel_Name"] or "awq_gemv_mfma_kernel" in r["Kernel_Name"]):
            targs = r["Kernel_Name"].split("<")[1].split(">")[0]
            acc[(r["Kernel_Name"].split("<")[0].split("::")[-1], targs, int(r["Grid_Size"]))].append(float(r["Counter_Value"]) * 2 * 1024)
hit = int(os.environ.get("EXPERTS_HIT", "5"))
w1, w2 = hit * algorithmic_bytes(H, 2 * I, 1, g), hit * algorithmic_bytes(I, H, 1, g)
print(f"# Mixtral shape, bs = 4, top-2, {hit} distinct experts hit by 8 pairs: algorithmic bytes w1|w3 {w1 / 1e6:.1f} MB, w2 {w2 / 1e6:.1f} MB per MoE block")
for (k, targs, grid), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if len(v) < 8:
        continue  # the correctness pre-checks of the tool (single-pair calls)
    med = sorted(v)[len(v) // 2]
    near = min((w1, "w1|w3"), (w2, "w2"), key=lambda a: abs(a[0] - med))
    print(f"  {k}<{targs}> grid {grid:8d}: traffic {med / 1e6:8.1f} MB per launch = {med / near[0]:5.3f} x the distinct experts' bytes of {near[1]}  (n={len(v)})")
