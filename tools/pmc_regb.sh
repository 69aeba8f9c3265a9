#!/bin/bash
# SQ counters of the prefill GEMM kernels (BASELINE config 3: 4096 x 11008, M = 16384): MFMA busy, wait classes, LDS
# conflicts.  Two --pmc passes (kernel-trace only, as gpurun requires).  Output: gpurun_out/pmc_regb/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_regb
cat > /tmp/one_gemm.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from autoawq_amd import ops
from bench import rand_packed
dev = torch.device("cuda"); gen = torch.Generator(device=dev).manual_seed(0)
K, N, M = 4096, 11008, 16384
qw, qz, sc = rand_packed(K, N, 128, dev, gen)
x = torch.randn((M, K), device=dev, generator=gen).half()
for kern, nlog in ((ops.KERNEL_REGB, 2), (ops.KERNEL_REGB, 1), (ops.KERNEL_TILED, 2)):
    for _ in range(3):
        ops.gemm_forward(x, qw, sc, qz, flags=ops.gemm_flags(kern, nlog=nlog))
W = ops.dequantize_weights(qw, sc, qz)
for _ in range(3):
    torch.matmul(x, W)
torch.cuda.synchronize()
PY
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_regb -o p1 -- python /tmp/one_gemm.py > gpurun_out/pmc_regb_1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_regb -o p2 -- python /tmp/one_gemm.py > gpurun_out/pmc_regb_2.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_regb/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        if "regb" in k or "tiled" in k or "Cijk" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", f)
    for k, d in agg.items():
        print(k, {c: f"{sum(v) / len(v):.4g}" for c, v in d.items()})
PY
