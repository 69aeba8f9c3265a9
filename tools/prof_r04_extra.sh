#!/bin/bash
# round-4 extra evidence (GPU box): (1) rocprofv3 --kernel-trace --stats of the whole-model decode, one run per context length;
# (2) SQ counters (MFMA busy / GRBM active) of the prefill kernels at config 3: gemm_regb, its N-major form, dequant + vendor GEMM.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
HEAD=${1:-unknown}
O=gpurun_out
FP=$(python -c "import bench; print(bench.kernel_fingerprint())")
: > $O/r04_whole_model_gemv_kernel_stats.txt
for ctx in 64 2048; do
  rm -rf $O/prof_wm
  rocprofv3 --kernel-trace --stats -d $O/prof_wm -o wm -- python tools/bench_decode_model.py --layout gemv --contexts $ctx --steps 32 > $O/r04_wm_$ctx.log 2>&1
  DB=$(find $O/prof_wm -name "*.db" | head -1)
  { echo "# git head $HEAD; kernel source fingerprint: $FP"; echo "# rocprofv3 --kernel-trace --stats -- python tools/bench_decode_model.py --layout gemv --contexts $ctx --steps 32"; grep "tok/s" $O/r04_wm_$ctx.log; python tools/rocpd_summary.py $DB | head -24; echo; } >> $O/r04_whole_model_gemv_kernel_stats.txt 2>&1
done
rm -rf $O/prof_wm $O/pmc_regb4
cat > /tmp/one_gemm.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from autoawq_amd import ops
from bench import rand_packed, rand_packed_nk
dev = torch.device("cuda"); gen = torch.Generator(device=dev).manual_seed(0)
K, N, M = 4096, 11008, 16384
qw, qz, sc = rand_packed(K, N, 128, dev, gen)
x = torch.randn((M, K), device=dev, generator=gen).half()
for _ in range(3):
    ops.gemm_forward(x, qw, sc, qz, flags=ops.gemm_flags(ops.KERNEL_REGB, nlog=1))
nq, nz, ns = rand_packed_nk(K, N, 128, dev, gen)
for _ in range(3):
    ops.gemv_forward(x, nq, ns, nz, 128, flags=ops.gemm_flags(kernel=ops.GEMV_KERNEL_PREFILL))
W = ops.dequantize_weights(qw, sc, qz)
for _ in range(3):
    torch.matmul(x, W)
torch.cuda.synchronize()
PY
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_regb4 -o p1 -- python /tmp/one_gemm.py > $O/pmc_regb4.log 2>&1
python3 - > $O/r04_pmc_mfma_prefill.txt <<PY
import csv, glob, collections
print("# git head $HEAD; kernel source fingerprint: $FP")
print("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace  (own pass; tools/prof_r04_extra.sh)")
print("# 4096 x 11008, M = 16384: gemm_regb (GEMM layout), its N-major form on the GEMV layout's buffers, the vendor GEMM after dequantise")
for f in sorted(glob.glob("$O/pmc_regb4/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in agg.items():
        if not any(s in k for s in ("regb", "Cijk", "dequant")):
            continue
        m = {n: sum(v) / len(v) for n, v in c.items()}
        busy, act = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), m.get("GRBM_GUI_ACTIVE", 0)
        # SQ_VALU_MFMA_BUSY_CYCLES: summed over the chip's SIMD-level counters as rocprofv3 reports them; ratio to GRBM_GUI_ACTIVE x 4 SIMDs x 256 CUs
        frac = busy / (act / 8 * 1024) if act else float("nan")  # GRBM_GUI_ACTIVE is summed over the 8 XCDs (profiles/r02_regb_counters_and_experiments.txt)
        print(f"{k:90s} n={len(next(iter(c.values()))):3d}  " + "  ".join(f"{n}={v:.3e}" for n, v in sorted(m.items())) + f"   MFMA busy / (GRBM active x 1024 SIMDs) = {frac:.3f}")
PY
rm -rf $O/pmc_regb4
cat $O/r04_whole_model_gemv_kernel_stats.txt | cut -c1-180 | head -60; cat $O/r04_pmc_mfma_prefill.txt | cut -c1-260
