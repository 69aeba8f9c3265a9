#!/bin/bash
# end-of-round evidence at one head (GPU box): smoke(), the driver's `pytest -m gpu` selection, with `full` the whole matrix
# (AWQ_FULL_MATRIX=1), tools/prof_r06.sh, the bench line.  Usage: final_r06.sh <git head> [full]
cd $GRAFT_REPO_ROOT
HEAD=${1:-unknown}
O=gpurun_out/r06_final_$HEAD
mkdir -p $O
FP=$(python -c "import bench; print(bench.kernel_fingerprint())")
echo "head $HEAD fingerprint $FP" > $O/fingerprint.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_smoke.log 2>&1; echo "smoke rc=$?" >> $O/fingerprint.txt
(time python -m pytest tests/ -q -m gpu --durations=20) > $O/r06_pytest_gpu_default.log 2>&1; echo "pytest default rc=$?" >> $O/fingerprint.txt
if [ "$2" == "full" ]; then
  (time AWQ_FULL_MATRIX=1 python -m pytest tests/ -q -m gpu) > $O/r06_pytest_gpu_full_matrix.log 2>&1; echo "pytest full rc=$?" >> $O/fingerprint.txt
fi
bash tools/prof_r06.sh $HEAD > $O/r06_prof.log 2>&1
cp gpurun_out/r06_bench_kernel_trace_stats.txt gpurun_out/r06_pmc_fetch_size.txt gpurun_out/r06_pmc_gemm_bs.txt gpurun_out/r06_pmc_mfma_prefill.txt gpurun_out/r06_pmc_moe.txt $O/ 2>/dev/null
python bench.py > $O/r06_bench_n1_final.json 2> $O/r06_bench_n1_final.err; echo "bench rc=$?" >> $O/fingerprint.txt
cat $O/fingerprint.txt; tail -3 $O/r06_pytest_gpu_default.log; tail -3 $O/r06_pytest_gpu_full_matrix.log 2>/dev/null; cut -c1-300 $O/r06_bench_n1_final.json
