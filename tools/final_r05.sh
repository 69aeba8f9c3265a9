#!/bin/bash
# End-of-round evidence at ONE head: smoke(), the driver's GPU selection, the FULL matrix (AWQ_FULL_MATRIX=1: every parametrisation +
# both guard-band placements) when asked for ("full"), the rocprofv3 passes, the bench line.  Logs unedited, named by the head.
# usage (GPU box): tools/final_r05.sh <git head> [full]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
HEAD=${1:-unknown}
O=gpurun_out/r5final
mkdir -p $O
timeout 600 python -c "import __graft_entry__ as e; e.smoke()" > $O/r05_smoke_${HEAD}.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
T0=$SECONDS
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/r05_pytest_gpu_${HEAD}.log 2>&1; echo "pytest -m gpu rc=$? ($((SECONDS - T0)) s) $(tail -1 $O/r05_pytest_gpu_${HEAD}.log)" | tee -a $O/summary.txt
if [ "${2:-}" = "full" ]; then
  T0=$SECONDS
  AWQ_FULL_MATRIX=1 timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/r05_pytest_gpu_full_matrix_${HEAD}.log 2>&1; echo "full matrix rc=$? ($((SECONDS - T0)) s) $(tail -1 $O/r05_pytest_gpu_full_matrix_${HEAD}.log)" | tee -a $O/summary.txt
fi
bash tools/prof_r05.sh $HEAD > $O/prof_console.log 2>&1; tail -5 $O/prof_console.log
cp gpurun_out/r05_pmc_fetch_size.txt profiles/r05_pmc_fetch_size.txt 2>/dev/null  # (this tree's own pass: bench.py checks its fingerprint before reporting roofline.traffic)
timeout 900 python bench.py > $O/r05_bench_n1_final.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
cp gpurun_out/r05_bench_kernel_trace_stats.txt gpurun_out/r05_pmc_fetch_size.txt $O/ 2>/dev/null
cat $O/summary.txt
