#!/bin/bash
# End-of-round evidence at ONE head (no product commit follows it): smoke() in fresh processes, the full GPU suite, the whole
# suite under the guard-band allocator (both placements), the rocprofv3 passes, the bench line.  Logs are written unedited, named
# by the head.  usage (GPU box): tools/final_r04.sh <git head>
cd $GRAFT_REPO_ROOT
HEAD=${1:-unknown}
O=gpurun_out/r4final
mkdir -p $O
for i in 1 2; do
  timeout 600 python -c "import __graft_entry__ as e; e.smoke()" > $O/r04_smoke_${HEAD}_$i.log 2>&1; echo "smoke $i rc=$?" | tee -a $O/summary.txt
done
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/r04_pytest_gpu_${HEAD}.log 2>&1; echo "pytest -m gpu rc=$? $(tail -1 $O/r04_pytest_gpu_${HEAD}.log)" | tee -a $O/summary.txt
bash tools/guard_run.sh $O/guard > $O/guard_console.log 2>&1; cat $O/guard/summary.txt | tee -a $O/summary.txt
bash tools/prof_r04.sh $HEAD > $O/prof_console.log 2>&1; tail -5 $O/prof_console.log
timeout 900 python bench.py > $O/r04_bench_n1_final.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
cp gpurun_out/r04_bench_kernel_trace_stats.txt gpurun_out/r04_pmc_fetch_size.txt $O/ 2>/dev/null
cat $O/summary.txt
