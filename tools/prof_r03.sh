#!/bin/bash
# round-3 evidence run (GPU box): rocprofv3 --kernel-trace --stats of bench.py (every leg, then the headline alone), an own
# --pmc FETCH_SIZE pass with its calibration on the linear-read probe, summaries as text.  Usage: prof_r03.sh <git head>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
HEAD=${1:-unknown}
rm -rf gpurun_out/prof_r03 gpurun_out/pmc_r03 gpurun_out/pmc_probe_r03
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r03 -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03_prof_bench.log 2>&1
DB=$(find gpurun_out/prof_r03 -name "*.db" | head -1)
{ echo "# git head $HEAD"; echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline   (tools/prof_r03.sh; every leg of bench.py)"; echo "# bench line of this profiled run:"; grep '^{"metric' gpurun_out/r03_prof_bench.log | cut -c1-400; python tools/rocpd_summary.py $DB; } > gpurun_out/r03_bench_kernel_trace_stats.txt 2>&1
rm -rf gpurun_out/prof_r03
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r03 -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r03_prof_bench_headline.log 2>&1
DB=$(find gpurun_out/prof_r03 -name "*.db" | head -1)
{ echo; echo "# ---- the headline leg alone: rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary"; grep '^{"metric' gpurun_out/r03_prof_bench_headline.log | cut -c1-400; python tools/rocpd_summary.py $DB | head -14; } >> gpurun_out/r03_bench_kernel_trace_stats.txt 2>&1
rm -rf gpurun_out/prof_r03
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_probe_r03 -o probe -- tools/bin/stream_probe2 > gpurun_out/r03_pmc_probe.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_r03 -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/r03_pmc_bench.log 2>&1
{ echo "# git head $HEAD"; echo "# rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary   (own pass; tools/prof_r03.sh)"; KERNEL=awq_gemv_rows_kernel python tools/pmc_summary.py gpurun_out/pmc_r03 gpurun_out/pmc_probe_r03; } > gpurun_out/r03_pmc_fetch_size.txt 2>&1
rm -rf gpurun_out/pmc_r03 gpurun_out/pmc_probe_r03
tail -16 gpurun_out/r03_bench_kernel_trace_stats.txt; tail -12 gpurun_out/r03_pmc_fetch_size.txt
