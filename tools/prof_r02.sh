#!/bin/bash
# round-2 evidence run (GPU box): rocprofv3 --kernel-trace --stats of bench.py, --pmc FETCH_SIZE pass, summaries as text
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/prof_r02 gpurun_out/pmc_r02 gpurun_out/pmc_probe_r02
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r02 -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02_prof_bench.log 2>&1
DB=$(find gpurun_out/prof_r02 -name "*.db" | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline   (tools/prof_r02.sh; every leg of bench.py)"; echo "# bench line of this profiled run:"; grep '^{"metric' gpurun_out/r02_prof_bench.log | cut -c1-400; python tools/rocpd_summary.py $DB; } > gpurun_out/r02_bench_kernel_trace_stats.txt 2>&1
rm -rf gpurun_out/prof_r02
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r02 -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r02_prof_bench_headline.log 2>&1
DB=$(find gpurun_out/prof_r02 -name "*.db" | head -1)
{ echo; echo "# ---- the headline leg alone: rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary"; grep '^{"metric' gpurun_out/r02_prof_bench_headline.log | cut -c1-400; python tools/rocpd_summary.py $DB | head -12; } >> gpurun_out/r02_bench_kernel_trace_stats.txt 2>&1
rm -rf gpurun_out/prof_r02
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_probe_r02 -o probe -- tools/bin/stream_probe2 > gpurun_out/r02_pmc_probe.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_r02 -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/r02_pmc_bench.log 2>&1
{ echo "# rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary   (own pass; tools/prof_r02.sh)"; python tools/pmc_summary.py gpurun_out/pmc_r02 gpurun_out/pmc_probe_r02; } > gpurun_out/r02_pmc_fetch_size.txt 2>&1
rm -rf gpurun_out/pmc_r02 gpurun_out/pmc_probe_r02
tail -5 gpurun_out/r02_bench_kernel_trace_stats.txt; tail -12 gpurun_out/r02_pmc_fetch_size.txt
