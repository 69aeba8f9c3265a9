#!/bin/bash
# round-5 evidence run (GPU box): rocprofv3 --kernel-trace --stats of bench.py (every leg, then the headline alone), an own
# --pmc FETCH_SIZE pass with its calibration on the linear-read probe (built here: tools/bin/ holds experiment builds only and is kept empty otherwise), summaries as
# text.  Every file records the git head AND the kernel-source fingerprint bench.py checks.  Usage: prof_r05.sh <git head>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
HEAD=${1:-unknown}
O=gpurun_out
FP=$(python -c "import bench; print(bench.kernel_fingerprint())")
mkdir -p $O/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $O/bin/stream_probe2 tools/stream_probe2.hip > $O/r05_probe_build.log 2>&1
rm -rf $O/prof_r05 $O/pmc_r04 $O/pmc_probe_r04
rocprofv3 --kernel-trace --stats -d $O/prof_r05 -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r05_prof_bench.log 2>&1
DB=$(find $O/prof_r05 -name "*.db" | head -1)
{ echo "# git head $HEAD; kernel source fingerprint: $FP"; echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline   (tools/prof_r05.sh; every leg of bench.py)"; echo "# bench line of this profiled run:"; grep '^{"metric' $O/r05_prof_bench.log | cut -c1-400; python tools/rocpd_summary.py $DB; } > $O/r05_bench_kernel_trace_stats.txt 2>&1
rm -rf $O/prof_r05
rocprofv3 --kernel-trace --stats -d $O/prof_r05 -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/r05_prof_bench_headline.log 2>&1
DB=$(find $O/prof_r05 -name "*.db" | head -1)
{ echo; echo "# ---- the headline leg alone: rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary"; grep '^{"metric' $O/r05_prof_bench_headline.log | cut -c1-400; python tools/rocpd_summary.py $DB; } >> $O/r05_bench_kernel_trace_stats.txt 2>&1
rm -rf $O/prof_r05
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_probe_r04 -o probe -- $O/bin/stream_probe2 > $O/r05_pmc_probe.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_r04 -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/r05_pmc_bench.log 2>&1
{ echo "# git head $HEAD; kernel source fingerprint: $FP"; echo "# rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary   (own pass; tools/prof_r05.sh)"; KERNEL=awq_gemv_rows_kernel python tools/pmc_summary.py $O/pmc_r04 $O/pmc_probe_r04; } > $O/r05_pmc_fetch_size.txt 2>&1
rm -rf $O/pmc_r04 $O/pmc_probe_r04 $O/bin
tail -16 $O/r05_bench_kernel_trace_stats.txt; tail -12 $O/r05_pmc_fetch_size.txt
