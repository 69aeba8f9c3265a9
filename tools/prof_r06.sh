#!/bin/bash
# round-6 evidence run (GPU box): rocprofv3 --kernel-trace --stats of bench.py (every leg, then the headline alone), own --pmc passes
# (FETCH_SIZE of the headline with its calibration on the linear-read probe; FETCH_SIZE and WRITE_SIZE of the batched-decode legs;
# MFMA-busy counters of the prefill GEMMs), summaries as text.  Every file records the git head AND the kernel-source fingerprint
# bench.py checks.  Usage: prof_r06.sh <git head> [quick]   (quick: the counter passes only)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
HEAD=${1:-unknown}
O=gpurun_out
FP=$(python -c "import bench; print(bench.kernel_fingerprint())")
mkdir -p $O/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $O/bin/stream_probe2 tools/stream_probe2.hip > $O/r06_probe_build.log 2>&1
if [ "$2" != "quick" ]; then
rm -rf $O/prof_r06
rocprofv3 --kernel-trace --stats -d $O/prof_r06 -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r06_prof_bench.log 2>&1
DB=$(find $O/prof_r06 -name "*.db" | head -1)
{ echo "# git head $HEAD; kernel source fingerprint: $FP"; echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline   (tools/prof_r06.sh; every leg of bench.py)"; echo "# bench line of this profiled run:"; grep '^{"metric' $O/r06_prof_bench.log | cut -c1-400; python tools/rocpd_summary.py $DB; } > $O/r06_bench_kernel_trace_stats.txt 2>&1
rm -rf $O/prof_r06
rocprofv3 --kernel-trace --stats -d $O/prof_r06 -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/r06_prof_bench_headline.log 2>&1
DB=$(find $O/prof_r06 -name "*.db" | head -1)
{ echo; echo "# ---- the headline leg alone: rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary"; grep '^{"metric' $O/r06_prof_bench_headline.log | cut -c1-400; python tools/rocpd_summary.py $DB; } >> $O/r06_bench_kernel_trace_stats.txt 2>&1
rm -rf $O/prof_r06
fi
rm -rf $O/pmc_r06 $O/pmc_probe_r06 $O/pmc_bs_f $O/pmc_bs_w $O/pmc_mfma_r06
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_probe_r06 -o probe -- $O/bin/stream_probe2 > $O/r06_pmc_probe.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_r06 -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/r06_pmc_bench.log 2>&1
{ echo "# git head $HEAD; kernel source fingerprint: $FP"; echo "# rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary   (own pass; tools/prof_r06.sh; the headline = the DEPENDENT chain)"; KERNEL=awq_gemv_rows_kernel python tools/pmc_summary.py $O/pmc_r06 $O/pmc_probe_r06; } > $O/r06_pmc_fetch_size.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_bs_f -o bs -- python tools/pmc_gemm_bs.py run > $O/r06_pmc_bs_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_bs_w -o bs -- python tools/pmc_gemm_bs.py run > $O/r06_pmc_bs_w.log 2>&1
{ echo "# git head $HEAD; kernel source fingerprint: $FP"; echo "# rocprofv3 --pmc FETCH_SIZE (and, own pass, WRITE_SIZE) --kernel-trace -- python tools/pmc_gemm_bs.py run   (tools/prof_r06.sh): bench.py's gemm_bs legs, 4096 x 11008 g128, 28 distinct matrices"; grep "kernel" $O/r06_pmc_bs_f.log | sed 's/^/# /'; python tools/pmc_gemm_bs.py summarize $O/pmc_bs_f $O/pmc_bs_w $O/pmc_probe_r06; } > $O/r06_pmc_gemm_bs.txt 2>&1
bash tools/pmc_mfma_r06.sh "$HEAD" "$FP" > $O/r06_pmc_mfma_prefill.txt 2>&1
rm -rf $O/pmc_moe
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_moe -o moe -- python tools/bench_moe.py > $O/r06_pmc_moe.log 2>&1
{ echo "# git head $HEAD; kernel source fingerprint: $FP"; echo "# rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python tools/bench_moe.py   (own pass; tools/prof_r06.sh): the MoE decode block on the GEMV-layout twins (awq_gemv_rows_kernel, grouped form) and on the GEMM-layout stacks (awq_gemv_mfma_kernel, grouped)"; grep Mixtral $O/r06_pmc_moe.log | cut -c1-400; python tools/pmc_moe.py $O/pmc_moe; } > $O/r06_pmc_moe.txt 2>&1
rm -rf $O/pmc_moe
rm -rf $O/pmc_r06 $O/pmc_probe_r06 $O/pmc_bs_f $O/pmc_bs_w $O/bin
cat $O/r06_pmc_moe.txt
tail -12 $O/r06_bench_kernel_trace_stats.txt 2>/dev/null; tail -8 $O/r06_pmc_fetch_size.txt; cat $O/r06_pmc_gemm_bs.txt; tail -20 $O/r06_pmc_mfma_prefill.txt
