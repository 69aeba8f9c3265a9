cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06c
mkdir -p $O
python tools/bench_moe.py --prefill > $O/moe_prefill_eager.txt 2>&1
rm -rf /tmp/prof_moe
rocprofv3 --kernel-trace --stats -d /tmp/prof_moe -o moe -- python tools/bench_moe.py --prefill > $O/moe_prefill_prof.log 2>&1
DB=$(find /tmp/prof_moe -name "*.db" | head -1)
python tools/rocpd_summary.py $DB > $O/moe_prefill_kernel_stats.txt 2>&1
cat $O/moe_prefill_eager.txt; head -40 $O/moe_prefill_kernel_stats.txt
