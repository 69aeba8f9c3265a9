cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06c
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_moe_rows.py tests/test_gpu_parity.py tests/test_mixtral.py tests/test_gpu_ep.py tests/test_gpu_shims.py tests/test_gpu_baseline_configs.py tests/test_gpu_route_a_replay.py -x -q -m gpu -k "moe or grouped or mixtral or ep or route or Mixtral or regb" 2>&1 | tail -12 > $O/moe_prefill_tests.txt
cat $O/moe_prefill_tests.txt
rm -rf /tmp/prof_moe
rocprofv3 --kernel-trace --stats -d /tmp/prof_moe -o moe -- python tools/bench_moe.py --prefill > $O/moe_prefill_prof.log 2>&1
DB=$(find /tmp/prof_moe -name "*.db" | head -1)
python tools/rocpd_summary.py $DB > $O/moe_prefill_fused_kernel_stats.txt 2>&1
grep Mixtral $O/moe_prefill_prof.log; grep -i "awq\|calls" $O/moe_prefill_fused_kernel_stats.txt | cut -c1-180 | head -24
