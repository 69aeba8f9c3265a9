cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06c
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_moe_rows.py tests/test_gpu_parity.py tests/test_mixtral.py tests/test_gpu_ep.py tests/test_gpu_shims.py tests/test_gpu_baseline_configs.py tests/test_gpu_route_a_replay.py -x -q -m gpu -k "moe or grouped or mixtral or ep or route or Mixtral" 2>&1 | tail -8 > $O/moe_rows_tests.txt
cat $O/moe_rows_tests.txt
python tools/bench_moe.py --sweep 2>&1 | grep -v amdgpu.ids | tee $O/moe_rows_sweep.txt
