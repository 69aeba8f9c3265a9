cd $GRAFT_REPO_ROOT
O=gpurun_out/r06d
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_ep.py tests/test_gpu_moe_rows.py tests/test_gpu_parity.py tests/test_decoder.py tests/test_mixtral.py -x -q -m gpu -k "ep or expert or grouped_rows or block_fusions or forward_ex or decoder or llama or graph or mixtral or moe_block" 2>&1 | tail -6 | tee $O/tests.txt
python tools/fx_overhead.py 2>&1 | grep -v amdgpu.ids | tee $O/fx_overhead.txt
