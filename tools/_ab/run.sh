cp autoawq_amd/csrc/libawq_hip.so /tmp/cur.so
for v in cur base cur base; do
  case $v in cur) cp /tmp/cur.so autoawq_amd/csrc/libawq_hip.so;; *) cp tools/_ab/lib_$v.so autoawq_amd/csrc/libawq_hip.so;; esac
  timeout 120 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'])"
done
cp /tmp/cur.so autoawq_amd/csrc/libawq_hip.so
timeout 200 python tools/sweep_m8.py 2>&1 | grep "splitk= 0 waves=0 unit=0"
timeout 300 python tools/bench_moe.py 2>&1 | grep Mixtral | cut -c1-120
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "test_gemm_vs_oracle_all_variants or golden or moe or grouped" 2>&1 | tail -2
