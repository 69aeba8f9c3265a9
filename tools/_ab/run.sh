cp autoawq_amd/csrc/libawq_hip.so /tmp/cur.so
for v in cur base; do
  case $v in cur) cp /tmp/cur.so autoawq_amd/csrc/libawq_hip.so;; *) cp tools/_ab/lib_$v.so autoawq_amd/csrc/libawq_hip.so;; esac
  for lay in gemm gemv gemvfast; do
  timeout 120 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --layout $lay 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v $lay', d['value'])"
  done
done
cp /tmp/cur.so autoawq_amd/csrc/libawq_hip.so
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "gemv" 2>&1 | tail -2
