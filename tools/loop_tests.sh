#!/bin/bash
# usage: tools/loop_tests.sh <count> <pytest -k expression>   (GPU box)
n=$1; shift
for i in $(seq 1 $n); do
  python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "$1" 2>&1 | grep -v amdgpu | tail -12
done
