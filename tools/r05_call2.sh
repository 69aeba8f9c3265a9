#!/bin/bash
# Round 5, second GPU call: the new batched kernel (sweep + its tests), then the whole GPU suite once with durations.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_call2
mkdir -p "$OUT"
export TMPDIR=/tmp
step() { local name=$1 secs=$2; shift 2; local t0=$SECONDS; timeout "$secs" "$@" > "$OUT/$name.txt" 2>&1; echo "$name: rc=$?, $((SECONDS - t0)) s" | tee -a "$OUT/summary.txt"; }
step sweep 400 python tools/sweep_gemv_batch.py
step batch_tests 600 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "gemv_batch or gemv_lds or gemv_layout_vs or gemv_rows_refuses or gemv_module"
step suite 1100 python -m pytest tests/ -q -m gpu -p no:cacheprovider --durations=150
tail -n 5 "$OUT"/sweep.txt "$OUT"/batch_tests.txt; tail -n 30 "$OUT"/suite.txt | cut -c1-200
cat "$OUT/summary.txt"
