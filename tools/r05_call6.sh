#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_call6
mkdir -p "$OUT"
export TMPDIR=/tmp
step() { local name=$1 secs=$2; shift 2; local t0=$SECONDS; timeout "$secs" "$@" > "$OUT/$name.txt" 2>&1; echo "$name: rc=$?, $((SECONDS - t0)) s" | tee -a "$OUT/summary.txt"; }
step dbg 100 python tools/dbg_batch.py
step tests 300 python -m pytest tests/ -q -x -m gpu -p no:cacheprovider -k "gemv_batch"
step sweep 300 python tools/sweep_gemv_batch.py
step trace 200 python tools/trace_gemv_batch.py
grep -v amdgpu "$OUT"/dbg.txt | head -12; tail -n 4 "$OUT"/tests.txt | cut -c1-200; grep "N=11008 M=\|N=4096 M=\|check" "$OUT"/sweep.txt | cut -c1-250
cat "$OUT/summary.txt"
