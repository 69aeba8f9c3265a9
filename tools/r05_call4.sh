#!/bin/bash
# Round 5, fourth GPU call: gemv_batch v2 (staged activations, group-level scale): tests, sweep, trace; attention + guard tests.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_call4
mkdir -p "$OUT"
export TMPDIR=/tmp
step() { local name=$1 secs=$2; shift 2; local t0=$SECONDS; timeout "$secs" "$@" > "$OUT/$name.txt" 2>&1; echo "$name: rc=$?, $((SECONDS - t0)) s" | tee -a "$OUT/summary.txt"; }
step tests 400 python -m pytest tests/ -q -x -m gpu -p no:cacheprovider -k "gemv_batch or prefill_attention or repack or gemv_layout_prefill or guard or allreduce or quant_attention_fused or graphed"
step sweep 300 python tools/sweep_gemv_batch.py
step trace 200 python tools/trace_gemv_batch.py
tail -n 8 "$OUT"/tests.txt | cut -c1-200; grep "M=8:\|M=16:\|M=32:\|M=64:" "$OUT"/sweep.txt | head -8 | cut -c1-250
cat "$OUT/summary.txt"
