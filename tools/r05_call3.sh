#!/bin/bash
# Round 5, third GPU call: phase trace of the batched kernel, repack / prefill-route timing, the default GPU suite (thinned) with durations.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_call3
mkdir -p "$OUT"
export TMPDIR=/tmp
step() { local name=$1 secs=$2; shift 2; local t0=$SECONDS; timeout "$secs" "$@" > "$OUT/$name.txt" 2>&1; echo "$name: rc=$?, $((SECONDS - t0)) s" | tee -a "$OUT/summary.txt"; }
step trace 300 python tools/trace_gemv_batch.py
step prefill_routes 200 python tools/time_prefill_routes.py
step suite 900 python -m pytest tests/ -q -m gpu -p no:cacheprovider --durations=40
tail -n 12 "$OUT"/prefill_routes.txt; tail -n 25 "$OUT"/suite.txt | cut -c1-200
cat "$OUT/summary.txt"
