#!/bin/bash
# Round 5, first GPU call: everything staged at the end of round 4 (tools/experimental/README.md) in ONE box session, each step under
# its own timeout, outputs under gpurun_out/r05_first/.   gpurun --timeout 1500 -- 'bash tools/r05_first_call.sh'
# BEFORE the call, in the build container: `for p in gemm_skinny_nk prefill_attention mall_prefetch; do python tools/experimental/$p/probe.py --build-only; done`
# (tools/bin/ travels with the snapshot; the probes then find their libraries built and the box compiles nothing).
# Budget: smoke ~1 min, three probes ~2 min each (they build their own libraries: hipcc on the box), suite under xdist ~2-3 min.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_first
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "head: $(cat .git_head 2>/dev/null || echo unknown)" > "$OUT/summary.txt"
step() {  # name, seconds, command...
    local name=$1 secs=$2; shift 2
    local t0=$SECONDS
    timeout "$secs" "$@" > "$OUT/$name.txt" 2>&1
    local rc=$?
    echo "$name: rc=$rc, $((SECONDS - t0)) s" | tee -a "$OUT/summary.txt"
}
step smoke 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')"
step skinny_nk 300 python tools/experimental/gemm_skinny_nk/probe.py
step prefill_attn 400 python tools/experimental/prefill_attention/probe.py
step mall_prefetch 300 python tools/experimental/mall_prefetch/probe.py
# the GPU suite on 8 worker processes (is it xdist-safe on one GPU, and how long does it take?); the round-end run stays the
# driver's exact single-process command
step suite_xdist8 900 python -m pytest tests/ -q -m gpu -p no:cacheprovider -n 8 --durations=120
tail -n 3 "$OUT"/*.txt | tail -n 60
cat "$OUT/summary.txt"
