#!/bin/bash
# rocprofv3 --kernel-trace --stats of the whole-model decode (tools/bench_decode_model.py, GEMV layout) at 64 and 2048 tokens of
# context: per-kernel time of one token (the five launches of a fused decoder block + lm_head).  Usage: whole_model_trace.sh <git head>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
HEAD=${1:-unknown}
O=gpurun_out
FP=$(python -c "import bench; print(bench.kernel_fingerprint())")
: > $O/r06_whole_model_gemv_kernel_stats.txt
for CTX in 64 2048; do
  rm -rf /tmp/prof_wm
  rocprofv3 --kernel-trace --stats -d /tmp/prof_wm -o wm -- python tools/bench_decode_model.py --layout gemv --contexts $CTX --steps 32 > $O/r06_wm_$CTX.log 2>&1
  DB=$(find /tmp/prof_wm -name "*.db" | head -1)
  { echo "# git head $HEAD; kernel source fingerprint: $FP"; echo "# rocprofv3 --kernel-trace --stats -- python tools/bench_decode_model.py --layout gemv --contexts $CTX --steps 32"; grep "whole-model" $O/r06_wm_$CTX.log; python tools/rocpd_summary.py $DB | grep -v "at::native\|rocclr" | head -24 | cut -c1-200; echo; } >> $O/r06_whole_model_gemv_kernel_stats.txt
done
python tools/bench_decode_model.py --layout gemv --contexts 64,512,2048 2>&1 | grep -v amdgpu.ids >> $O/r06_whole_model_gemv_kernel_stats.txt
cat $O/r06_whole_model_gemv_kernel_stats.txt | cut -c1-180
