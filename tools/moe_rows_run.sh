#!/bin/bash
# MoE decode on GEMV-layout expert twins (round 6): the measurements behind profiles/r06_moe_rows.txt, on a GPU box:
#   the tests of the path, the bs = 4 block (twins vs GEMM-layout stacks), the parts-per-matrix sweep on the bench's routing,
#   the token sweep with the hand-over lifted, the prefill block, the dense small-batch probe.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/moe_rows
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_moe_rows.py -x -q -m gpu 2>&1 | tail -5 | tee $O/tests.txt
python tools/bench_moe.py 2>&1 | grep -v amdgpu.ids | tee $O/bench.txt
timeout 900 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/parts.txt
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import bench_moe
from autoawq_amd.modules.fused import moe
for p1, p2 in ((64, 64), (128, 64), (256, 64), (512, 64), (256, 32), (256, 96), (0, 0)):
    moe.ROWS_PARTS = (p1, p2)
    print("parts", (p1, p2), end=": ")
    bench_moe.run()
PY
python tools/bench_moe.py --sweep 2>&1 | grep -v amdgpu.ids | tee $O/sweep.txt
python tools/bench_moe.py --prefill 2>&1 | grep -v amdgpu.ids | tee $O/prefill.txt
python tools/probe_pairs_dense.py 2>&1 | grep -v amdgpu.ids | tee $O/pairs_dense.txt
