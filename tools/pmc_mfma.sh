#!/bin/bash
# MFMA utilisation of the fused tiled GEMM (BASELINE config 3: 4096 x 11008, M = 16384) from the SQ counters
# (own --pmc pass, MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts busy cycles per SIMD, GRBM_GUI_ACTIVE gives
# the effective clock).  Output: gpurun_out/pmc_mfma/*.csv
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_mfma
cat > /tmp/one_gemm.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from autoawq_amd import ops
from bench import rand_packed
dev = torch.device("cuda"); gen = torch.Generator(device=dev).manual_seed(0)
K, N, M = 4096, 11008, 16384
qw, qz, sc = rand_packed(K, N, 128, dev, gen)
x = torch.randn((M, K), device=dev, generator=gen).half()
for nlog in (2, 1):
    for _ in range(3):
        ops.gemm_forward(x, qw, sc, qz, flags=ops.gemm_flags(ops.KERNEL_TILED, nlog=nlog))
W = ops.dequantize_weights(qw, sc, qz)
for _ in range(3):
    torch.matmul(x, W)
torch.cuda.synchronize()
PY
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_mfma -o mfma -- python /tmp/one_gemm.py > gpurun_out/pmc_mfma.log 2>&1
find gpurun_out/pmc_mfma -name "*.csv" | head
