#!/bin/bash
# HBM traffic of the decode kernels from the TCC fabric counters (MI355X_MICROARCH.md, HBM section):
# separate --pmc pass, calibrated on a kernel with a known byte count (the linear-read probe).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_bench gpurun_out/pmc_probe
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_probe -o probe -- tools/bin/stream_probe2 > gpurun_out/pmc_probe.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_bench -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_bench.log 2>&1
find gpurun_out/pmc_bench gpurun_out/pmc_probe -name "*.csv" | head
