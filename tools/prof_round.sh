#!/bin/bash
# round-end evidence run (GPU box): bench JSON, rocprofv3 kernel-trace stats of the same command,
# FETCH_SIZE counter pass, TP shard timing.  Outputs under gpurun_out/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r75}
python bench.py > gpurun_out/${TAG}_bench.json.log 2>&1
rm -rf gpurun_out/prof_${TAG} gpurun_out/pmc_${TAG}
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${TAG} -o bench -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-whole-model > gpurun_out/${TAG}_prof_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG} -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-whole-model > gpurun_out/${TAG}_pmc_bench.log 2>&1
python tools/bench_tp_shards.py > gpurun_out/${TAG}_tp_shards.log 2>&1
find gpurun_out/prof_${TAG} gpurun_out/pmc_${TAG} -type f | head -20
