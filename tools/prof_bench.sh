set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q --tb=line 2>&1 | tail -5 > gpurun_out/r23_tests.log
python bench.py --steps 100 --warmup 10 > gpurun_out/r23_bench.log 2>&1
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r23 -o bench -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r23_prof_bench.log 2>&1
ls -R gpurun_out/prof_r23 | head -30
