#!/bin/bash
# what the driver runs at round end, on a GPU box: the default `pytest -m gpu` selection, smoke(), bench.py
cd $GRAFT_REPO_ROOT
O=gpurun_out/driver_like
mkdir -p $O
(time python -m pytest tests/ -x -q -m gpu) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee $O/rc.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/rc.txt
(time python bench.py) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
tail -4 $O/pytest_gpu.log; tail -2 $O/smoke.log; cut -c1-260 $O/bench.json; tail -4 $O/bench.err
