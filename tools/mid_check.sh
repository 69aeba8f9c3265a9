cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06c
mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > $O/mid_pytest_gpu.txt
cat $O/mid_pytest_gpu.txt
timeout 900 python bench.py > $O/mid_bench.json 2> $O/mid_bench.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r06c/mid_bench.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"])
for k in ("moe_bs4", "moe_prefill"):
    print(k, json.dumps(d.get(k))[:600])
PY
