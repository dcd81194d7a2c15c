# GPU box: the whole -m gpu suite in one process, then the default bench (host_path included)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_check; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -15 $O/pytest.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("bench", round(d["value"],1), "img/s", d["ms_per_step"], "ms; frac", round(d["roofline"]["frac"],4), "host_path", json.dumps(d.get("host_path"))[:600])
PY
