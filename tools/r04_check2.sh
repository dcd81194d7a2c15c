# GPU box: the whole -m gpu suite, the default bench, the codec seam rates, then workgroup timelines of conv_pc / conv_pk
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_check2; mkdir -p $O; cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -12 $O/pytest.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("bench", round(d["value"],1), "img/s", round(d["ms_per_step"],3), "ms; frac", round(d["roofline"]["frac"],4), "per-layer", round(d["roofline"]["frac_of_per_layer_roofline"],4))
print("host_path", json.dumps(d.get("host_path"))[:700])
print("secondary", json.dumps(d.get("secondary"))[:900])
PY
timeout -k 10 600 python tools/codec_seam_rate.py 1024 > $O/codec.json 2> $O/codec.err; tail -c 3000 $O/codec.json; tail -3 $O/codec.err
TLV="32 32r 64 64r 128 256" bash tools/r04_tl.sh > $O/tl.txt 2>&1; grep -E "^==|gn_fold  |prologue  |whole workgroup|exit  |entry  |shader clock" $O/tl.txt | head -80
