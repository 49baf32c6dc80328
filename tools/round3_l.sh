# all GPU tests, then bench.py with its variants (real-valued data, Gaussian visibles), config 5 (three paths), the score cost
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03l}; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -1 $O/tests.txt
timeout -k 10 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("steps/s %.0f" % d["value"], "| " + "  ".join("%s %.1f us" % (k, 1e3 * v["ms_per_step"]) for k, v in d["paths"].items()))
PY
timeout -k 10 600 python tools/bench_config5.py > $O/config5.json 2> $O/config5.err; cut -c1-700 $O/config5.json
timeout -k 10 600 python tools/score_times.py > $O/score_times.txt 2>&1; tail -6 $O/score_times.txt
