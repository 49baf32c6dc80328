# round 3, second pass: all GPU tests, the verbose-fit cost, config 5 with the new tile choice
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03b}
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/tests.txt 2>&1; tail -12 $O/tests.txt
python tools/score_times.py > $O/score_times.txt 2>&1; cat $O/score_times.txt
python tools/bench_config5.py > $O/config5.json 2> $O/config5.err; cat $O/config5.json
