# round 3: full GPU test tier, bench, verbose-fit cost, DBN config 4, config 5
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03h}
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/tests.txt 2>&1; tail -4 $O/tests.txt
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 200 $O/bench.json; echo
python tools/score_times.py > $O/score_times.txt 2>&1; cat $O/score_times.txt
python tools/bench_dbn.py > $O/dbn.json 2> $O/dbn.err; cat $O/dbn.json
python tools/bench_config5.py > $O/config5.json 2> $O/config5.err; cat $O/config5.json
python tools/bench_fit.py > $O/bench_fit.txt 2>&1; cat $O/bench_fit.txt
