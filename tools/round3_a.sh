# round 3, first measurement pass: tests, bench, kernel breakdown of the real-valued / Gaussian variants, config 5
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03a}
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; tail -4 $O/tests.txt
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json; echo
for v in "binary bern" "real bern" "real gauss"; do
  set -- $v
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$1_$2 -o s -- python3 tools/x3_profile_run.py x3 4096 784 1024 $1 $2 > /dev/null 2> $O/rocprof_$1_$2.err
done
python tools/bench_config5.py > $O/config5.json 2> $O/config5.err; cat $O/config5.json
KURBM_X3_TALL=0 python tools/bench_config5.py > $O/config5_tall0.json 2>> $O/config5.err; cat $O/config5_tall0.json
find $O -name "*kernel_stats.csv"
