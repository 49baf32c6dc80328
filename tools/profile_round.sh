set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r02a}
mkdir -p $O
cd $R
python bench.py > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-variants > $O/bench_under_rocprof.json 2> $O/rocprof1.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 tools/x3_profile_run.py x3 > /dev/null 2> $O/rocprof2.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 tools/x3_profile_run.py x3 > /dev/null 2> $O/rocprof3.err
find $O -name "*.csv" | head -20
