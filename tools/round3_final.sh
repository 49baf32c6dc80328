# round 3, final kernels: all GPU tests, smoke, then the measurement record (tools/round3_profile.sh), config 5 under rocprofv3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03g}; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/tests.txt 2>&1; tail -1 $O/tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
bash tools/round3_profile.sh ${1:-r03g} > $O/profile.log 2>&1; tail -3 $O/profile.log
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5stats -o c -- python3 $R/tools/bench_config5.py --only bf16 > $O/config5_under_rocprof.json 2> $O/c5.err
find $O/c5stats -name "*kernel_stats.csv"
