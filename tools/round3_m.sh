# README numbers of the final kernels: DBN config 4, fit() loop, the example-sized config
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03m}; mkdir -p $O; cd $R
timeout -k 10 600 python tools/bench_dbn.py > $O/dbn_config4.json 2> $O/dbn.err; cut -c1-600 $O/dbn_config4.json
timeout -k 10 600 python tools/bench_fit.py > $O/fit.json 2> $O/fit.err; cut -c1-600 $O/fit.json
timeout -k 10 600 python tools/bench_example_config.py > $O/example.json 2> $O/example.err; cut -c1-600 $O/example.json
timeout -k 10 300 python tools/soak.py > $O/soak.txt 2>&1; tail -2 $O/soak.txt
