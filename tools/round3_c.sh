# round 3, third pass: byte A planes on the rounded-bf16 path -- tests that touch it, config 5 timing, its kernel breakdown
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03c}
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "bf16 or config5 or fused or workspace or dp_step" > $O/tests.txt 2>&1; tail -5 $O/tests.txt
python tools/bench_config5.py --only bf16 > $O/config5.json 2> $O/config5.err; cat $O/config5.json
KURBM_X3_BYTES=0 python tools/bench_config5.py --only bf16 > $O/config5_nobytes.json 2>> $O/config5.err; cat $O/config5_nobytes.json
KURBM_X3_TALL=1 python tools/bench_config5.py --only bf16 > $O/config5_tall.json 2>> $O/config5.err; cat $O/config5_tall.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -o s -- python3 tools/bench_config5.py --only bf16 > /dev/null 2> $O/rocprof.err
head -8 $O/stats_c5/s_kernel_stats.csv | cut -c1-160
