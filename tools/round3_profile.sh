# round 3: the measurement record of the kernels that ship -- bench (default and driver protocol), rocprofv3 kernel stats of the
# same command, HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes), SQ / TCC counters (four passes) -> gpurun_out/$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03p}
mkdir -p $O
cd $R
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 200 $O/bench.json; echo
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_protocol.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-variants > $O/bench_under_rocprof.json 2> $O/rocprof1.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 tools/x3_profile_run.py x3 > /dev/null 2> $O/rocprof2.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 tools/x3_profile_run.py x3 > /dev/null 2> $O/rocprof3.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $O/sq_a -o a -- python3 tools/x3_profile_run.py x3 > /dev/null 2> $O/sq_a.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/sq_b -o b -- python3 tools/x3_profile_run.py x3 > /dev/null 2> $O/sq_b.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum --output-format csv -d $O/sq_c -o c -- python3 tools/x3_profile_run.py x3 > /dev/null 2> $O/sq_c.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $O/sq_d -o d -- python3 tools/x3_profile_run.py x3 > /dev/null 2> $O/sq_d.err || echo "pass d (MFMA busy) not available"
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py $F $W $O/hbm_traffic.json
python tools/pmc_sq.py $O/pmc_sq.json $(find $O/sq_a $O/sq_b $O/sq_c $O/sq_d -name "*counter_collection.csv")
find $O -name "*kernel_stats.csv"
