# The measurement record of the kernels that ship, one script for every round:
#     bash tools/profile_round.sh TAG [part ...]          -> gpurun_out/TAG/   (copy what is to be judged into profiles/ as TAG_*)
# parts (default: all):
#   bench     bench.py with its defaults and under the driver's protocol (--steps 20 --warmup 5)
#   stats     rocprofv3 --kernel-trace --stats of bench.py (config 2, 0/1 data) and of the real-valued-data runs (grey / Bernoulli,
#             grey / Gaussian: the reference's default mode)
#   traffic   HBM bytes per launch: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (MI355X_MICROARCH.md, HBM section)
#   sq        SQ / TCC counter passes (waits, LDS conflicts, L2 hit rate, MFMA busy)
#   score     fit(verbose = 1) against the quiet loop, and the score pass's kernels
#   config5   one GPU's share of config 5 (time, then its kernels under rocprofv3)
#   small     the one-launch step of small RBMs against the five-launch path, and fit(verbose=1) at the reference example's size
# rocprofv3 gets python3 directly behind `--`, and counters are collected in passes of their own (no trace domains beside --pmc).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r04x}; shift
PARTS=${*:-bench stats traffic sq score config5 small}
O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
has() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
if has bench; then
  python bench.py > $O/bench.json 2> $O/bench.err; tail -c 200 $O/bench.json; echo
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_protocol.json 2>> $O/bench.err
fi
if has stats; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-variants > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err
  for v in "real bern" "real gauss"; do
    set -- $v
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$1_$2 -o s -- python3 tools/x3_profile_run.py x3 4096 784 1024 $1 $2 > /dev/null 2> $O/rocprof_$1_$2.err
  done
  find $O -name "*kernel_stats.csv"
fi
if has traffic; then
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 tools/x3_profile_run.py x3 > /dev/null 2> $O/rocprof_fetch.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 tools/x3_profile_run.py x3 > /dev/null 2> $O/rocprof_write.err
  F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
  python tools/pmc_traffic.py $F $W $O/hbm_traffic.json
fi
if has sq; then
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $O/sq_a -o a -- python3 tools/x3_profile_run.py x3 > /dev/null 2> $O/sq_a.err
  rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/sq_b -o b -- python3 tools/x3_profile_run.py x3 > /dev/null 2> $O/sq_b.err
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum --output-format csv -d $O/sq_c -o c -- python3 tools/x3_profile_run.py x3 > /dev/null 2> $O/sq_c.err
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $O/sq_d -o d -- python3 tools/x3_profile_run.py x3 > /dev/null 2> $O/sq_d.err || echo "pass d (MFMA busy) not available"
  python tools/pmc_sq.py $O/pmc_x3_gemm.json $(find $O/sq_a $O/sq_b $O/sq_c $O/sq_d -name "*counter_collection.csv")
fi
if has score; then
  python tools/score_times.py 2>&1 | grep -v amdgpu.ids > $O/score_times.txt; cat $O/score_times.txt
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_score -o s -- python3 tools/score_profile_run.py > /dev/null 2> $O/rocprof_score.err
fi
if has config5; then
  python tools/bench_config5.py > $O/config5_one_gpu_share.json 2> $O/config5.err; cat $O/config5_one_gpu_share.json
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_config5 -o c -- python3 tools/bench_config5.py --only bf16 > /dev/null 2> $O/rocprof_config5.err
fi
if has small; then
  python tools/small_times.py 2>&1 | grep -v amdgpu.ids > $O/small_times.txt; cat $O/small_times.txt
  python tools/verbose_small.py 2>&1 | grep -v amdgpu.ids > $O/verbose_small.txt; cat $O/verbose_small.txt   # fit(verbose=1) at the example's size
fi
