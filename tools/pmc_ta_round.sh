# texture-address / L1 path counters of the x3 kernels over 60 steps (each pass its own run, --pmc only) -> gpurun_out/$1/ta_*
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
rocprofv3 --pmc GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr TA_BUFFER_WAVEFRONTS_sum --output-format csv -d $O/ta_a -o a -- python3 tools/x3_profile_run.py x3 > /dev/null 2> $O/ta_a.err || echo "pass a failed"
rocprofv3 --pmc TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $O/ta_b -o b -- python3 tools/x3_profile_run.py x3 > /dev/null 2> $O/ta_b.err || echo "pass b failed"
rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum --output-format csv -d $O/ta_c -o c -- python3 tools/x3_profile_run.py x3 > /dev/null 2> $O/ta_c.err || echo "pass c failed"
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $O/ta_d -o d -- python3 tools/x3_profile_run.py x3 > /dev/null 2> $O/ta_d.err || echo "pass d failed"
find $O -name "*counter_collection.csv" | grep ta_
