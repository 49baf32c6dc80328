# SQ / TCC counter passes over 60 x3 steps (each pass its own run, --pmc only) -> gpurun_out/$1/sq_*
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $O/sq_a -o a -- python3 tools/x3_profile_run.py x3 > /dev/null 2> $O/sq_a.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/sq_b -o b -- python3 tools/x3_profile_run.py x3 > /dev/null 2> $O/sq_b.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum --output-format csv -d $O/sq_c -o c -- python3 tools/x3_profile_run.py x3 > /dev/null 2> $O/sq_c.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $O/sq_d -o d -- python3 tools/x3_profile_run.py x3 > /dev/null 2> $O/sq_d.err || echo "pass d (MFMA busy) not available"
find $O -name "*counter_collection.csv"
