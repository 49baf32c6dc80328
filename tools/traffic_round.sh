# FETCH_SIZE / WRITE_SIZE passes over 60 x3 steps -> gpurun_out/$1/{pmc_fetch,pmc_write}
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 tools/x3_profile_run.py x3 > /dev/null 2> $O/rocprof2.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 tools/x3_profile_run.py x3 > /dev/null 2> $O/rocprof3.err
