# per-launch times of the real-valued-data and Gaussian-visible steps (rocprofv3 kernel stats)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03o}; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/real -o r -- python3 $R/tools/x3_profile_run.py x3 4096 784 1024 real > /dev/null 2> $O/real.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/gauss -o g -- python3 $R/tools/x3_profile_run.py x3 4096 784 1024 real gauss > /dev/null 2> $O/gauss.err
find $O -name "*kernel_stats.csv"
