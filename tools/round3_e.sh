# does the k loop of a real-valued A operand follow the staged bytes or the number of tiles?  (timing-only build ablate8)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03i}
mkdir -p $O
cd $R
for lib in libkurbm.so libkurbm_ablate8.so; do
  export KURBM_LIB=$R/keras_unsupervised_amd/csrc/$lib
  for v in "real bern" "real gauss"; do
    set -- $v
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/${lib}_$1_$2 -o s -- python3 tools/x3_profile_run.py x3 4096 784 1024 $1 $2 > /dev/null 2> $O/err_${lib}_$1_$2.txt
    echo "== $lib $1 $2"; head -6 $O/${lib}_$1_$2/s_kernel_stats.csv | cut -d, -f1-4 | cut -c1-140
  done
done
