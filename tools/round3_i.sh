# round 3: the deep pipeline for one-piece byte tiles (config 5's half steps): parity first, then A/B against the old schedule
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03v}
mkdir -p $O
cd $R
C=$R/keras_unsupervised_amd/csrc
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "bf16 or config5 or eight_shards" > $O/tests_bf16.txt 2>&1 || { tail -30 $O/tests_bf16.txt; exit 1; }
tail -3 $O/tests_bf16.txt
for rep in 1 2; do
  for lib in ${LIBS:-libkurbm.so libkurbm_deep0.so}; do
    echo "== $lib" >> $O/ab.txt
    KURBM_LIB=$C/$lib timeout -k 10 300 python tools/bench_config5.py --only bf16 >> $O/ab.txt 2>> $O/ab.err || exit 1
  done
done
cat $O/ab.txt
KURBM_LIB=$C/libkurbm_stamps.so timeout -k 10 300 python tools/stamp_c5.py > $O/stamps_c5.txt 2>&1; sed -n 3,6p $O/stamps_c5.txt
