# A/B: the kernel-argument warm-up at kernel entry (libkurbm.so) against the build without it
# (make -C keras_unsupervised_amd/csrc variant VAR="-DKURBM_WARM_ARGS=0 -DKURBM_PRIO_ENTRY=0" OUT=libkurbm_nowarm.so)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03z}; mkdir -p $O; cd $R
C=$R/keras_unsupervised_amd/csrc
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
for rep in 1 2; do
  for lib in libkurbm.so libkurbm_nowarm.so; do
    echo "== $lib" >> $O/warm_ab.txt
    KURBM_LIB=$C/$lib timeout -k 10 300 python bench.py --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench.py steps/s', d['value'], 'ms', d['ms_per_step'])" >> $O/warm_ab.txt
    KURBM_LIB=$C/$lib timeout -k 10 300 python tools/bench_config5.py --only bf16 2>/dev/null | cut -c60-200 >> $O/warm_ab.txt
  done
done
cat $O/warm_ab.txt
KURBM_LIB=$C/libkurbm_stamps.so timeout -k 10 300 python tools/stamp_x3.py > $O/stamps_x3_warm.txt 2>&1; cat $O/stamps_x3_warm.txt
