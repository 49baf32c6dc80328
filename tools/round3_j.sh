# round 3: the entry of a launch -- parity, bench.py / config 5 A/B between builds, stamps
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03j}; mkdir -p $O; cd $R
C=$R/keras_unsupervised_amd/csrc
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
for rep in 1 2; do
  for lib in ${LIBS:-libkurbm.so}; do
    echo "== $lib" >> $O/ab.txt
    KURBM_LIB=$C/$lib timeout -k 10 300 python bench.py --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench.py steps/s', d['value'], 'ms', d['ms_per_step'])" >> $O/ab.txt
    KURBM_LIB=$C/$lib timeout -k 10 300 python tools/bench_config5.py --only bf16 2>/dev/null | cut -c60-200 >> $O/ab.txt
  done
done
cat $O/ab.txt
KURBM_LIB=$C/libkurbm_stamps.so timeout -k 10 300 python tools/stamp_x3.py > $O/stamps_x3.txt 2>&1; grep -v "wave [145]" $O/stamps_x3.txt
