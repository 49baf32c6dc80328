# A/B of a variant build against the shipped one: all GPU tests on the SHIPPED build, then bench.py and per-launch times of both
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03k}; mkdir -p $O; cd $R
C=$R/keras_unsupervised_amd/csrc
V=${VARIANT:-libkurbm_var.so}
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -1 $O/tests.txt
# (TESTVAR=1: the parity tests of the x3 path on the VARIANT too)
[ -n "$TESTVAR" ] && { KURBM_LIB=$C/$V timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "x3 or config2 or score or stats or fused" > $O/tests_variant.txt 2>&1 || { tail -30 $O/tests_variant.txt; exit 1; }; tail -1 $O/tests_variant.txt; }
for rep in 1 2 3; do
  for lib in libkurbm.so $V; do
    echo "== $lib" >> $O/ab.txt
    KURBM_LIB=$C/$lib timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-variants 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']; print('bench.py steps/s %.0f' % d['value'], ' '.join('%s %.1f' % (n[3:], 1e3*v['ms']) for n,v in k.items()))" >> $O/ab.txt
  done
done
cat $O/ab.txt
[ -n "$STAMPS" ] && { KURBM_LIB=$C/libkurbm_stamps.so timeout -k 10 300 python tools/stamp_x3.py 2>&1 | grep -v "wave [145]\|since the\|amdgpu.ids"; }
