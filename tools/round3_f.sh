# landed-flag hand-off: parity (the tests that touch the x3 / bf16 kernels) and A/B timing against the build without it
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03l}
mkdir -p $O
cd $R
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/tests.txt 2>&1; tail -4 $O/tests.txt
for rep in 1 2; do
  for lib in libkurbm.so libkurbm_no_landed.so; do
    echo "== $lib"; KURBM_LIB=$R/keras_unsupervised_amd/csrc/$lib python tools/x3_stage_times.py 2>&1 | grep -v amdgpu.ids
  done
done > $O/ab.txt 2>&1; cat $O/ab.txt
