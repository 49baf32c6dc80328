# A/B/C of the k loop: V2 = three B stages + opportunistic early reads (libkurbm.so), V1 = three B stages, reads behind the barrier
# (libkurbm_no_landed.so), old = two stages (libkurbm_old.so)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03n}
mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_configs_at_size.py -m gpu -q -x -k "config2 or config3" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
for rep in 1 2; do
  for lib in libkurbm.so libkurbm_no_landed.so libkurbm_old.so; do
    echo "== $lib"; KURBM_LIB=$R/keras_unsupervised_amd/csrc/$lib python tools/x3_stage_times.py 2>&1 | grep -v amdgpu.ids
  done
done > $O/ab.txt 2>&1; cat $O/ab.txt
