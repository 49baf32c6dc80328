set -e
for e in "X=0" "KURBM_X3_ABLATE=2" "KURBM_X3_ABLATE=4" "KURBM_X3_ABLATE=8" "KURBM_X3_ABLATE=6" "KURBM_X3_ABLATE=14"; do
  env $e timeout -k 10 100 python tools/x3_times.py
done
