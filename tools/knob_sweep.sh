# per-launch times of the x3 step under single-knob changes (tools/x3_stage_times.py prints the knobs it ran with)
cd $GRAFT_REPO_ROOT
for kv in "" KURBM_BF16_SPLIT=2 KURBM_BF16_SPLIT=3 KURBM_BF16_SPLIT=5 KURBM_BF16_SPLIT=8 KURBM_X3_STATS_MFAST=1 KURBM_X3_MFAST=0 KURBM_X3_TALL=0 \
          KURBM_REDUCE_TR=16 KURBM_REDUCE_TR=64 KURBM_X3_XCD2D=0 KURBM_X3_F8POS=0 KURBM_X3_BYTES=0 KURBM_X3_STATS_TALL=0; do
    env $kv timeout -k 10 120 python tools/x3_stage_times.py 2>/dev/null | tail -1
done
