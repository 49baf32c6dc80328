# config 5's one-GPU share (rounded bf16, 4096 x 4096, PCD-10, 1024 rows) for ~4 s with rocm-smi sampled beside it: is it at the
# board's power cap?   bash tools/config5_power.sh OUTDIR
O=${1:-gpurun_out/c5power}; mkdir -p $O
python tools/bench_config5.py --only bf16 --steps 4000 > $O/config5_long.json 2> $O/config5_long.err &
P=$!
sleep 1.5
for i in 1 2 3 4 5 6 7 8; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" ; sleep 0.25; done > $O/smi.txt
wait $P
cat $O/config5_long.json
sort $O/smi.txt | uniq -c | sort -rn | head -8
