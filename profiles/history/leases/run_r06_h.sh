#!/bin/bash
# Round 6, lease h: stage clocks + stage-B census (variants/clk.so = -DVG_STAGE_CLOCKS) of the shipped tree, default and repeat-rich hg38-scale genome, one 8 M-read step each.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_h
mkdir -p $OUT
cd $R
for g in default repeats30; do
	EXTRA=""; [ $g = repeats30 ] && EXTRA="--repeats 0.3"
	VARGENO_HIP_LIB=$R/variants/clk.so timeout 1500 python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-gather-probe --no-ingest --secondary none --sustain-seconds 0 --job-reads 0 --no-pretouch $EXTRA > $OUT/clk_$g.txt 2> $OUT/clk_$g.err
	echo "== $g"; grep -E "^CLK" $OUT/clk_$g.txt | head -60 | awk '{for(i=1;i<=NF;i++) if ($i ~ /^(iters|refill|A|B0|B1|vote|walk|wload|wloop|watom|Akmer|Adx|Ascan)$/) {s[$i]+=$(i+1)}; n++} END {printf "mean of %d sampled waves:", n; for (k in s) printf " %s %.0f", k, s[k]/n; printf "\n"}'
	grep -E "^DBG" $OUT/clk_$g.txt | awk '{for(i=1;i<=NF;i++) if ($i ~ /^(iters|pairs|items|rounds|dualq|large|secbad)$/) {s[$i]+=$(i+1)}; n++} END {printf "mean of %d sampled waves:", n; for (k in s) printf " %s %.1f", k, s[k]/n; printf "\n"}'
	grep "dbg\] list overflows" $OUT/clk_$g.err | tail -2
done
