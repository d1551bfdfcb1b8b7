#!/bin/bash
# stage clocks of the tree at hg38 scale, repeat-rich and default genome (Ascan = stage A's per-chunk processing after the direct-table wait, A = the cooperative row expansion)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_h
mkdir -p $OUT
cd $R
for rep in 0.3 0; do
VARGENO_HIP_LIB=$R/variants/clk.so timeout 600 python3 bench.py --no-gather-probe --no-ingest --secondary none --sustain-seconds 0 --cpu-reference no --job-reads 0 --repeats $rep --steps 1 --warmup 0 --cpu-sample 0 > $OUT/clk_$rep.json 2> $OUT/clk_$rep.err
echo "== repeats $rep"; grep -h "^CLK.*ecap 14" $OUT/clk_$rep.json $OUT/clk_$rep.err | sort | head -10; grep -h "^DBG" $OUT/clk_$rep.json $OUT/clk_$rep.err | sort | tail -4
done
