#!/bin/bash
# quick: stage-B census on default vs repeat-rich chr22-scale genomes (clk build)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_e
mkdir -p $OUT
cd $R
for rep in 0.3 0; do
	tag=$( [ $rep = 0 ] && echo def || echo rep30 )
	VARGENO_HIP_LIB=$R/variants/clk.so timeout 600 python3 bench.py --workload chr22 --repeats $rep --cpu-sample 0 --no-gather-probe --no-ingest --secondary none --sustain-seconds 0 --steps 1 --warmup 0 > $OUT/${tag}_clk.txt 2> $OUT/${tag}_clk.err
	grep DBG $OUT/${tag}_clk.txt | head -12
done
