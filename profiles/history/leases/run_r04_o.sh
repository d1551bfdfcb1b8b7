#!/bin/bash
# quick: stage clocks incl. the row loops of stage A, default vs repeat-rich chr22-scale genomes (clk2 build, out of tree)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_o
mkdir -p $OUT
cd $R
for rep in 0.3 0; do
	tag=$( [ $rep = 0 ] && echo def || echo rep30 )
	VARGENO_HIP_LIB=$R/variants/clk2.so timeout 600 python3 bench.py --workload chr22 --repeats $rep --cpu-sample 0 --no-gather-probe --no-ingest --secondary none --sustain-seconds 0 --steps 1 --warmup 0 > $OUT/${tag}_clk.txt 2> $OUT/${tag}_clk.err
	grep -c CLK $OUT/${tag}_clk.txt
done
