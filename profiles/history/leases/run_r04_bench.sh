#!/bin/bash
# the driver's command on the shipped build, once more after the counter passes of the closing lease were committed (so that the line quotes them)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_bench
mkdir -p $OUT
cd $R
( time timeout 1750 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -8 $OUT/bench_default.err | cut -c1-200
