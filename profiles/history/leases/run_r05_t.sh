#!/bin/bash
# Round 5, lease t: what the driver runs at the end of the round, on the shipped library: the whole GPU suite, then smoke().
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_t
mkdir -p $OUT
cd $R
( time timeout 2400 python3 -m pytest tests/ -x -q -m gpu --durations=15 ) > $OUT/gpu_suite.txt 2>&1
tail -25 $OUT/gpu_suite.txt
( time python3 -c "import __graft_entry__ as g; g.smoke()" ) > $OUT/smoke.txt 2>&1
tail -4 $OUT/smoke.txt
