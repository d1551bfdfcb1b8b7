#!/bin/bash
# Round 6: the whole GPU suite + smoke(), as the driver runs them (the driver allows 1 200 s).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_suite
mkdir -p $OUT
cd $R
( time timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=15 ) > $OUT/suite.txt 2>&1
tail -25 $OUT/suite.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
