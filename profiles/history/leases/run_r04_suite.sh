#!/bin/bash
# the whole GPU suite + smoke(), as the driver runs them
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_suite
mkdir -p $OUT
cd $R
( time timeout 1750 python3 -m pytest tests -x -q -m gpu --durations=12 ) > $OUT/pytest.log 2>&1
tail -22 $OUT/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
