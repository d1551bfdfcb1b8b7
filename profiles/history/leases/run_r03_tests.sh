#!/bin/bash
# Round 3: the GPU suite without the hg38-scale tests (index builds of minutes), after the loader / boundary changes.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_tests
mkdir -p $OUT
cd $R
( time python3 -m pytest tests -x -q -m gpu -k "not hg38" ) > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
