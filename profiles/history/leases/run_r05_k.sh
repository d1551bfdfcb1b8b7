#!/bin/bash
# Round 5, lease k: the command line's tests after its exit path changed, then the driver's command in full (all secondary legs, the
# job leg) under the memory / disk watch.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_k
mkdir -p $OUT
cd $R
( time timeout 900 python3 -m pytest tests/test_gpu_cli.py tests/test_gpu_fastq.py -m gpu -q -x -k "not hg38" ) > $OUT/pytest.txt 2>&1
tail -6 $OUT/pytest.txt
bash profiles/run_r05_stage.sh k
