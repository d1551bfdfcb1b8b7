#!/bin/bash
# Round 5: after bench.py's last changes (files read once before the open, fetch in the warm-up, host_enqueue_ms_per_step): the GPU
# tests that run bench.py, then its main line alone.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_last
mkdir -p $OUT
cd $R
( time timeout 900 python3 -m pytest tests/test_gpu_multi.py -m gpu -q -x ) > $OUT/pytest.txt 2>&1
grep "passed\|failed\|Error" $OUT/pytest.txt | tail -3
bash profiles/run_r05_stage.sh last --secondary none --cpu-reference no --no-ingest --job-reads 0 --sustain-seconds 0
grep "read once\|taken once\|index resident" $R/gpurun_out/r05_stage_last/bench.err | cut -c1-200
