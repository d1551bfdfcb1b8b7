#!/bin/bash
# development loop: GPU tests without the hg38-scale ones, then the default (hg38-scale) bench with phase timings
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
( time python -m pytest tests -x -q -m gpu -k "not hg38" ) > gpurun_out/r2d/pytest.log 2>&1
grep -E "passed|failed" gpurun_out/r2d/pytest.log | tail -2
( time VG_VERBOSE=1 VARGENO_VERBOSE=1 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r2d/bench_hg38.json 2> gpurun_out/r2d/bench_hg38.err
grep -vE "^\[vargeno index\]" gpurun_out/r2d/bench_hg38.err | tail -25
