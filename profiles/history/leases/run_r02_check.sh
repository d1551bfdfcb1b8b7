#!/bin/bash
# chr22-scale bench + the GPU test-suite without the hg38-scale tests (development loop)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
( time python -m pytest tests -x -q -m gpu -k "not hg38" ) > gpurun_out/r2c/pytest.log 2>&1
tail -5 gpurun_out/r2c/pytest.log
( time python bench.py --workload chr22 --steps 20 --warmup 3 --cpu-reference no ) > gpurun_out/r2c/bench_chr22.json 2> gpurun_out/r2c/bench_chr22.err
tail -4 gpurun_out/r2c/bench_chr22.err
