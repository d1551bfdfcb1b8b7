#!/bin/bash
# hg38-scale experiment (BASELINE.json configs[2] shape): 3.1 Gbp in 24 sequences, 10 M SNPs, one 8 M-read batch of its 30x reads.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/hg38
df -h /tmp | tail -1
export VG_BENCH_DIR=/tmp/vg_bench_hg38
( time python3 $R/bench.py --workload hg38 --cpu-sample 8000000 --steps 10 --warmup 2 ) > $R/gpurun_out/hg38/bench_hg38.json 2> $R/gpurun_out/hg38/bench_hg38.err
tail -25 $R/gpurun_out/hg38/bench_hg38.err
cat $R/gpurun_out/hg38/bench_hg38.json
ls -la /tmp/vg_bench_hg38/*/ | head -12
