#!/bin/bash
# The index of BASELINE.json configs[4] (hg38 + ~100 M SNPs: 3.2 G SNP k-mers, 6.1 G k-mers in the two dictionaries) on ONE
# replica: index build, load, parity of the 8 M-read batch against the oracle, throughput on the layout without merged view /
# direct table.  ~100 GB of index files: the work directory goes to /dev/shm.   -> gpurun_out/hg38f/
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/hg38f
mkdir -p $OUT
cd $R
df -h /dev/shm | tail -1 > $OUT/run.log
export VG_BENCH_DIR=/dev/shm/vg_bench VARGENO_VERBOSE=1 VG_VERBOSE=1
( time timeout 2400 python3 bench.py --workload hg38f --no-ingest --no-gather-probe --steps 10 --warmup 2 ) > $OUT/bench_hg38f.json 2> $OUT/bench_hg38f.err
tail -25 $OUT/bench_hg38f.err
rm -rf /dev/shm/vg_bench
