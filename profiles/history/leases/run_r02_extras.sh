#!/bin/bash
# Round-2 extras on one box lease: the final profile of the shipped build, the 50 % low-quality stress profile at hg38 and chr22
# scale (with the parity check), and the default bench with the reference binary timed beside it (cpu_baseline.reference_binary).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/extras_r02
mkdir -p $OUT
cd $R
bash profiles/run_prof_r02.sh r02 > $OUT/prof.log 2>&1
tail -3 $OUT/prof.log
( time python3 bench.py --lowq 0.5 --no-ingest --no-gather-probe ) > $OUT/bench_hg38_lowq50.json 2> $OUT/bench_hg38_lowq50.err
tail -3 $OUT/bench_hg38_lowq50.err
( time python3 bench.py --workload chr22 --lowq 0.5 --no-ingest --no-gather-probe --cpu-reference no ) > $OUT/bench_chr22_lowq50.json 2> $OUT/bench_chr22_lowq50.err
( time python3 bench.py --cpu-reference yes --no-ingest --no-gather-probe ) > $OUT/bench_hg38_with_reference.json 2> $OUT/bench_hg38_with_reference.err
tail -3 $OUT/bench_hg38_with_reference.err
