#!/bin/bash
# Round 6, lease a: the once-only FASTQ route, the late store (lane tier once per synchronisation), the compact bench line.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_a
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fastq.py -x -q -m gpu -k "deep_tier_leaves or all_tiers or wave_tier or not_a_regular_file or truncated_final or read_store or ftiny_counts" > $OUT/tests.txt 2>&1
tail -5 $OUT/tests.txt
timeout 600 python3 bench.py --workload chr22 --steps 40 --warmup 5 --secondary none > $OUT/bench_chr22.json 2> $OUT/bench_chr22.err
tail -3 $OUT/bench_chr22.err | cut -c1-300; wc -c $OUT/bench_chr22.json
timeout 1200 python3 bench.py --steps 20 --warmup 5 --secondary len250 --job-reads 0 --no-ingest --cpu-reference no --sustain-seconds 0 > $OUT/bench_len250.json 2> $OUT/bench_len250.err
grep -E "secondary|parity|sustained|index resident" $OUT/bench_len250.err | cut -c1-300; wc -c $OUT/bench_len250.json
cp /tmp/vg_bench/bench_detail_*.json $OUT/ 2>/dev/null
