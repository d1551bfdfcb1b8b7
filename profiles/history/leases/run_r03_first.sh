#!/bin/bash
# Round 3, first lease: the line-granularity probe, host facts, the new GPU tests, the default bench (with the reference binary beside it).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_first
mkdir -p $OUT
cd $R
( free -g; nproc; df -h /tmp /dev/shm | cat ) > $OUT/host.txt 2>&1
bash profiles/run_line_probe_r03.sh > $OUT/line_probe.log 2>&1
( time python3 -m pytest tests -x -q -m gpu -k "empty_stream or corrupt or device_framing" ) > $OUT/pytest_new.log 2>&1
tail -3 $OUT/pytest_new.log
( time python3 bench.py --steps 20 --warmup 5 ) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -12 $OUT/bench_default.err
cat $OUT/host.txt
