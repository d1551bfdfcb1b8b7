#!/bin/bash
# Round 6: the GPU suite as the driver runs it (the driver allows 1 200 s), the container's memory and the root file system watched every 5 s.
#   bash profiles/run_r06_suite.sh [pytest selection arguments]       (default: the whole suite)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_suite
mkdir -p $OUT
# (the pool hands the same box to consecutive leases: what earlier ones left behind goes first -- a 79 GB root fills up)
rm -rf /tmp/vg_bench /tmp/vg_bench_job /dev/shm/vg_bench* /tmp/pytest-of-* 2>/dev/null
cd $R
( while true; do echo "$(date +%s) mem $(cat /sys/fs/cgroup/memory.current 2>/dev/null) max $(cat /sys/fs/cgroup/memory.max 2>/dev/null) root $(df --output=used -B1 / | tail -1) shm $(df --output=used -B1 /dev/shm | tail -1)"; sleep 5; done ) > $OUT/watch.txt 2>&1 &
W=$!
if [ $# -eq 0 ]; then set -- tests/; fi
( time timeout 1500 python -m pytest "$@" -x -q -m gpu --durations=15 ) > $OUT/suite.txt 2>&1
kill $W
tail -28 $OUT/suite.txt
awk '{ if ($3 > m) m = $3; if ($7 > r) r = $7; if ($9 > s) s = $9; mx = $5 } END { printf "peak cgroup memory %.1f GB of %.1f, root fs used %.1f GB, shm used %.1f GB\n", m / 1e9, mx / 1e9, r / 1e9, s / 1e9 }' $OUT/watch.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
