#!/bin/bash
# Round 6: the driver's command, the host's memory / disk watched every 5 s (a container that fills its 300 GiB memory limit or its 79 GB root is killed).
#   bash profiles/run_r06_stage.sh <tag> [bench.py arguments]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/r06_stage_$TAG
mkdir -p $OUT
# (the pool hands the same box to consecutive leases: what earlier ones left behind goes first -- a 79 GB root fills up)
rm -rf /tmp/vg_bench /tmp/vg_bench_job /dev/shm/vg_bench* /tmp/pytest-of-* 2>/dev/null
cat /sys/fs/cgroup/memory.stat 2>/dev/null | grep -E "^(anon|file|shmem|kernel|slab) " | tr '\n' ' '; echo
cd $R
( while true; do echo "$(date +%s) mem $(cat /sys/fs/cgroup/memory.current) root $(df --output=used -B1 / | tail -1) shm $(df --output=used -B1 /dev/shm | tail -1)"; sleep 5; done ) > $OUT/watch.txt 2>&1 &
W=$!
( time timeout 2400 python3 bench.py --gpus 1 --steps 20 --warmup 5 "$@" ) > $OUT/bench.json 2> $OUT/bench.err
kill $W
grep -E "^\[bench\]" $OUT/bench.err | tail -60 | cut -c1-250
wc -c $OUT/bench.json
cp /tmp/vg_bench/bench_detail_g3100000000_s10000000_c24.json $OUT/bench_detail.json 2>/dev/null
python3 - $OUT/bench.json <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("value %.4g ms/step %.3f frac %.3f kernel %.3f pack %.3f open %.2f wall %.0f traffic %s" % (j["value"], j["ms_per_step"], j["roofline"]["frac"], j["device_ms_per_step"]["wave"], j["device_ms_per_step"]["pack"], j["config"]["index_open_s"], j["bench_wall_s"], j["roofline"]["traffic"]))
print("job", json.dumps(j.get("job")))
print("job_stream", json.dumps(j.get("job_stream")))
for k, v in (j.get("secondary") or {}).items():
    print(k, json.dumps(v))
print("cpu", json.dumps(j.get("cpu_baseline"))[:400])
PY
awk '{ if ($3 > m) m = $3; if ($5 > r) r = $5; if ($7 > s) s = $7 } END { printf "peak cgroup memory %.1f GB, root fs used %.1f GB, shm used %.1f GB\n", m / 1e9, r / 1e9, s / 1e9 }' $OUT/watch.txt
