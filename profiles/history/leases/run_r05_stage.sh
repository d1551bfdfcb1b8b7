#!/bin/bash
# Round 5: the driver's command in stages, the host's memory / disk watched every 5 s (a container that fills its 300 GiB memory limit or its 79 GB root is killed).
#   bash profiles/run_r05_stage.sh <tag> [bench.py arguments]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/r05_stage_$TAG
mkdir -p $OUT
cd $R
( while true; do echo "$(date +%s) mem $(cat /sys/fs/cgroup/memory.current) root $(df --output=used -B1 / | tail -1) shm $(df --output=used -B1 /dev/shm | tail -1)"; sleep 5; done ) > $OUT/watch.txt 2>&1 &
W=$!
( time timeout 2400 python3 bench.py --gpus 1 --steps 20 --warmup 5 "$@" ) > $OUT/bench.json 2> $OUT/bench.err
kill $W
tail -45 $OUT/bench.err | cut -c1-260
python3 - $OUT/bench.json <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("value %.4g ms/step %.3f frac %.3f kernel %.3f pack %.3f open %.2f wall %.0f" % (j["value"], j["ms_per_step"], j["roofline"]["frac"], j["device_ms_per_step"]["wave"], j["device_ms_per_step"]["pack"], j["config"]["index_open_s"], j["bench_wall_s"]))
print("job", json.dumps({k: v for k, v in (j.get("job") or {}).items() if k != "index_open_phases"})[:1800])
for k, v in (j.get("secondary") or {}).items():
    print(k, v.get("skipped") or "%.4g reads/s ms/step %.3f frac %.3f parity %s wall %.0f open %s" % (v["value"], v["ms_per_step"], v["roofline"]["frac"], (v.get("parity") or {}).get("equal"), v.get("wall_s", 0), v.get("index_open_s")))
print("ingest", json.dumps({k: (v.get("value") if isinstance(v, dict) else v) for k, v in (j.get("ingest_end_to_end") or {}).get("paths", {}).items()}), (j.get("ingest_end_to_end") or {}).get("chosen"))
print("cpu", (j.get("cpu_baseline") or {}).get("value"), (j.get("cpu_baseline") or {}).get("kind"))
PY
awk '{ if ($3 > m) m = $3; if ($5 > r) r = $5; if ($7 > s) s = $7 } END { printf "peak cgroup memory %.1f GB, root fs used %.1f GB, shm used %.1f GB\n", m / 1e9, r / 1e9, s / 1e9 }' $OUT/watch.txt
