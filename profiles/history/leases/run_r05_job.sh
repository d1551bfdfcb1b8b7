#!/bin/bash
# Round 5: the job leg alone, at a size passed on the command line (reads), with the host's memory / disk watched every 5 s.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_job
mkdir -p $OUT
cd $R
N=${1:-48000000}
( while true; do echo "$(date +%s) mem $(cat /sys/fs/cgroup/memory.current) root $(df --output=used -B1 / | tail -1) shm $(df --output=used -B1 /dev/shm | tail -1)"; sleep 5; done ) > $OUT/watch_$N.txt 2>&1 &
W=$!
( time timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 --secondary none --cpu-reference no --cpu-sample 200000 --no-gather-probe --sustain-seconds 0 --job-reads $N ) > $OUT/bench_$N.json 2> $OUT/bench_$N.err
kill $W
grep -h "job\|ingest" $OUT/bench_$N.err | cut -c1-400
python3 - $OUT/bench_$N.json <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("value %.4g ms/step %.3f frac %.3f open %.2f wall %.0f" % (j["value"], j["ms_per_step"], j["roofline"]["frac"], j["config"]["index_open_s"], j["bench_wall_s"]))
print("job", json.dumps(j.get("job"))[:2500])
print("ingest", json.dumps({k: (v.get("value") if isinstance(v, dict) else v) for k, v in (j.get("ingest_end_to_end") or {}).get("paths", {}).items()}), (j.get("ingest_end_to_end") or {}).get("chosen"))
PY
awk '{ if ($3 > m) m = $3; if ($5 > r) r = $5; if ($7 > s) s = $7 } END { printf "peak cgroup memory %.1f GB, root fs used %.1f GB, shm used %.1f GB\n", m / 1e9, r / 1e9, s / 1e9 }' $OUT/watch_$N.txt
