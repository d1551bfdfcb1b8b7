#!/bin/bash
# Round 6, lease t (final tree; host code changed since the one-call suite run, the kernels did not): the FASTQ / command-line / multi-rank
# GPU tests, then the driver's command without the child legs (main line, lowq50 / len101 / len250 on the open index, ingest, job, job_stream).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_t
mkdir -p $OUT
rm -rf /tmp/vg_bench /tmp/vg_bench_job /dev/shm/vg_bench* /tmp/pytest-of-* 2>/dev/null
cd $R
( while true; do echo "$(date +%s) mem $(cat /sys/fs/cgroup/memory.current 2>/dev/null) max $(cat /sys/fs/cgroup/memory.max 2>/dev/null) root $(df --output=used -B1 / | tail -1) shm $(df --output=used -B1 /dev/shm | tail -1)"; sleep 5; done ) > $OUT/watch.txt 2>&1 &
W=$!
( time timeout 700 python -m pytest tests/test_gpu_fastq.py tests/test_gpu_multi.py tests/test_gpu_cli.py -x -q -m gpu --durations=8 -k "not hg38_scale" ) > $OUT/tests.txt 2>&1
tail -14 $OUT/tests.txt
( time timeout 1100 python3 bench.py --gpus 1 --steps 20 --warmup 5 --secondary lowq50,len101,len250 --cpu-reference no ) > $OUT/bench.json 2> $OUT/bench.err
grep -E "^\[bench\]" $OUT/bench.err | tail -30 | cut -c1-300
wc -c $OUT/bench.json
cp /tmp/vg_bench/bench_detail_g3100000000_s10000000_c24.json $OUT/bench_detail.json 2>/dev/null
python3 - $OUT/bench.json <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("value %.4g ms/step %.3f frac %.3f kernel %.3f pack %.3f open %.2f wall %.0f traffic %s" % (j["value"], j["ms_per_step"], j["roofline"]["frac"], j["device_ms_per_step"]["wave"], j["device_ms_per_step"]["pack"], j["config"]["index_open_s"], j["bench_wall_s"], j["roofline"]["traffic"]))
    print("job", json.dumps(j.get("job")))
    print("job_stream", json.dumps(j.get("job_stream")))
    for k, v in (j.get("secondary") or {}).items():
        print(k, json.dumps(v))
    print("cpu", json.dumps(j.get("cpu_baseline"))[:400])
except Exception as e:
    print("no bench line: %r" % (e,))
PY
kill $W
awk '{ if ($3 > m) m = $3; if ($7 > r) r = $7; if ($9 > s) s = $9; mx = $5 } END { printf "peak cgroup memory %.1f GB of %.1f, root fs used %.1f GB, shm used %.1f GB\n", m / 1e9, mx / 1e9, r / 1e9, s / 1e9 }' $OUT/watch.txt
