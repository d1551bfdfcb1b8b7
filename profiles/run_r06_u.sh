#!/bin/bash
# Round 6, lease u (host code only): the once-only FASTQ route with the reader thread dealing the stream to copier threads through
# private pipes (splice moves page references, the copies run side by side): its GPU tests, the `job_stream` leg alone, and the
# same-box A/B of copier counts x feeds (profiles/pipe_ab.py).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_u
mkdir -p $OUT
rm -rf /tmp/vg_bench /tmp/vg_bench_job /dev/shm/vg_bench* /dev/shm/vg_keep.fq /tmp/pytest-of-* 2>/dev/null
cd $R
( while true; do echo "$(date +%s) mem $(cat /sys/fs/cgroup/memory.current 2>/dev/null) max $(cat /sys/fs/cgroup/memory.max 2>/dev/null) root $(df --output=used -B1 / | tail -1) shm $(df --output=used -B1 /dev/shm | tail -1)"; sleep 5; done ) > $OUT/watch.txt 2>&1 &
W=$!
( time timeout 500 python -m pytest tests/test_gpu_fastq.py -x -q -m gpu --durations=5 -k "not_a_regular_file or long_line or truncated_final" ) > $OUT/tests.txt 2>&1
tail -9 $OUT/tests.txt
( time VG_BENCH_KEEP_FASTQ=/dev/shm/vg_keep.fq timeout 1000 python3 bench.py --gpus 1 --steps 5 --warmup 2 --secondary none --no-gather-probe --no-ingest --cpu-reference no --sustain-seconds 0 --cpu-sample 0 --job-reads 8000000 --stream-reads 620000000 ) > $OUT/bench.json 2> $OUT/bench.err
grep -E "^\[bench\]" $OUT/bench.err | tail -4 | cut -c1-400
python3 - $OUT/bench.json <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("value %.4g ms/step %.3f kernel %.3f open %.2f" % (j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"], j["config"]["index_open_s"]))
    print("job_stream", json.dumps(j.get("job_stream")))
except Exception as e:
    print("no bench line: %r" % (e,))
PY
cp /tmp/vg_bench/bench_detail_g3100000000_s10000000_c24.json $OUT/bench_detail.json 2>/dev/null
D=/tmp/vg_bench/g3100000000_s10000000_c24
if [ -s /dev/shm/vg_keep.fq ] && [ -e $D/idx.done ]; then
	timeout 900 python3 profiles/pipe_ab.py $D /dev/shm/vg_keep.fq 24 \
		lend_cop0:LEND=1,VARGENO_PIPE_COPIERS=0 lend_cop2:LEND=1,VARGENO_PIPE_COPIERS=2 lend_cop4:LEND=1 lend_cop8:LEND=1,VARGENO_PIPE_COPIERS=8 \
		write_cop0:LEND=0,VARGENO_PIPE_COPIERS=0 write_cop4:LEND=0 lend_cop4_again:LEND=1 lend_cop6:LEND=1,VARGENO_PIPE_COPIERS=6 > $OUT/pipe_ab.jsonl 2> $OUT/pipe_ab.err
	cat $OUT/pipe_ab.jsonl | cut -c1-760
	tail -3 $OUT/pipe_ab.err
fi
rm -f /dev/shm/vg_keep.fq
kill $W
awk '{ if ($3 > m) m = $3; if ($7 > r) r = $7; if ($9 > s) s = $9; mx = $5 } END { printf "peak cgroup memory %.1f GB of %.1f, root fs used %.1f GB, shm used %.1f GB\n", m / 1e9, mx / 1e9, r / 1e9, s / 1e9 }' $OUT/watch.txt
