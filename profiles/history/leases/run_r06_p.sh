#!/bin/bash
# Round 6, lease p: stage A with the second entries of ALL FOUR chunks' buckets in one wait (variants/a4.so: 128 VGPRs) against the shipped library (two chunks per wait, 116 VGPRs), same box.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_p
mkdir -p $OUT
rm -rf /tmp/vg_bench_job /tmp/pytest-of-* 2>/dev/null
cd $R
one() {
	timeout 900 python3 bench.py --steps 20 --warmup 5 --secondary none --job-reads 0 --no-ingest --cpu-sample 0 --sustain-seconds 0 --no-gather-probe --no-pretouch $2 > $OUT/$1.json 2> $OUT/$1.err
	python3 - $OUT/$1.json $1 <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("%-26s ms/step %.3f kernel %.3f pack %.3f" % (sys.argv[2], j["ms_per_step"], j["roofline"]["kernel_ms"], j["device_ms_per_step"]["pack"]))
PY
}
for g in "" "--lowq 0.5" "--repeats 0.3"; do
	tag=default; [ "$g" = "--repeats 0.3" ] && tag=repeats30; [ "$g" = "--lowq 0.5" ] && tag=lowq50
	for rep in 1 2; do
		one ship_${tag}_$rep "$g"
		VARGENO_HIP_LIB=$R/variants/a4.so one a4_${tag}_$rep "$g"
	done
done
