#!/bin/bash
# Round 6, lease q: the rank-block walk for reads of five to eight chunks -- parity (fixtures with 224 / 250 bp reads, every tier), then 250 bp and default workloads against the
# library as profiled so far (variants/ship_ae99.so), same box.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_q
mkdir -p $OUT
rm -rf /tmp/vg_bench_job /tmp/pytest-of-* 2>/dev/null
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fastq.py -x -q -m gpu -k "not not_a_regular_file and not read_store_that_fills" > $OUT/tests.txt 2>&1
tail -3 $OUT/tests.txt
one() {
	timeout 900 python3 bench.py --steps 20 --warmup 5 --secondary none --job-reads 0 --no-ingest --cpu-sample $3 --sustain-seconds 0 --no-gather-probe --no-pretouch --cpu-reference no $2 > $OUT/$1.json 2> $OUT/$1.err
	python3 - $OUT/$1.json $1 <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("%-26s ms/step %.3f kernel %.3f pack %.3f parity %s" % (sys.argv[2], j["ms_per_step"], j["roofline"]["kernel_ms"], j["device_ms_per_step"]["pack"], (j.get("parity") or {}).get("equal")))
PY
}
one tree_len250_parity "--read-len 250" 2000000
for rep in 1 2; do
	VARGENO_HIP_LIB=$R/variants/ship_ae99.so one ae99_len250_$rep "--read-len 250" 0
	one tree_len250_$rep "--read-len 250" 0
done
for rep in 1 2; do
	VARGENO_HIP_LIB=$R/variants/ship_ae99.so one ae99_default_$rep "" 0
	one tree_default_$rep "" 0
done
VARGENO_HIP_LIB=$R/variants/ship_ae99.so one ae99_len200 "--read-len 200" 0
one tree_len200 "--read-len 200" 0
