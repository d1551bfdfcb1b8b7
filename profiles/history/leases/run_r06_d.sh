#!/bin/bash
# Round 6, lease d: the plan takes the widest tables the budget holds; chr22-scale under budgets; work-chunk sizes for 1 M-read batches.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_d
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -x -q -m gpu -k "device_memory_budget or four_full_replicas or fallback_layouts" > $OUT/tests.txt 2>&1
tail -3 $OUT/tests.txt
run() {
	timeout 600 python3 bench.py --workload chr22 --steps 40 --warmup 5 --secondary none --no-ingest --cpu-reference no --sustain-seconds 0 --no-gather-probe > $OUT/bench_$1.json 2> $OUT/bench_$1.err
	python3 - $OUT/bench_$1.json "$1" <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
d = json.load(open(j["detail"]))
print(sys.argv[2], "value %.4g ms/step %.4f kernel %.4f frac %.3f hbm %.2f GB parity %s | %s" % (j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"], j["roofline"]["frac"], j["config"]["index_bytes_hbm"] / 1e9, (j.get("parity") or {}).get("equal"), d["config"]["index_plan"][:400]))
PY
}
run whole
VG_MAX_DEVICE_BYTES=24000000000 run b24
VG_MAX_DEVICE_BYTES=12000000000 run b12
VG_MAX_DEVICE_BYTES=8000000000 run b8
VG_MAX_DEVICE_BYTES=6500000000 run b6_5
VG_WORK_CHUNK=32 run chunk32
VG_WORK_CHUNK=64 run chunk64
VG_WORK_CHUNK=256 run chunk256
