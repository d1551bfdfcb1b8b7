#!/bin/bash
# Round 6, lease c: which of the two scaled tables costs the chr22-scale kernel its time?  2 x 2: direct table 2^27 / 2^32 buckets x reference jump table 2^26 / 2^32 entries.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_c
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "device_memory_budget" > $OUT/tests.txt 2>&1
tail -3 $OUT/tests.txt
for v in "nat nat" "32 nat" "nat 32" "32 32" "28 nat" "26 nat"; do
	set -- $v
	unset VG_DX_BITS VG_REF_JG_BITS
	[ $1 != nat ] && export VG_DX_BITS=$1
	[ $2 != nat ] && export VG_REF_JG_BITS=$2
	timeout 600 python3 bench.py --workload chr22 --steps 40 --warmup 5 --secondary none --no-ingest --cpu-reference no --sustain-seconds 0 --no-gather-probe > $OUT/bench_$1_$2.json 2> $OUT/bench_$1_$2.err
	python3 - $OUT/bench_$1_$2.json "dx $1 jg $2" <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[2], "value %.4g ms/step %.4f kernel %.4f pack %.4f frac %.3f hbm %.2f GB open %.2f s parity %s" % (j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"], j["device_ms_per_step"]["pack"], j["roofline"]["frac"], j["config"]["index_bytes_hbm"] / 1e9, j["config"]["index_open_s"], (j.get("parity") or {}).get("equal")))
PY
done
