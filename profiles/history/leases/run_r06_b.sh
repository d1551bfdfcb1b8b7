#!/bin/bash
# Round 6, lease b: tables that scale with the index -- parity on every layout knob, then chr22-scale before / after (VG_DX_BITS=32 VG_REF_JG_BITS=32 = the r05 layout).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_b
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fallback_layouts or deep_tier_leaves or device_memory_budget or self_complementary or dense_snp or ftiny_counts or all_tiers" > $OUT/tests.txt 2>&1
tail -8 $OUT/tests.txt
for v in new old; do
	if [ $v = old ]; then export VG_DX_BITS=32 VG_REF_JG_BITS=32; fi
	timeout 600 python3 bench.py --workload chr22 --steps 40 --warmup 5 --secondary none --no-ingest --cpu-reference no --sustain-seconds 1 > $OUT/bench_chr22_$v.json 2> $OUT/bench_chr22_$v.err
	python3 - $OUT/bench_chr22_$v.json $v <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[2], "value %.4g ms/step %.4f kernel %.4f pack %.4f frac %.3f hbm %.2f GB open %.2f s parity %s" % (j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"], j["device_ms_per_step"]["pack"], j["roofline"]["frac"], j["config"]["index_bytes_hbm"] / 1e9, j["config"]["index_open_s"], (j.get("parity") or {}).get("equal")))
PY
done
unset VG_DX_BITS VG_REF_JG_BITS
cp /tmp/vg_bench/bench_detail_*.json $OUT/ 2>/dev/null
