#!/bin/bash
# Round 4, lease G: reads that meet auxiliary rows set aside for a launch of their own (same-box A/B against one launch), parity.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_g
mkdir -p $OUT
cd $R
( time timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fastq.py -x -q -m gpu --durations=5 ) > $OUT/pytest.log 2>&1
tail -8 $OUT/pytest.log
run() {
	local name=$1; shift
	timeout 600 python3 bench.py --workload chr22 --cpu-reference no --no-gather-probe --no-ingest --secondary none --sustain-seconds 0 --steps 40 --warmup 5 "$@" > $OUT/$name.json 2> $OUT/$name.err
	python3 - $OUT/$name.json $name <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    d = j["device_ms_per_step"]
    o = j["other_input_form"]
    print("%-16s reads/s %.4g  ms/step %.3f  wave %.3f  pack %.3f  tiers %.3f  frac %.3f  spilled %s | gate words: %.4g  ms/step %.3f wave %.3f pack %.3f" % (sys.argv[2], j["value"], j["ms_per_step"], d["wave"], d["pack"], d["spill_tiers_overlapped"], j["roofline"]["frac"], j.get("reads_per_step_redone_by_deep_list_tier"), o["value"], o["ms_per_step"], o["wave_ms"], o["pack_ms"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
	grep -E "parity" $OUT/$name.err | tee -a $OUT/summary.txt
}
for rep in 0.3 0; do
	tag=$( [ $rep = 0 ] && echo def || echo rep30 )
	run ${tag}_new --repeats $rep
	VG_NO_HEAVY_PASS=1 run ${tag}_oneLaunch --repeats $rep --cpu-sample 0
	for v in r03 kt; do VARGENO_HIP_LIB=$R/variants/$v.so run ${tag}_$v --repeats $rep --cpu-sample 0; done
	run ${tag}_new2 --repeats $rep --cpu-sample 0
	VG_NO_HEAVY_PASS=1 run ${tag}_oneLaunch2 --repeats $rep --cpu-sample 0
	VARGENO_HIP_LIB=$R/variants/clk.so timeout 600 python3 bench.py --workload chr22 --repeats $rep --cpu-sample 0 --no-gather-probe --no-ingest --secondary none --sustain-seconds 0 --steps 1 --warmup 0 > $OUT/${tag}_clk.txt 2> $OUT/${tag}_clk.err
	grep "dbg" $OUT/${tag}_clk.err | tail -1 | tee -a $OUT/summary.txt
done
