#!/bin/bash
# Round 4, first lease: the key-table build of the wave kernel -- parity (GPU suite without the hg38-scale tests) and a same-box A/B
# against the round-3 library on chr22-scale indexes (default genome and repeat-rich genome).
#   bash profiles/run_r04_a.sh   -> gpurun_out/r04_a/
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_a
mkdir -p $OUT
cd $R
( time timeout 900 python3 -m pytest tests -x -q -m gpu -k "not hg38" ) > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
run() {   # name, bench args..., env through VARGENO_HIP_LIB set by caller
	local name=$1; shift
	timeout 600 python3 bench.py --workload chr22 --cpu-reference no --no-gather-probe --no-ingest --steps 40 --warmup 5 "$@" > $OUT/$name.json 2> $OUT/$name.err
	python3 - $OUT/$name.json $name <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    d = j["device_ms_per_step"]
    print("%-16s reads/s %.4g  ms/step %.3f  wave %.3f  pack %.3f  tiers %.3f (deep %.3f)  frac %.3f  spilled %s  lane tier %s" % (sys.argv[2], j["value"], j["ms_per_step"], d["wave"], d["pack"], d["spill_tiers_overlapped"], d["of_which_deep_list_wave_tier"], j["roofline"]["frac"], j.get("reads_per_step_redone_by_deep_list_tier"), j.get("reads_per_step_sent_on_to_lane_tier")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
	grep -E "parity" $OUT/$name.err | tee -a $OUT/summary.txt
}
for rep in 0 0.3; do
	tag=$( [ $rep = 0 ] && echo def || echo rep30 )
	run ${tag}_new --repeats $rep
	VARGENO_HIP_LIB=$R/variants/r03.so run ${tag}_r03 --repeats $rep --cpu-sample 0
	VARGENO_HIP_LIB=$R/variants/e16n4.so run ${tag}_e16n4 --repeats $rep --cpu-sample 0
	run ${tag}_new2 --repeats $rep --cpu-sample 0
	VARGENO_HIP_LIB=$R/variants/clk.so timeout 600 python3 bench.py --workload chr22 --repeats $rep --cpu-sample 0 --no-gather-probe --no-ingest --steps 1 --warmup 0 > $OUT/${tag}_clk.txt 2> $OUT/${tag}_clk.err
	grep "dbg" $OUT/${tag}_clk.err | tail -1 | tee -a $OUT/summary.txt
done
