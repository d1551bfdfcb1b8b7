#!/bin/bash
# Round 4, second lease: whole-row fetches + host-packed FASTQ ingest -- parity of the FASTQ / parity test files, same-box A/B of
# capacity variants on the repeat-rich chr22-scale genome, and a functional run of the new bench.py (secondary legs, both ingest paths).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_b
mkdir -p $OUT
cd $R
( time timeout 1200 python3 -m pytest tests/test_gpu_fastq.py tests/test_gpu_parity.py -x -q -m gpu --durations=8 ) > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
run() {
	local name=$1; shift
	timeout 600 python3 bench.py --workload chr22 --cpu-reference no --no-gather-probe --no-ingest --secondary none --sustain-seconds 0 --steps 40 --warmup 5 "$@" > $OUT/$name.json 2> $OUT/$name.err
	python3 - $OUT/$name.json $name <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    d = j["device_ms_per_step"]
    print("%-16s reads/s %.4g  ms/step %.3f  wave %.3f  pack %.3f  tiers %.3f (deep %.3f)  frac %.3f  spilled %s  lane tier %s  other form %.4g" % (sys.argv[2], j["value"], j["ms_per_step"], d["wave"], d["pack"], d["spill_tiers_overlapped"], d["of_which_deep_list_wave_tier"], j["roofline"]["frac"], j.get("reads_per_step_redone_by_deep_list_tier"), j.get("reads_per_step_sent_on_to_lane_tier"), j["other_input_form"]["value"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
	grep -E "parity" $OUT/$name.err | tee -a $OUT/summary.txt
}
for rep in 0.3 0; do
	tag=$( [ $rep = 0 ] && echo def || echo rep30 )
	run ${tag}_new --repeats $rep
	for v in r03 e12n8 e10n10; do VARGENO_HIP_LIB=$R/variants/$v.so run ${tag}_$v --repeats $rep --cpu-sample 0; done
	run ${tag}_new2 --repeats $rep --cpu-sample 0
	VARGENO_HIP_LIB=$R/variants/clk.so timeout 600 python3 bench.py --workload chr22 --repeats $rep --cpu-sample 0 --no-gather-probe --no-ingest --secondary none --sustain-seconds 0 --steps 1 --warmup 0 > $OUT/${tag}_clk.txt 2> $OUT/${tag}_clk.err
	grep "dbg" $OUT/${tag}_clk.err | tail -1 | tee -a $OUT/summary.txt
done
# the whole new bench line at chr22 scale: reference binary in its quiet window, both ingest paths, sustained, two secondary legs
( time timeout 1500 python3 bench.py --workload chr22 --cpu-reference yes --secondary lowq50,chr22 --steps 40 --warmup 5 ) > $OUT/bench_chr22_full.json 2> $OUT/bench_chr22_full.err
tail -3 $OUT/bench_chr22_full.err
python3 - $OUT/bench_chr22_full.json <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("bench chr22 full: value %.4g  other %.4g  sustained %s  ingest %s  cpu %s  secondary %s  wall %.0fs" % (j["value"], j["other_input_form"]["value"], j["sustained"] and j["sustained"]["ms_per_step"], j["ingest_end_to_end"] and {k: v.get("value") for k, v in j["ingest_end_to_end"]["paths"].items()}, j["cpu_baseline"] and (j["cpu_baseline"]["value"], j["cpu_baseline"].get("contended_s")), {k: (v.get("value"), v.get("skipped")) for k, v in (j["secondary"] or {}).items()}, j["bench_wall_s"]))
except Exception as e:
    print("bench chr22 full FAILED", e)
PY
