#!/bin/bash
# Round 4, lease K: ONE deep tier behind the main tier -- parity, chr22-scale (default + repeat-rich) and hg38-scale step times.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_k
mkdir -p $OUT
cd $R
( time timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fastq.py -x -q -m gpu --durations=5 ) > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
summ() {
	python3 - $OUT/$1.json $1 <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    d = j["device_ms_per_step"]
    o = j["other_input_form"]
    print("%-16s reads/s %.4g  ms/step %.3f  wave %.3f  pack %.3f  tiers %.3f  frac %.3f  spilled %s lane %s | gate words: %.4g  ms/step %.3f wave %.3f pack %.3f" % (sys.argv[2], j["value"], j["ms_per_step"], d["wave"], d["pack"], d["spill_tiers_overlapped"], j["roofline"]["frac"], j.get("reads_per_step_redone_by_deep_list_tier"), j.get("reads_per_step_sent_on_to_lane_tier"), o["value"], o["ms_per_step"], o["wave_ms"], o["pack_ms"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
c22() { local name=$1; shift; timeout 600 python3 bench.py --workload chr22 --cpu-reference no --no-gather-probe --no-ingest --secondary none --sustain-seconds 0 --steps 40 --warmup 5 "$@" > $OUT/$name.json 2> $OUT/$name.err; summ $name; grep parity $OUT/$name.err | tee -a $OUT/summary.txt; }
c22 c22_rep30 --repeats 0.3
c22 c22_rep30b --repeats 0.3 --cpu-sample 0
VG_W2_WPC=2 c22 c22_rep30_wpc2 --repeats 0.3 --cpu-sample 0
c22 c22_def
c22 c22_defb --cpu-sample 0
h38() { local name=$1; shift; env "$@" timeout 900 python3 bench.py --cpu-sample 0 --no-gather-probe --no-ingest --secondary none --sustain-seconds 0 --steps 20 --warmup 5 > $OUT/$name.json 2> $OUT/$name.err; summ $name; }
h38 hg38 X=1
h38 hg38_wpc2 VG_W2_WPC=2
h38 hg38b X=1
