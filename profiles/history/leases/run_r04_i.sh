#!/bin/bash
# Round 4, lease I: the step's gap at hg38 scale -- third-tier workgroups that fit where one main-tier workgroup retired; reads per tier-2 wave.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_i
mkdir -p $OUT
cd $R
ab() {
	local name=$1; shift
	env "$@" timeout 900 python3 bench.py --cpu-sample 0 --no-gather-probe --no-ingest --secondary none --sustain-seconds 0 --steps 20 --warmup 5 > $OUT/$name.json 2> $OUT/$name.err
	python3 - $OUT/$name.json $name <<'PY' | tee -a $OUT/ab_summary.txt
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    d = j["device_ms_per_step"]
    o = j["other_input_form"]
    print("%-14s reads/s %.4g  ms/step %.3f  wave %.3f  pack %.3f  tiers %.3f (deep %.3f)  frac %.3f  spilled %s | gate words: %.4g  ms/step %.3f wave %.3f pack %.3f" % (sys.argv[2], j["value"], j["ms_per_step"], d["wave"], d["pack"], d["spill_tiers_overlapped"], d["of_which_deep_list_wave_tier"], j["roofline"]["frac"], j.get("reads_per_step_redone_by_deep_list_tier"), o["value"], o["ms_per_step"], o["wave_ms"], o["pack_ms"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
ab base X=1
ab w2c64 VG_W2_CHUNK=64
ab w2c32wpc2 VG_W2_CHUNK=32 VG_W2_WPC=2
ab base2 X=1
ab w2c64b VG_W2_CHUNK=64
