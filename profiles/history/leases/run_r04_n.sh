#!/bin/bash
# Round 4, lease N: out-of-tree variants ($VARIANTS, built into variants/) against the same sources without their macro (base0), chr22-scale, both genomes.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${TAG:-r04_n}
mkdir -p $OUT
cd $R
summ() {
	python3 - $OUT/$1.json $1 <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    d = j["device_ms_per_step"]
    o = j["other_input_form"]
    print("%-16s reads/s %.4g  ms/step %.3f  wave %.3f  pack %.3f  tiers %.3f  frac %.3f  spilled %s | gate words: %.4g  ms/step %.3f wave %.3f pack %.3f" % (sys.argv[2], j["value"], j["ms_per_step"], d["wave"], d["pack"], d["spill_tiers_overlapped"], j["roofline"]["frac"], j.get("reads_per_step_redone_by_deep_list_tier"), o["value"], o["ms_per_step"], o["wave_ms"], o["pack_ms"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
c22() { local name=$1; shift; timeout 600 python3 bench.py --workload chr22 --cpu-reference no --no-gather-probe --no-ingest --secondary none --sustain-seconds 0 --steps 40 --warmup 5 "$@" > $OUT/$name.json 2> $OUT/$name.err; summ $name; grep parity $OUT/$name.err | tee -a $OUT/summary.txt; }
for cfg in ${CFGS:-rep30 def}; do
	case $cfg in rep30) A="--repeats 0.3";; def) A="--repeats 0";; lowq50) A="--repeats 0 --lowq 0.5";; esac
	tag=$cfg
	VARGENO_HIP_LIB=$R/variants/base0.so c22 ${tag}_base $A --cpu-sample 0
	for v in $VARIANTS; do VARGENO_HIP_LIB=$R/variants/$v.so c22 ${tag}_${v}_a $A; done
	VARGENO_HIP_LIB=$R/variants/base0.so c22 ${tag}_base_b $A --cpu-sample 0
	for v in $VARIANTS; do VARGENO_HIP_LIB=$R/variants/$v.so c22 ${tag}_${v}_b $A --cpu-sample 0; done
done
